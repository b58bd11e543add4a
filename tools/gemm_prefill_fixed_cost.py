"""Fixed cost of a prefill GEMM launch: one round of 256 x 256 tiles (M = 4096, N = 4096: 256 tiles on
256 CUs) at several K, and several rounds (N = 4096 r) at one K; weights rotate so that they stream
from HBM.  A least-squares line through the K sweep gives the per-k-tile time and what a round costs
beyond its k-tiles: python tools/gemm_prefill_fixed_cost.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
dev = "cuda"
M = 4096

def run(n, k):
    copies = max(4, int(600e6 / (n * k * 2)) + 1)
    ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, False)
          for _ in range(copies)]
    x = torch.randn(M, k, device=dev).to(torch.bfloat16)
    y = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    for w in ws:
        gemm_prefill(x, w, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for w in ws:
            gemm_prefill(x, w, y)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * copies) * 1e3

ks = [1024, 2048, 4096, 8192, 14336]
ts = [run(4096, k) for k in ks]
for k, t in zip(ks, ts):
    print(f"N=4096 K={k:6d}: {t:7.1f} us  {2 * M * 4096 * k / t / 1e6:7.1f} TFLOP/s", flush=True)
A = np.vstack([np.array(ks) / 32.0, np.ones(len(ks))]).T
slope, icpt = np.linalg.lstsq(A, np.array(ts), rcond=None)[0]
print(f"one round: {slope:.3f} us per 32-wide k-tile + {icpt:.1f} us fixed")
for r in (1, 2, 3, 4, 7):
    t = run(4096 * r, 4096)
    print(f"N={4096 * r:6d} K=4096 ({r} rounds): {t:7.1f} us = {t / r:6.1f} per round, {2 * M * 4096 * r * 4096 / t / 1e6:7.1f} TFLOP/s", flush=True)
