"""A/B of the wide prefill GEMM's tile width and of cutting left-over tiles, in ONE process with the
variants interleaved (separate short processes differ by 20 % with the chip's clocks):
python tools/ab_gemm_prefill.py M N K [swiglu]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
m, n, k = (int(a) for a in sys.argv[1:4])
sw = len(sys.argv) > 4
L = nv.lib()
copies = max(4, int(600e6 / (n * k * 2)) + 1)
ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02), n, k, sw) for _ in range(copies)]
x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
y = torch.empty(m, n // 2 if sw else n, dtype=torch.bfloat16, device="cuda")
variants = [(c, s) for s in (16, 14, 12) for c in (1, 0)] + [(-1, -1)]
times = {v: [] for v in variants}
def run(v):
    L.mv_debug_prefill_options(*v)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for w in ws:
        gemm_prefill(x, w, y, swiglu=sw)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / copies * 1e3
for _ in range(3):
    for v in variants: run(v)
for _ in range(8):
    for v in variants: times[v].append(run(v))
for v in variants:
    t = np.array(times[v])
    name = "the plan's own choice" if v == (-1, -1) else f"{v[1]} strips, cut {v[0]}"
    print(f"M={m} N={n} K={k}{' swiglu' if sw else ''}  {name:22s}: median {np.median(t):7.1f} us  min {t.min():7.1f}")
