"""GEMV time when the weight is re-read every launch (cache-resident) vs rotated over >600 MB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm
dev = "cuda"
for (n, k) in [(4096, 4096), (6144, 4096), (4096, 14336), (28672, 4096)]:
    for copies in (1, 2, 4, max(4, int(600e6 / (n * k * 2)) + 1)):
        ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k) for _ in range(copies)]
        x = torch.randn(1, k, device=dev).to(torch.bfloat16)
        y = torch.empty(1, n, dtype=torch.bfloat16, device=dev)
        def fn():
            for r in range(max(1, 16 // copies)):
                for w in ws:
                    wgemm(x, w, y)
        t = timed_graph(fn) / (copies * max(1, 16 // copies))
        print(f"N={n} K={k} copies={copies} ({copies*n*k*2/1e6:.0f} MB): {t*1e6:.2f} us {n*k*2/t/1e12:.2f} TB/s", flush=True)
