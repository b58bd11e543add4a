"""Projections of one rank of a tensor-parallel Llama-3-8B (TP = argv[1], default 8; batch M = argv[2],
default 1): are the small shards launch-bound?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, wgemm_norm
dev = "cuda"
tp = int(sys.argv[1]) if len(sys.argv) > 1 else 8
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, I, HQ, HKV, D = 4096, 14336, 32, 8, 128
shapes = [("qkv", (HQ + 2 * HKV) * D // tp, H, "norm"), ("o", H, HQ * D // tp, "plain"),
          ("gate/up", 2 * I // tp, H, "addnorm"), ("down", H, I // tp, "plain")]
tot = 0.0
for name, n, k, mode in shapes:
    copies = max(8, int(300e6 / (n * k * 2)) + 1)
    sw = mode == "addnorm"
    ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, sw) for _ in range(copies)]
    x = torch.randn(M, k, device=dev).to(torch.bfloat16); lw = torch.ones(k, dtype=torch.bfloat16, device=dev)
    r = torch.randn(M, k, device=dev).to(torch.bfloat16); so = torch.empty_like(r)
    y = torch.empty(M, n // 2 if sw else n, dtype=torch.bfloat16, device=dev)
    def fn():
        for w in ws:
            if M > 4: wgemm(x, w, y, swiglu=sw)
            elif mode == "norm": wgemm_norm(x, lw, 1e-5, w, y)
            elif mode == "addnorm": wgemm_norm(x, lw, 1e-5, w, y, add_r=r, sum_out=so, swiglu=True)
            else: wgemm(x, w, y)
    t = timed_graph(fn) / copies
    tot += t
    print(f"TP={tp} M={M} {name:8s} N={n:6d} K={k:6d}: {t*1e6:6.2f} us  {n*k*2/1e6:6.1f} MB  {n*k*2/t/1e12:5.2f} TB/s", flush=True)
print(f"sum {tot*1e6:.1f} us per layer (no attention, no all-reduce)")
