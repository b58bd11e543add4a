"""M=1 projections with few strips (tensor-parallel shards): forced k-split 1/2/4/8 against the planner."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, wgemm_norm, gemm_workspace
dev = "cuda"
L = nv.lib()
ws = gemm_workspace(dev, 64 << 20)
for (n, k, mode) in [(768, 4096, "norm"), (1536, 4096, "norm"), (3072, 4096, "norm"), (4096, 512, "plain"),
                     (4096, 1024, "plain"), (3584, 4096, "addnorm"), (4096, 1792, "plain"), (4096, 3584, "plain")]:
    copies = max(8, int(300e6 / (n * k * 2)) + 1)
    sw = mode == "addnorm"
    wts = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, sw) for _ in range(copies)]
    x = torch.randn(1, k, device=dev).to(torch.bfloat16); lw = torch.ones(k, dtype=torch.bfloat16, device=dev)
    r = torch.randn(1, k, device=dev).to(torch.bfloat16); so = torch.empty_like(r)
    y = torch.empty(1, n // 2 if sw else n, dtype=torch.bfloat16, device=dev)
    res = []
    for ks in (0, 1, 2, 4, 8):
        L.mv_wgemm_tune(1, ks)
        for waves in (0, 8, 16):
            L.mv_wgemm_tune(2, waves)
            def fn():
                for w in wts:
                    if mode == "norm": wgemm_norm(x, lw, 1e-5, w, y, workspace=ws)
                    elif mode == "addnorm": wgemm_norm(x, lw, 1e-5, w, y, add_r=r, sum_out=so, swiglu=True, workspace=ws)
                    else: wgemm(x, w, y, workspace=ws)
            t = timed_graph(fn) / copies
            res.append((ks, waves, t))
    L.mv_wgemm_tune(1, 0); L.mv_wgemm_tune(2, 0)
    print(f"N={n:5d} K={k:5d} {mode:8s} " + " ".join(f"ks{ks}w{wv}:{t*1e6:.1f}" for ks, wv, t in res), flush=True)
