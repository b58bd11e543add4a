"""Sweep strips-per-workgroup (SG) and k-split for the M > 4 weight-streaming kernel:
python tools/sweep_mfma_gemv.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, gemm_workspace
L = nv.lib()
dev = "cuda"
ws = gemm_workspace(dev, 256 << 20)
for m in ([int(a) for a in sys.argv[1:]] or [32]):
    for (n, k, sw) in [(6144, 4096, False), (4096, 4096, False), (28672, 4096, True), (4096, 14336, False)]:
        copies = max(4, int(600e6 / (n * k * 2)) + 1)
        wts = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, sw) for _ in range(copies)]
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        y = torch.empty(m, n // 2 if sw else n, dtype=torch.bfloat16, device=dev)
        res = []
        for sg in (1, 2, 3, 4, 7):
            if L.mv_wgemm_tune(3, sg) != 0:
                continue
            for ks in (1, 2, 3, 4, 8):
                L.mv_wgemm_tune(1, ks)
                for bpc in (1, 2):
                    L.mv_wgemm_tune(0, bpc)
                    def fn():
                        for w in wts:
                            wgemm(x, w, y, swiglu=sw, workspace=ws)
                    t = timed_graph(fn) / copies
                    res.append((t, sg, ks, bpc))
        res.sort()
        L.mv_wgemm_tune(3, 0); L.mv_wgemm_tune(1, 0); L.mv_wgemm_tune(0, 0)
        def fn():
            for w in wts:
                wgemm(x, w, y, swiglu=sw, workspace=ws)
        t0 = timed_graph(fn) / copies
        best = " | ".join(f"sg{sg} ks{ks} bpc{bpc}: {t*1e6:.1f}us {n*k*2/t/1e12:.2f}TB/s" for t, sg, ks, bpc in res[:4])
        print(f"M={m} N={n} K={k}: planner {t0*1e6:.1f}us {n*k*2/t0/1e12:.2f}TB/s || {best}", flush=True)
