"""The decode projections at batch M as ordinary calls and as slab calls (product left as fp32
k-slabs for the next kernel to add): python tools/bench_slabs.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, wgemm_slabs, gemm_workspace
dev = "cuda"
ws = gemm_workspace(dev, 256 << 20)
for m in ([int(a) for a in sys.argv[1:]] or [32]):
    for (n, k) in [(6144, 4096), (4096, 4096), (4096, 14336), (768, 4096), (4096, 512), (4096, 1792)]:
        copies = max(4, int(600e6 / (n * k * 2)) + 1)
        wts = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, False) for _ in range(copies)]
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        def plain():
            for w in wts:
                wgemm(x, w, y, workspace=ws)
        def slabs():
            for w in wts:
                wgemm_slabs(x, w, ws)
        t0, t1 = timed_graph(plain) / copies, timed_graph(slabs) / copies
        ks = nv.lib().mv_wgemm_slab_count(m, n, k)
        # same sums: plan_slabs call == slab sum + rounding
        y2 = torch.empty_like(y)
        wgemm(x, wts[0], y2, workspace=ws, plan_slabs=True)
        wgemm_slabs(x, wts[0], ws)
        tot = ws.view(torch.float32)[: ks * m * n].view(ks, m, n)
        acc = torch.zeros(m, n, device=dev)
        for i in range(ks):
            acc = acc + tot[i]
        same = bool(torch.equal(acc.to(torch.bfloat16), y2))
        print(f"M={m} N={n} K={k}: plain {t0*1e6:6.1f} us, slabs {t1*1e6:6.1f} us ({ks} slabs), plan_slabs==slab sum: {same}", flush=True)
