"""The four decode projections at batch M through the planner (no sweep):
python tools/bench_mfma_gemv.py [M ...]; MV_BENCH_SG=1|2|4 forces the strips per workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, gemm_workspace
dev = "cuda"
ws = gemm_workspace(dev, 256 << 20)
if os.environ.get("MV_BENCH_SG"):
    nv.lib().mv_wgemm_tune(3, int(os.environ["MV_BENCH_SG"]))
for m in ([int(a) for a in sys.argv[1:]] or [32]):
    tot = 0.0
    for (n, k, sw) in [(6144, 4096, False), (4096, 4096, False), (28672, 4096, True), (4096, 14336, False)]:
        copies = max(4, int(600e6 / (n * k * 2)) + 1)
        wts = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, sw) for _ in range(copies)]
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        y = torch.empty(m, n // 2 if sw else n, dtype=torch.bfloat16, device=dev)
        def fn():
            for w in wts:
                wgemm(x, w, y, swiglu=sw, workspace=ws)
        t = timed_graph(fn) / copies
        tot += t
        print(f"M={m} N={n} K={k}: {t*1e6:.1f}us {n*k*2/t/1e12:.2f}TB/s", flush=True)
    print(f"M={m} sum {tot*1e6:.1f} us", flush=True)
