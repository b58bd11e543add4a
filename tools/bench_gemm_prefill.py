"""TFLOP/s of mv_gemm_prefill on the prefill projections of the 8B shape.  Weights rotate over
more than 600 MB so that every launch streams them from HBM (a single weight would sit in the
Infinity Cache and flatter short prompts): python tools/bench_gemm_prefill.py [M ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
dev = "cuda"
ms = [int(a) for a in sys.argv[1:]] or [512, 2048, 4096]
for m in ms:
    tot = 0.0
    for (n, k, sw) in [(6144, 4096, False), (4096, 4096, False), (28672, 4096, True), (4096, 14336, False)]:
        copies = max(4, int(600e6 / (n * k * 2)) + 1)
        ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, sw)
              for _ in range(copies)]
        pad = int(os.environ.get("XPAD", "0"))     # extra elements per row of x (row stride k + pad)
        x = torch.randn(m, k + pad, device=dev).to(torch.bfloat16)[:, :k]
        ypad = int(os.environ.get("YPAD", "0"))    # extra elements per row of y
        nc = n // 2 if sw else n
        y = torch.empty(m, nc + ypad, dtype=torch.bfloat16, device=dev)[:, :nc]
        for w in ws:
            gemm_prefill(x, w, y, swiglu=sw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        bands = os.environ.get("MBANDS")           # A/B inside one process: tile-walk band heights
        for band in (bands.split(",") if bands else [None]):
            if band is not None:
                os.environ["MVLLM_PREFILL_MBAND"] = band
            e0.record()
            for _ in range(reps):
                for w in ws:
                    gemm_prefill(x, w, y, swiglu=sw)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / (reps * copies) * 1e-3
            print(f"M={m:5d} N={n:6d} K={k:6d}: {t*1e6:9.1f} us  {2*m*n*k/t/1e12:7.1f} TFLOP/s"
                  + (f"  band {band}" if band is not None else ""), flush=True)
        tot += t
    print(f"M={m:5d} sum {tot*1e6:.1f} us", flush=True)
