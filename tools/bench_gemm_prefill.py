"""TFLOP/s of mv_gemm_prefill on the prefill projections of the 8B shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
dev = "cuda"
ms = [int(a) for a in sys.argv[1:]] or [512, 2048, 4096]
for m in ms:
    for (n, k, sw) in [(6144, 4096, False), (4096, 4096, False), (28672, 4096, True), (4096, 14336, False)]:
        w = StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, sw)
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        y = torch.empty(m, n // 2 if sw else n, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            gemm_prefill(x, w, y, swiglu=sw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            gemm_prefill(x, w, y, swiglu=sw)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / reps * 1e-3
        print(f"M={m:5d} N={n:6d} K={k:6d}: {t*1e6:9.1f} us  {2*m*n*k/t/1e12:7.1f} TFLOP/s", flush=True)
