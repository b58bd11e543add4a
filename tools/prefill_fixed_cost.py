"""Time of the wide prefill GEMM against k at fixed M x N: the intercept is what a round of tiles costs
before and after its k walk (launch, pipeline fill, epilogue): python tools/prefill_fixed_cost.py [M]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
dev = "cuda"
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for n in (4096, 28672):
    for k in (256, 512, 1024, 2048, 4096, 8192):
        copies = max(4, int(600e6 / (n * k * 2)) + 1)
        ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, False)
              for _ in range(copies)]
        x = torch.randn(m, k, device=dev).to(torch.bfloat16)
        y = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        for w in ws:
            gemm_prefill(x, w, y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 3
        e0.record()
        for _ in range(reps):
            for w in ws:
                gemm_prefill(x, w, y)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / (reps * copies) * 1e-3
        print(f"M={m} N={n:6d} K={k:5d}: {t*1e6:8.1f} us  {2*m*n*k/t/1e12:7.1f} TFLOP/s", flush=True)
        del ws
