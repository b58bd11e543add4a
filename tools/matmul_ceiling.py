"""What the vendor library reaches on the prefill projection shapes of the 8B model (torch.matmul ->
hipBLASLt), weights rotating over > 600 MB like tools/bench_gemm_prefill.py: a practical ceiling for
this chip under its power limit.  python tools/matmul_ceiling.py [M]"""
import sys
import torch
m = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
tot = 0.0
for n, k in [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336)]:
    copies = max(4, int(600e6 / (n * k * 2)) + 1)
    ws = [torch.empty(n, k, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02) for _ in range(copies)]
    x = torch.randn(m, k, device="cuda").to(torch.bfloat16)
    y = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
    for w in ws:
        torch.matmul(x, w.t(), out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        for w in ws:
            torch.matmul(x, w.t(), out=y)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (3 * copies) * 1e-3
    tot += t
    print(f"M={m:5d} N={n:6d} K={k:6d}: {t*1e6:9.1f} us  {2*m*n*k/t/1e12:7.1f} TFLOP/s", flush=True)
print(f"M={m:5d} sum {tot*1e6:.1f} us")
