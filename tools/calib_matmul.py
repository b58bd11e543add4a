import torch
dev="cuda"
for m in (2048, 4096):
    for n,k in [(6144,4096),(4096,4096),(28672,4096),(4096,14336)]:
        x=torch.randn(m,k,device=dev).to(torch.bfloat16); w=(torch.randn(n,k,device=dev)*0.02).to(torch.bfloat16)
        for _ in range(3): y=x@w.t()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): y=x@w.t()
        e1.record(); torch.cuda.synchronize()
        t=e0.elapsed_time(e1)/20*1e-3
        print(f"hipBLASLt M={m} N={n} K={k}: {t*1e6:.1f} us {2*m*n*k/t/1e12:.0f} TFLOP/s", flush=True)
