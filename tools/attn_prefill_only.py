"""A few launches of the prefill attention kernel (8B shape, one layer) for timing / counters:
python tools/attn_prefill_only.py [S] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_vllm_amd
from micro_vllm_amd.core.prefill_attention import paged_attn_prefill_rows
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
hq, hkv, d = 32, 8, 128
dev = "cuda"
pages = (S + 255) // 256
nb = B * pages
kc = torch.randn(nb, 256, hkv, d, device=dev).to(torch.bfloat16)
vc = torch.randn(nb, 256, hkv, d, device=dev).to(torch.bfloat16)
q = torch.randn(B * S, hq * d, device=dev).to(torch.bfloat16)
out = torch.empty_like(q)
bt = torch.randperm(nb, device=dev).to(torch.int32).reshape(B, pages)
sl = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(3):
    paged_attn_prefill_rows(q, B, S, hq, d, kc, vc, sl, bt, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    paged_attn_prefill_rows(q, B, S, hq, d, kc, vc, sl, bt, out)
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10 * 1e-3
fl = 4.0 * hq * d * B * S * S / 2
print(f"prefill attention B={B} S={S}: {t*1e6:.1f} us  {fl/t/1e12:.0f} TFLOP/s", flush=True)
