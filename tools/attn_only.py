"""A few eager launches of the decode attention kernel (for counter collection):
python tools/attn_only.py BS HQ HKV D CTX SPLITS"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_vllm_amd
from micro_vllm_amd import _native as nv
from micro_vllm_amd.core.paged_attention import paged_attn_decode, rotary_tables
bs, hq, hkv, d, ctx, splits = [int(a) for a in sys.argv[1:7]]
dev = "cuda"
pages = (ctx + 256) // 256 + 1
nb = bs * pages
kc = torch.randn(nb, 256, hkv, d, device=dev).to(torch.bfloat16)
vc = torch.randn(nb, 256, hkv, d, device=dev).to(torch.bfloat16)
qkv = torch.randn(bs, (hq + 2 * hkv) * d, device=dev).to(torch.bfloat16)
q = qkv[:, :hq * d].unflatten(1, (hq, d)); k = qkv[:, hq * d:(hq + hkv) * d].unflatten(1, (hkv, d)); v = qkv[:, (hq + hkv) * d:].unflatten(1, (hkv, d))
perm = torch.randperm(nb, device=dev).to(torch.int32).reshape(bs, pages)
sl = torch.full((bs,), ctx, dtype=torch.int32, device=dev)
cos, sin = rotary_tables(d, 16384, device=dev)
out = torch.empty(bs, hq, d, dtype=torch.bfloat16, device=dev)
L = nv.lib()
ws = torch.zeros(L.mv_paged_attn_decode_workspace_bytes(bs, hq, d, max(splits, 2)) // 4 + 4, dtype=torch.float32, device=dev)
for _ in range(5):
    paged_attn_decode(q, k, v, kc, vc, sl, perm, cos, sin, num_splits=splits, out=out, workspace=ws)
torch.cuda.synchronize()
