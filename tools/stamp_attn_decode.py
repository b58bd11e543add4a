"""Where the workgroups of one decode-attention launch spend their time (stamped build):
   python -c "import sys; sys.path.insert(0, 'micro-vllm_amd'); import build; \
              build.build_variant('astamp', ['MV_ATTN_STAMP'], only=('attn_decode.hip',))"
   python tools/stamp_attn_decode.py BS HQ HKV D CTX SPLITS
Stamps are the constant 100 MHz counter (10 ns per tick) and are comparable across workgroups;
times below are microseconds after the earliest workgroup start of the launch."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MVLLM_LIB"] = os.path.join(ROOT, "micro-vllm_amd", "lib", "libmvllm_astamp.so")
sys.path.insert(0, ROOT)
import numpy as np
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
L.mv_debug_read_attn_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
ws = torch.zeros(L.mv_paged_attn_decode_workspace_bytes(bs, hq, d, max(splits, 2)) // 4 + 4, dtype=torch.float32, device=dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
names = ["entry", "q/k ready", "rows absorbed", "partials issued", "stores drained", "ticket known", "merge summed", "end",
         "first chunk absorbed", "merge: lse landed", "merge: weights done", "pos known", "page ids landed"]
order = [0, 11, 12, 1, 8, 2, 3, 4, 5, 9, 10, 6, 7]
for rep in range(4):
    if "--noflush" not in sys.argv:
        flush.zero_()                 # push the KV rows out of the caches, as a real step's weights do
    if "--touch" in sys.argv:         # one element per page: translations warm, rows still cold
        tk = kc[:, 0, 0, 0].float().sum() + vc[:, 0, 0, 0].float().sum() + cos[ctx, 0].float() + sin[ctx, 0].float()
    torch.cuda.synchronize()
    L.mv_debug_clear_attn_stamps()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    paged_attn_decode(q, k, v, kc, vc, sl, perm, cos, sin, num_splits=splits, out=out, workspace=ws)
    e1.record(); torch.cuda.synchronize()
    buf = np.zeros(16 * 4096, dtype=np.uint64)
    assert L.mv_debug_read_attn_stamps(buf.ctypes.data, buf.nbytes) == 0
    nblk = min(4096, bs * hkv * splits)
    st = buf.reshape(4096, 16)[:nblk].astype(np.float64)
    t0 = st[:, 0].min()
    print(f"rep {rep}: launch-to-end by events {e0.elapsed_time(e1)*1e3:.1f} us, {nblk} workgroups")
    for sl_ in order:
        col = st[:, sl_]
        have = col > 0
        if not have.any():
            continue
        us = (col[have] - t0) / 100.0
        print(f"   {names[sl_]:22s} n={have.sum():4d}  min {us.min():6.2f}  median {np.median(us):6.2f}  max {us.max():6.2f}")
