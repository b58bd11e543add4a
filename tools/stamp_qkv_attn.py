"""Where the workgroups of the fused QKV + attention launch spend their time (stamped build):
   python -c "import sys; sys.path.insert(0, 'micro-vllm_amd'); import build; \
              build.build_variant('qstamp', ['MV_QA_STAMP'], only=('qkv_attn.hip',))"
   python tools/stamp_qkv_attn.py BS CTX [--noflush]
8B head shape (32 query heads on 8 kv heads of 128, hidden 4096).  Stamps are the constant 100 MHz
counter (10 ns per tick), comparable across workgroups; times are microseconds after the earliest
workgroup start of the launch.  The two stand-alone launches are timed beside it with events."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MVLLM_LIB"] = os.path.join(ROOT, "micro-vllm_amd", "lib", "libmvllm_qstamp.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import micro_vllm_amd
from micro_vllm_amd import _native as nv
from micro_vllm_amd.core.paged_attention import paged_attn_decode, qkv_attn_decode, rotary_tables
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm_norm

bs, ctx = int(sys.argv[1]), int(sys.argv[2])
hq, hkv, d, hidden = 32, 8, 128, 4096
dev = "cuda"
cus = torch.cuda.get_device_properties(0).multi_processor_count
splits = cus // (bs * hkv)
pages = (ctx + 256) // 256 + 1
nb = bs * pages
n = (hq + 2 * hkv) * d
kc = torch.randn(nb, 256, hkv, d, device=dev).to(torch.bfloat16)
vc = torch.randn(nb, 256, hkv, d, device=dev).to(torch.bfloat16)
x = torch.randn(bs, hidden, device=dev).to(torch.bfloat16)
lnw = torch.ones(hidden, device=dev, dtype=torch.bfloat16)
w = StreamedWeight.from_linear((torch.randn(n, hidden, device=dev) * 0.03).to(torch.bfloat16))
qkv = torch.empty(bs, n, dtype=torch.bfloat16, device=dev)
perm = torch.randperm(nb, device=dev).to(torch.int32).reshape(bs, pages)
sl = torch.full((bs,), ctx, dtype=torch.int32, device=dev)
cos, sin = rotary_tables(d, 16384, device=dev)
out = torch.empty(bs, hq, d, dtype=torch.bfloat16, device=dev)
L = nv.lib()
L.mv_debug_read_qa_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
ws = torch.zeros(L.mv_paged_attn_decode_workspace_bytes(bs, hq, d, splits) // 4 + 4, dtype=torch.float32, device=dev)
sync = torch.zeros((L.mv_qkv_attn_workspace_bytes() + 3) // 4, dtype=torch.int32, device=dev)
status = torch.zeros(1, dtype=torch.int32).pin_memory()
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
names = ["entry", "x staged", "strips done", "stores drained", "head's flags up", "q, k, v landed",
         "rows absorbed", "end"]
for rep in range(4):
    if "--noflush" not in sys.argv:
        flush.zero_()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert qkv_attn_decode(x, lnw, 1e-5, w, None, qkv, hq, kc, vc, sl, perm, cos, sin, splits, out, ws, sync, status)
    e1.record(); torch.cuda.synchronize()
    buf = np.zeros(16 * 1024, dtype=np.uint64)
    assert L.mv_debug_read_qa_stamps(buf.ctypes.data, buf.nbytes) == 0
    nblk = bs * hkv * splits
    st = buf.reshape(1024, 16)[:nblk].astype(np.float64)
    t0 = st[:, 0].min()
    print(f"rep {rep}: fused launch by events {e0.elapsed_time(e1)*1e3:.1f} us, {nblk} workgroups, status {int(status[0])}")
    for i, name in enumerate(names):
        col = st[:, i]
        have = col > 0
        if not have.any():
            continue
        us = (col[have] - t0) / 100.0
        print(f"   {name:18s} n={have.sum():4d}  min {us.min():6.2f}  median {np.median(us):6.2f}  max {us.max():6.2f}")
    heavy = np.arange(nblk) // hkv < (6 * d // 16 - nblk // hkv)     # members with two strips
    if heavy.any() and (~heavy).any():
        print(f"   strips done, two-strip workgroups median {np.median((st[heavy, 2] - t0) / 100):6.2f}, "
              f"one-strip {np.median((st[~heavy, 2] - t0) / 100):6.2f}")
    if "--noflush" not in sys.argv:
        flush.zero_()
    torch.cuda.synchronize()
    e0.record()
    wgemm_norm(x, lnw, 1e-5, w, qkv)
    e1.record()
    q = qkv[:, :hq * d].unflatten(1, (hq, d)); k = qkv[:, hq * d:(hq + hkv) * d].unflatten(1, (hkv, d)); v = qkv[:, (hq + hkv) * d:].unflatten(1, (hkv, d))
    paged_attn_decode(q, k, v, kc, vc, sl, perm, cos, sin, num_splits=splits, out=out, workspace=ws)
    e2 = torch.cuda.Event(enable_timing=True); e2.record(); torch.cuda.synchronize()
    print(f"   two launches by events: projection {e0.elapsed_time(e1)*1e3:.1f} us, + attention {e0.elapsed_time(e2)*1e3:.1f} us")
