"""Where the workgroups of a chained-projection launch spend their time (stamped build):
   python -c "import sys; sys.path.insert(0,'micro-vllm_amd'); import build; build.build_variant('cstamp', ['MV_CHAIN_STAMP'], only=('decode_chain.hip',))"
   python tools/stamp_chain.py [M]
Per phase, over the workgroups of the last launch (100 MHz ticks -> us, relative to the earliest
phase-0 start): when the phase started (barrier passed), x staged, main loop done, stores
acknowledged (arrival)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MVLLM_LIB"] = os.path.join(ROOT, "micro-vllm_amd", "lib", "libmvllm_cstamp.so")
import numpy as np
import torch

import micro_vllm_amd  # noqa: F401
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, chain_phase, decode_chain

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda"
hid, qkv_n, inter = 4096, 6144, 14336
bf = dict(dtype=torch.bfloat16, device=dev)


def sw(n, k, swiglu=False):
    return StreamedWeight(torch.empty(n * k, **bf).normal_(0, 0.02), n, k, swiglu)


layers = [dict(o=sw(hid, hid), gu=sw(2 * inter, hid, True), d=sw(hid, inter), qkv=sw(qkv_n, hid)) for _ in range(6)]
ln = torch.ones(hid, **bf)
attn = torch.randn(M, hid, **bf)
attn_out, resid, mlp_out, h0 = (torch.empty(M, hid, **bf) for _ in range(4))
h0.normal_()
act = torch.empty(M, inter, **bf)
qkv = torch.empty(M, qkv_n, **bf)
cws = torch.zeros(nv.lib().mv_decode_chain_workspace_bytes() // 4 + 4, dtype=torch.float32, device=dev)
for rep in range(3):
    for L in layers:
        ph = [chain_phase(attn, L["o"], attn_out),
              chain_phase(attn_out, L["gu"], act, ln_weight=ln, eps=1e-5, add_r=h0, sum_out=resid, swiglu=True),
              chain_phase(act, L["d"], mlp_out, residual=resid),
              chain_phase(mlp_out, L["qkv"], qkv, ln_weight=ln, eps=1e-5)]
        assert decode_chain(ph, M, cws)
torch.cuda.synchronize()
n = 512 * 4 * 4
buf = (ctypes.c_ulonglong * n)()
lib = nv.lib()
lib.mv_debug_read_chain_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.mv_debug_read_chain_stamps(buf, n * 8) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(512, 4, 4).astype(np.int64)
nwg = torch.cuda.get_device_properties(0).multi_processor_count
st = st[:nwg]
t0 = st[:, 0, 0].min()
us = (st - t0) / 100.0
names = ["O", "gate/up", "down", "QKV"]
slots = ["start", "x staged", "loop done", "arrived"]
for p in range(4):
    for sl in range(4):
        if p == 3 and sl == 3:
            continue
        v = us[:, p, sl]
        print(f"{names[p]:8s} {slots[sl]:10s} min {v.min():7.2f}  median {np.median(v):7.2f}  max {v.max():7.2f}")
