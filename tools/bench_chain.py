"""Projections of one decoder block at Llama-3-8B widths, M rows: separate launches against
mv_decode_chain with 4, 2+2 and 1+1+1+1 phases per launch (us per block, hipGraph of `layers` blocks
with distinct weights, HIP events).  Usage: python tools/bench_chain.py [M] [layers]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import micro_vllm_amd  # noqa: F401
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import (StreamedWeight, chain_phase, decode_chain, gemm_workspace, wgemm,
                                          wgemm_norm)

M = int(sys.argv[1]) if len(sys.argv) > 1 else 1
LAYERS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = "cuda"
hid, qkv_n, inter = 4096, 6144, 14336
bf = dict(dtype=torch.bfloat16, device=dev)


def sw(n, k, swiglu=False):
    return StreamedWeight(torch.empty(n * k, **bf).normal_(0, 0.02), n, k, swiglu)


layers = [dict(o=sw(hid, hid), gu=sw(2 * inter, hid, True), d=sw(hid, inter), qkv=sw(qkv_n, hid)) for _ in range(LAYERS)]
ln = torch.ones(hid, **bf)
attn = torch.randn(M, hid, **bf)
attn_out, resid, mlp_out, h0 = (torch.empty(M, hid, **bf) for _ in range(4))
h0.normal_()
act = torch.empty(M, inter, **bf)
qkv = torch.empty(M, qkv_n, **bf)
ws = gemm_workspace(dev, 1 << 20)
cws = torch.zeros(nv.lib().mv_decode_chain_workspace_bytes() // 4 + 4, dtype=torch.float32, device=dev)


def phases(L, h_in):
    return [chain_phase(attn, L["o"], attn_out),
            chain_phase(attn_out, L["gu"], act, ln_weight=ln, eps=1e-5, add_r=h_in, sum_out=resid, swiglu=True),
            chain_phase(act, L["d"], mlp_out, residual=resid),
            chain_phase(mlp_out, L["qkv"], qkv, ln_weight=ln, eps=1e-5)]


def separate():
    h_in = h0
    for L in layers:
        wgemm(attn, L["o"], attn_out, workspace=ws)
        wgemm_norm(attn_out, ln, 1e-5, L["gu"], act, add_r=h_in, sum_out=resid, swiglu=True, workspace=ws)
        wgemm(act, L["d"], mlp_out, residual=resid, workspace=ws)
        wgemm_norm(mlp_out, ln, 1e-5, L["qkv"], qkv, workspace=ws)
        h_in = mlp_out


def chained(groups):
    def run():
        h_in = h0
        for L in layers:
            ph = phases(L, h_in)
            i = 0
            for g in groups:
                assert decode_chain(ph[i:i + g], M, cws)
                i += g
            h_in = mlp_out
    return run


def timed(fn, reps=10):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
        side.synchronize()
        nv.check("b", nv.lib().mv_graph_begin_capture(side.cuda_stream))
        fn()
        h = ctypes.c_void_p()
        nv.check("e", nv.lib().mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
        for _ in range(2):
            nv.lib().mv_graph_launch(h, side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            nv.lib().mv_graph_launch(h, side.cuda_stream)
        e1.record(side)
        side.synchronize()
        nv.lib().mv_graph_destroy(h)
    torch.cuda.current_stream().wait_stream(side)
    return e0.elapsed_time(e1) * 1e3 / (reps * LAYERS)


mb = sum(w.nbytes for w in layers[0].values()) / 1e6
for name, fn in (("separate launches", separate), ("chain 1+1+1+1", chained([1, 1, 1, 1])),
                 ("chain 2+2", chained([2, 2])), ("chain 4", chained([4]))):
    t = timed(fn)
    print(f"M={M} {name:20s} {t:8.2f} us per block  ({mb / t * 1e-3:.2f} TB/s)", flush=True)
st = ctypes.c_int(-1)
nv.lib().mv_decode_chain_status(cws.data_ptr(), ctypes.byref(st))
print("chain status", st.value)

# ---- each projection alone: stand-alone launch against a one-phase chain launch -----------------
def per_proj(idx, name):
    def sep():
        h_in = h0
        for L in layers:
            if idx == 0:
                wgemm(attn, L["o"], attn_out, workspace=ws)
            elif idx == 1:
                wgemm_norm(attn_out, ln, 1e-5, L["gu"], act, add_r=h_in, sum_out=resid, swiglu=True, workspace=ws)
            elif idx == 2:
                wgemm(act, L["d"], mlp_out, residual=resid, workspace=ws)
            else:
                wgemm_norm(mlp_out, ln, 1e-5, L["qkv"], qkv, workspace=ws)

    def ch():
        for L in layers:
            assert decode_chain(phases(L, h0)[idx:idx + 1], M, cws)

    print(f"M={M} {name:8s} stand-alone {timed(sep):7.2f} us   one-phase chain {timed(ch):7.2f} us", flush=True)


for i, nm in enumerate(("O", "gate/up", "down", "QKV")):
    per_proj(i, nm)
