"""The four decode projections of one tensor-parallel rank over forced k-slice counts:
python tools/sweep_shard_proj.py [M] [hidden heads kv_heads ffn tp]   (default: 8, Llama-3-70B, TP 8)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, gemm_workspace
dev = "cuda"
a = [int(v) for v in sys.argv[1:]]
m = a[0] if a else 8
hid, hq, hkv, ffn, tp = a[1:6] if len(a) >= 6 else (8192, 64, 8, 28672, 8)
d = 128
ws = gemm_workspace(dev, 256 << 20)
shapes = [("qkv", (hq + 2 * hkv) * d // tp, hid, False), ("o", hid, hq * d // tp, False),
          ("gate_up", 2 * ffn // tp, hid, True), ("down", hid, ffn // tp, False)]
for name, n, k, sw in shapes:
    copies = max(4, int(600e6 / (n * k * 2)) + 1)
    wts = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, sw) for _ in range(copies)]
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    y = torch.empty(m, n // 2 if sw else n, dtype=torch.bfloat16, device=dev)
    def fn():
        for w in wts:
            wgemm(x, w, y, swiglu=sw, workspace=ws)
    row = []
    for ks in (0, 1, 2, 4, 8):
        nv.lib().mv_wgemm_tune(1, ks)
        row.append(f"ks={ks or 'auto'}: {timed_graph(fn) / copies * 1e6:6.1f}")
    nv.lib().mv_wgemm_tune(1, 0)
    print(f"M={m} {name:8s} N={n} K={k} ({n*k*2/1e6:.0f} MB)  " + "  ".join(row), flush=True)
