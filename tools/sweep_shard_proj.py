"""The four decode projections of one tensor-parallel rank over forced (strips per workgroup,
k-slice count) pairs, timed as slab calls (the product left as fp32 k-slabs: the projection alone,
without the launch that adds them) next to what the planner picks:
python tools/sweep_shard_proj.py [M] [hidden heads kv_heads ffn tp]   (default: 8, Llama-3-70B, TP 8)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm_slabs, gemm_workspace
dev = "cuda"
a = [int(v) for v in sys.argv[1:]]
m = a[0] if a else 8
hid, hq, hkv, ffn, tp = a[1:6] if len(a) >= 6 else (8192, 64, 8, 28672, 8)
d = 128
ws = gemm_workspace(dev, 512 << 20)
L = nv.lib()
shapes = [("qkv", (hq + 2 * hkv) * d // tp, hid), ("o", hid, hq * d // tp),
          ("gate_up", 2 * ffn // tp, hid), ("down", hid, ffn // tp)]
for name, n, k in shapes:
    copies = max(4, int(600e6 / (n * k * 2)) + 1)
    wts = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, False) for _ in range(copies)]
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    def fn():
        for w in wts:
            wgemm_slabs(x, w, ws)
    auto = timed_graph(fn) / copies * 1e6
    print(f"M={m} {name:8s} N={n} K={k} ({n*k*2/1e6:.0f} MB)  planner: {auto:6.1f} us ({L.mv_wgemm_slab_count(m, n, k)} slabs)", flush=True)
    for sg in (1, 2, 3, 4, 7):
        if sg in (3, 7) and m > 32:
            continue
        row = []
        for ks in (1, 2, 4, 8):
            L.mv_wgemm_tune(3, sg)
            L.mv_wgemm_tune(1, ks)
            if L.mv_wgemm_slab_count(m, n, k) != ks:
                row.append(f"ks={ks}:    -  ")
                continue
            row.append(f"ks={ks}: {timed_graph(fn) / copies * 1e6:6.1f}")
        print(f"     sg={sg}  " + "  ".join(row), flush=True)
    L.mv_wgemm_tune(3, 0)
    L.mv_wgemm_tune(1, 0)
