"""Time the paged decode attention kernel alone (hipGraph of L layer calls over distinct caches)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import micro_vllm_amd
from micro_vllm_amd import _native as nv
from micro_vllm_amd.core.paged_attention import paged_attn_decode, rotary_tables

def run(bs, hq, hkv, d, ctx, layers=32, splits=0, reps=20, ragged_seed=None, ragged_lo=256):
    dev = "cuda"
    pages = (ctx + 256) // 256 + 1
    nb = bs * pages
    g = torch.Generator(device=dev); g.manual_seed(0)
    kc = torch.empty(layers, nb, 256, hkv, d, dtype=torch.bfloat16, device=dev).normal_(generator=g)
    vc = torch.empty(layers, nb, 256, hkv, d, dtype=torch.bfloat16, device=dev).normal_(generator=g)
    qkv = torch.randn(bs, (hq + 2 * hkv) * d, device=dev).to(torch.bfloat16)
    q = qkv[:, :hq * d].unflatten(1, (hq, d)); k = qkv[:, hq * d:(hq + hkv) * d].unflatten(1, (hkv, d)); v = qkv[:, (hq + hkv) * d:].unflatten(1, (hkv, d))
    perm = torch.randperm(nb, device=dev).to(torch.int32).reshape(bs, pages)
    sl = torch.full((bs,), ctx, dtype=torch.int32, device=dev)
    if ragged_seed is not None:  # lengths uniform in [ragged_lo, ctx]
        import random
        rng = random.Random(ragged_seed)
        sl = torch.tensor([rng.randint(min(ragged_lo, ctx), ctx) for _ in range(bs)], dtype=torch.int32, device=dev)
    cos, sin = rotary_tables(d, 8192, device=dev)
    out = torch.empty(bs, hq, d, dtype=torch.bfloat16, device=dev)
    L = nv.lib()
    if splits <= 0:
        splits = L.mv_paged_attn_decode_pick_splits(bs, hkv, d, pages * 256)
    ws = torch.zeros(L.mv_paged_attn_decode_workspace_bytes(bs, hq, d, max(splits, 2)) // 4 + 4, dtype=torch.float32, device=dev)
    def one():
        for li in range(layers):
            paged_attn_decode(q, k, v, kc[li], vc[li], sl, perm, cos, sin, num_splits=splits, out=out, workspace=ws)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        one(); side.synchronize()
        nv.check("b", L.mv_graph_begin_capture(side.cuda_stream)); one()
        h = ctypes.c_void_p(); nv.check("e", L.mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
        for _ in range(3): L.mv_graph_launch(h, side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps): L.mv_graph_launch(h, side.cuda_stream)
        e1.record(side); side.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / (reps * layers)
    nbytes = int(sl.sum().item() + bs) * hkv * d * 2 * 2
    tag = ""
    if ragged_seed is not None:
        tag = f" ragged[{ragged_lo},{ctx}] longest/mean {sl.max().item() * bs / sl.sum().item():.2f}"
    print(f"bs={bs} hq={hq} hkv={hkv} d={d} ctx={ctx}{tag} splits={splits}: {t*1e6:8.2f} us/call  {nbytes/t/1e9:8.1f} GB/s", flush=True)

if __name__ == "__main__":
    L = nv.lib()
    for merge in (1, 0):
        L.mv_paged_attn_tune(0, merge)
        print("in-kernel merge" if merge else "second-kernel merge")
        for sp in (8, 16, 32, 64):
            run(1, 32, 8, 128, 4096, splits=sp)
    L.mv_paged_attn_tune(0, 1)
    for sp in (1, 2, 4, 8):
        run(32, 32, 8, 128, 4096, layers=8, splits=sp)
    run(8, 8, 1, 128, 8192, layers=16)
    run(1, 32, 4, 64, 2048)
