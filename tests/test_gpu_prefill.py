"""Prefill path on the GPU against the CPU oracle: rotate + store, causal paged attention on the
matrix cores, the keyword-compatible flash_attn_with_kvcache (Sq > 1), and the whole
ModelPrefillRunner (per layer on identical inputs, then end to end, then decode continuing on
the pages the prefill wrote)."""
import pytest
import torch

from conftest import assert_close_bf16
from oracle.ref_attention import flash_attn_with_kvcache_ref
from oracle.ref_cache import RefKVCache
from oracle.ref_model import RefRunner
from oracle.ref_ops import rope_prefill_ref, rotary_tables_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _paged_case(B, S, Hq, Hkv, D, starts, page=256, n_pages=24, seed=0, max_blocks=6):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(B, S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(B, S, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(B, S, Hkv, D, generator=g).to(torch.bfloat16)
    kc = torch.randn(n_pages, page, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(n_pages, page, Hkv, D, generator=g).to(torch.bfloat16)
    perm = torch.randperm(n_pages, generator=g)
    bt = torch.full((B, max_blocks), -1, dtype=torch.int32)
    used = 0
    for b in range(B):
        need = (starts[b] + S + page - 1) // page
        assert need <= max_blocks
        bt[b, :need] = perm[used:used + need].to(torch.int32)
        used += need
    return q, k, v, kc, vc, bt, torch.tensor(starts, dtype=torch.int32)


def test_rope_class_matches_the_bf16_tensor_expression(mv):
    """RoPE.forward against rope_prefill_ref (itself pinned to the reference's RoPE.forward outputs
    in tests/golden/rotary.npz): bit exact - same products, same roundings."""
    from micro_vllm_amd.core.layer.rope import RoPE

    for D, H, S in [(64, 4, 33), (128, 8, 300)]:
        g = torch.Generator().manual_seed(D)
        q = torch.randn(2, S, H, D, generator=g).to(torch.bfloat16)
        k = torch.randn(2, S, 2, D, generator=g).to(torch.bfloat16)
        cos, sin = rotary_tables_ref(D, 512)
        wq, wk = rope_prefill_ref(q, k, cos, sin)
        gq, gk = RoPE().forward(q.to(DEV), k.to(DEV), cos.to(DEV), sin.to(DEV))
        assert torch.equal(gq.cpu(), wq) and torch.equal(gk.cpu(), wk)


@pytest.mark.parametrize("D,Hq,Hkv", [(128, 8, 2), (64, 8, 8), (64, 4, 2), (128, 8, 1)])
def test_rope_store_fills_the_pages_bit_exactly(mv, D, Hq, Hkv):
    from micro_vllm_amd.core.prefill_attention import rope_store_prefill

    B, S = 3, 70
    starts = [0, 250, 13]
    q, k, v, kc, vc, bt, st = _paged_case(B, S, Hq, Hkv, D, starts, seed=D + Hq)
    cos, sin = rotary_tables_ref(D, 2048)
    qkv = torch.cat([q.reshape(B * S, -1), k.reshape(B * S, -1), v.reshape(B * S, -1)], dim=1)
    # oracle: rotation at positions start + i (rope.py arithmetic), then the append of flash-attn's contract
    wq, wk = torch.empty_like(q), torch.empty_like(k)
    want_k, want_v = kc.clone(), vc.clone()
    for b in range(B):
        p0 = starts[b]
        rq, rk = rope_prefill_ref(q[b:b + 1], k[b:b + 1], cos[p0:], sin[p0:])
        wq[b], wk[b] = rq[0], rk[0]
        for i in range(S):
            pg = int(bt[b, (p0 + i) // 256])
            want_k[pg, (p0 + i) % 256] = rk[0, i]
            want_v[pg, (p0 + i) % 256] = v[b, i]
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    got = rope_store_prefill(qkv.to(DEV), B, S, Hq, Hkv, D, kcd, vcd, st.to(DEV), bt.to(DEV),
                             cos.to(DEV), sin.to(DEV)).cpu()
    assert torch.equal(got[:, :Hq * D].reshape(B, S, Hq, D), wq)
    assert torch.equal(got[:, Hq * D:(Hq + Hkv) * D].reshape(B, S, Hkv, D), wk)
    assert torch.equal(kcd.cpu(), want_k) and torch.equal(vcd.cpu(), want_v)


@pytest.mark.parametrize("D,Hq,Hkv", [(128, 8, 2), (128, 4, 4), (64, 8, 1), (64, 4, 2),
                                      # 7, 6, 5 and 3 query heads per kv head: padded to 8 / 4 in the row stack
                                      (128, 14, 2), (64, 6, 1), (128, 5, 1), (64, 6, 2)])
@pytest.mark.parametrize("S,starts", [(1, [0, 5]), (17, [0, 300]), (130, [0, 0]), (300, [256, 1])])
def test_prefill_attention_vs_oracle(mv, D, Hq, Hkv, S, starts):
    """Kernel output against the exact-softmax oracle within 1e-3 + one output ulp + the bound of
    rounding P to bf16 (the kernel feeds P to the matrix cores in bf16, like the package the
    reference calls), and against the oracle that rounds P the same way."""
    from micro_vllm_amd.core.paged_attention import flash_attn_with_kvcache

    B = len(starts)
    q, k, v, kc, vc, bt, st = _paged_case(B, S, Hq, Hkv, D, starts, seed=S + D + Hq)
    kc1, vc1 = kc.clone(), vc.clone()
    exact, bound = flash_attn_with_kvcache_ref(q, kc1, vc1, k, v, None, None, st, bt, causal=True,
                                               p_bf16=False, return_p_bound=True)
    kc2, vc2 = kc.clone(), vc.clone()
    rounded = flash_attn_with_kvcache_ref(q, kc2, vc2, k, v, None, None, st, bt, causal=True, p_bf16=True)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    got = flash_attn_with_kvcache(q.to(DEV), kcd, vcd, k=k.to(DEV), v=v.to(DEV),
                                  cache_seqlens=st.to(DEV), block_table=bt.to(DEV), causal=True).cpu()
    assert torch.equal(kcd.cpu(), kc1) and torch.equal(vcd.cpu(), vc1)       # the append is bit exact
    err = (got.float() - exact.float()).abs()
    tol = 1e-3 + 2 ** -7 * exact.float().abs() + bound
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e}, worst excess {(err - tol).max().item():.3e}"
    assert_close_bf16(got, rounded, atol=4e-3, rtol=2 ** -6, what="vs bf16-P oracle")


def test_prefill_attention_long_chunks_take_256_row_workgroups(mv):
    """Chunks of 512 tokens and more run eight waves per workgroup (one staged K/V tile against 256
    stacked query rows); same bounds as above, non-zero starts included."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core.paged_attention import flash_attn_with_kvcache

    D, Hq, Hkv, S, starts = 64, 8, 4, 2048, [0, 256, 5, 300]
    assert nv.lib().mv_paged_attn_prefill_form(len(starts), S, Hq, Hkv) == 8, "meant to take the 8-wave form"
    assert nv.lib().mv_paged_attn_prefill_form(len(starts), 300, Hq, Hkv) == 4
    q, k, v, kc, vc, bt, st = _paged_case(len(starts), S, Hq, Hkv, D, starts, seed=77, n_pages=44, max_blocks=10)
    kc1, vc1 = kc.clone(), vc.clone()
    exact, bound = flash_attn_with_kvcache_ref(q, kc1, vc1, k, v, None, None, st, bt, causal=True,
                                               p_bf16=False, return_p_bound=True)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    got = flash_attn_with_kvcache(q.to(DEV), kcd, vcd, k=k.to(DEV), v=v.to(DEV),
                                  cache_seqlens=st.to(DEV), block_table=bt.to(DEV), causal=True).cpu()
    assert torch.equal(kcd.cpu(), kc1) and torch.equal(vcd.cpu(), vc1)
    err = (got.float() - exact.float()).abs()
    tol = 1e-3 + 2 ** -7 * exact.float().abs() + bound
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e}, worst excess {(err - tol).max().item():.3e}"


@pytest.mark.parametrize("D,Hq,Hkv,S,starts,page", [
    (128, 8, 2, 2048, [0, 256, 5, 300], 256),                    # four query heads per kv head (the 8B ratio)
    (128, 4, 4, 2048, [0, 7, 64, 100, 300, 1, 31, 255], 256),    # one: 256 positions per workgroup
    (128, 16, 2, 2048, [0, 33], 256),                            # eight: 32 positions per workgroup
    (128, 6, 2, 2048, [3, 0, 129, 64], 256),                     # three, padded to four in the row stack
    (128, 8, 2, 2100, [0, 1, 63, 64, 65, 200, 256], 256),        # a last query block that is partly empty
    (64, 4, 2, 2048, [0, 17, 250, 64, 1, 0, 0, 300], 256),       # heads of 64
    (128, 8, 2, 600, [0, 100, 257], 256),                        # mid-size chunks: few workgroups, ragged last tile
    (64, 4, 4, 513, [5, 0], 256),
    (128, 8, 2, 600, [1500, 0, 700], 256),                       # chunks late in a sequence: long runs of unmasked tiles
    (128, 8, 2, 600, [70, 0], 64),                          # a page per key tile: every tile looks its page up
    (64, 4, 2, 520, [0, 33], 128),
])
def test_prefill_attention_pipelined_form_vs_oracle(mv, D, Hq, Hkv, S, starts, page):
    """The software-pipelined eight-wave kernel (long chunks) at every stack shape it has: exact
    append, output within the same bounds as the plain kernel above.  The first tiles of a sequence
    that starts at 0, the masked tiles on the diagonal (starts that are not multiples of 32 or 64
    put the mask boundary inside a block) and rows past the chunk are all in these cases."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core.paged_attention import flash_attn_with_kvcache

    B = len(starts)
    assert nv.lib().mv_paged_attn_prefill_form(B, S, Hq, Hkv) == 8, "meant to take the 8-wave form"
    need = sum((st + S + page - 1) // page for st in starts)
    q, k, v, kc, vc, bt, st = _paged_case(B, S, Hq, Hkv, D, starts, page=page, seed=S + D + Hq, n_pages=need + 3,
                                          max_blocks=(max(starts) + S + page - 1) // page + 1)
    kc1, vc1 = kc.clone(), vc.clone()
    exact, bound = flash_attn_with_kvcache_ref(q, kc1, vc1, k, v, None, None, st, bt, causal=True,
                                               p_bf16=False, return_p_bound=True)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    got = flash_attn_with_kvcache(q.to(DEV), kcd, vcd, k=k.to(DEV), v=v.to(DEV),
                                  cache_seqlens=st.to(DEV), block_table=bt.to(DEV), causal=True).cpu()
    assert torch.equal(kcd.cpu(), kc1) and torch.equal(vcd.cpu(), vc1)
    err = (got.float() - exact.float()).abs()
    tol = 1e-3 + 2 ** -7 * exact.float().abs() + bound
    assert bool((err <= tol).all()), f"max err {err.max().item():.3e}, worst excess {(err - tol).max().item():.3e}"


_LAGGING_MAX_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import micro_vllm_amd
from test_gpu_prefill import _paged_case
from oracle.ref_attention import flash_attn_with_kvcache_ref
from micro_vllm_amd.core.paged_attention import flash_attn_with_kvcache
D, Hq, Hkv, S, starts = 128, 8, 2, 1100, [0, 200]
q, k, v, kc, vc, bt, st = _paged_case(2, S, Hq, Hkv, D, starts, seed=5, n_pages=12, max_blocks=6)
# scores that keep growing along the keys: every block raises the maximum a little
ramp = torch.linspace(0.5, 3.0, S).reshape(1, S, 1, 1)
k = (k.float() * ramp).to(torch.bfloat16)
kc1, vc1 = kc.clone(), vc.clone()
exact, bound = flash_attn_with_kvcache_ref(q, kc1, vc1, k, v, None, None, st, bt, causal=True, p_bf16=False,
                                           return_p_bound=True)
got = flash_attn_with_kvcache(q.cuda(), kc.cuda(), vc.cuda(), k=k.cuda(), v=v.cuda(), cache_seqlens=st.cuda(),
                              block_table=bt.cuda(), causal=True).cpu()
err = (got.float() - exact.float()).abs()
# a weight may now be as large as 2^8 when it is rounded: the rounding bound grows with it, by no
# more than the factor by which the largest weight may exceed 1 in the normalised sum (it does not:
# numerator and normaliser carry the same scale) - the same bound is asked for
tol = 1e-3 + 2 ** -7 * exact.float().abs() + bound
print("MAXERR", float(err.max()), "EXCESS", float((err - tol).max()))
assert bool((err <= tol).all())
"""


def test_prefill_attention_with_a_lagging_maximum_stays_within_the_same_bounds(mv):
    """MVLLM_PREFILL_ATTN_RAISE=8 (read once per process: a child process) - the running maximum
    follows only jumps of more than 2^8 - on keys whose scores keep growing along the sequence:
    same bound against the exact softmax as the default."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _LAGGING_MAX_CHILD, root], capture_output=True, text=True, timeout=280,
                       env=dict(os.environ, MVLLM_PREFILL_ATTN_RAISE="8"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "MAXERR" in r.stdout


def _build(mv, shape, qkv_bias=False, n_blocks=24, max_bs=4, p_bf16=False):
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner
    from micro_vllm_amd.core.layer.model_prefill import ModelPrefillRunner
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES[shape]
    cpu_model = build_synthetic_model(cfg, "cpu", seed=1234, qkv_bias=qkv_bias, norm_jitter=0.1)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=1234, qkv_bias=qkv_bias, norm_jitter=0.1).to(DEV)
    args = (gpu_model, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.head_dim,
            cfg.num_key_value_heads, cfg.hidden_size, cfg.intermediate_size, DEV)
    decode = ModelGraphRunner(*args, max_batch_size=max_bs)
    prefill = ModelPrefillRunner(*args, max_batch_size=max_bs)   # shares the relaid-out weights
    mgr = KVCacheManager(n_blocks, 256, cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim,
                         dtype=torch.bfloat16, device=DEV, max_tokens=2048, max_batch_size=max_bs)
    ref = RefRunner(cpu_model, cfg.head_dim, world=1, p_bf16=p_bf16)
    ref_cache = RefKVCache(n_blocks, 256, cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim,
                           max_tokens=2048, max_batch_size=max_bs, strict_reference=False)
    return cfg, decode, prefill, mgr, ref, ref_cache


@pytest.mark.parametrize("shape,bias,S", [("tiny-gqa", False, 40), ("tiny-mha", True, 150), ("tiny-d128", False, 290),
                                          ("tiny-gqa7", True, 70), ("tiny-gqa6-d128", False, 130)])
def test_prefill_runner_per_layer_and_end_to_end(mv, shape, bias, S):
    # the matrix-core kernel rounds P to bf16 like the package the reference calls: compare with
    # the oracle form that does the same
    cfg, _, prefill, mgr, ref, ref_cache = _build(mv, shape, qkv_bias=bias, p_bf16=True)
    B = 2
    g = torch.Generator().manual_seed(S)
    ids = torch.randint(0, cfg.vocab_size, (B, S), generator=g)
    for sid in range(B):
        ok, _ = mgr.alloc(sid, S)
        ok2, _ = ref_cache.alloc(sid, S)
        assert ok and ok2
    bt, sl = mgr.cache_batch_data(list(range(B)), [S] * B)
    rbt, _ = ref_cache.cache_batch_data(list(range(B)), [S] * B)
    assert torch.equal(bt.cpu(), rbt)
    trace = []
    want = ref.prefill(ids, [(ref_cache.k_caches, ref_cache.v_caches)], rbt, trace=trace)

    # teacher-forced: each block on the oracle's input (one block's arithmetic, no amplification)
    lens = mgr._cache_seqlens_buffer[:B].zero_()
    exact = []
    for li in range(cfg.num_hidden_layers):
        h_in = trace[li].reshape(B * S, -1).to(DEV)
        out = prefill._prefill_layer(h_in, li, B, S, mgr, lens, mgr._block_table_buffer[:B])
        w = trace[li + 1].reshape(B * S, -1)
        # The block output is bf16(mlp + bf16(attn + h_in)): a one-ulp flip of an addend shows at
        # ITS magnitude even where the sum cancels, and rounding P to bf16 (realised in a
        # different order on each side) adds noise of the same size.  Bound: the path's 1e-3
        # plus one bf16 ulp at the row's largest magnitude plus one at the element's own addends;
        # the mean error must stay far below.
        err = (out.float().cpu() - w.float()).abs()
        hin = trace[li].reshape(B * S, -1).float().abs()
        tol = (1e-3 + 2 ** -7 * w.float().abs().amax(dim=-1, keepdim=True)
               + 2 ** -7 * torch.maximum(w.float().abs(), hin))
        assert bool((err <= tol).all()), f"{shape} prefill layer {li}: max err {err.max().item():.3e}"
        assert err.mean().item() <= 5e-4, f"{shape} prefill layer {li}: mean err {err.mean().item():.3e}"
        exact.append(float((out.cpu() == w).float().mean()))
        # a one-ulp flip of k before the rotation moves both rotated halves: allow one ulp at the
        # head's scale
        kmax = float(ref_cache.k_caches[li].float().abs().max())
        assert_close_bf16(mgr.k_caches[li], ref_cache.k_caches[li], atol=1e-3 + 2 ** -7 * kmax, rtol=2 ** -7)
        assert_close_bf16(mgr.v_caches[li], ref_cache.v_caches[li], atol=1e-3, rtol=2 ** -7)
    assert min(exact) > 0.4, exact   # the rest: single-ulp flips (summation order, P rounding order)

    # free-running: same argmax at every position, small mean error, bounded worst case
    logits = prefill.forward(ids.to(DEV), mgr, B).float().cpu()
    assert logits.shape == (B, S, cfg.vocab_size)
    wantf = want.float()
    assert (logits - wantf).abs().mean().item() <= 4e-3
    top = wantf.topk(2, dim=-1).values
    clear = (top[..., 0] - top[..., 1]) > 4 * 2.0 ** -7 * wantf.abs().amax(dim=-1)
    assert bool((logits.argmax(-1) == wantf.argmax(-1))[clear].all())
    last = prefill.forward(ids.to(DEV), mgr, B, last_token_only=True).float().cpu()
    assert last.shape == (B, 1, cfg.vocab_size)
    assert (last[:, 0] - logits[:, -1]).abs().max().item() <= 2.0 ** -6 * float(wantf.abs().max())


def test_decode_continues_on_the_pages_prefill_wrote(mv):
    """Prefill S tokens on both sides, then decode with the engine's numbering (position S+1,
    index S never written: zeros in a fresh cache) and compare logits."""
    cfg, decode, prefill, mgr, ref, ref_cache = _build(mv, "tiny-gqa")
    B, S = 2, 60
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, cfg.vocab_size, (B, S), generator=g)
    for sid in range(B):
        mgr.alloc(sid, S)
        ref_cache.alloc(sid, S)
    mgr.cache_batch_data([0, 1], [S, S])
    rbt, _ = ref_cache.cache_batch_data([0, 1], [S, S])
    prefill.forward(ids.to(DEV), mgr, B, last_token_only=True)
    ref.prefill(ids, [(ref_cache.k_caches, ref_cache.v_caches)], rbt)
    nxt = torch.tensor([5, 9])
    for sid in range(B):
        mgr.append(sid)
        ref_cache.append(sid)
    bt, sl = mgr.cache_batch_data([0, 1], [S + 1, S + 1])
    rbt, rsl = ref_cache.cache_batch_data([0, 1], [S + 1, S + 1])
    assert torch.equal(bt.cpu(), rbt) and torch.equal(sl.cpu(), rsl)
    decode._input_ids[:B] = nxt.to(DEV)
    got = decode.decode(decode._input_ids[:B], B, mgr).float().cpu()
    want = ref.decode(nxt, [(ref_cache.k_caches, ref_cache.v_caches)], rsl, rbt).float()
    assert (got - want).abs().mean().item() <= 4e-3
    assert torch.equal(got.argmax(-1), want.argmax(-1))


def test_tokens_past_the_block_table_width_touch_no_other_sequence(mv):
    """start + S beyond the last block-table column (a caller error the engine's intake check
    prevents, reachable through the op-level call): the rows that have a page are stored, the ones
    past the table are dropped - nothing is read from the next row's page ids, nothing of any other
    sequence's pages changes, and attention stays finite."""
    from micro_vllm_amd.core.paged_attention import flash_attn_with_kvcache
    from micro_vllm_amd.core.prefill_attention import rope_store_prefill

    B, S, Hq, Hkv, D, page, W = 2, 100, 8, 2, 128, 256, 2
    g = torch.Generator().manual_seed(3)
    n_pages = 8
    kc = torch.randn(n_pages, page, Hkv, D, generator=g).to(torch.bfloat16)
    vc = torch.randn(n_pages, page, Hkv, D, generator=g).to(torch.bfloat16)
    bt = torch.tensor([[5, 1], [3, 6]], dtype=torch.int32)          # 2 columns: 512 positions per row
    starts = torch.tensor([450, 0], dtype=torch.int32)              # row 0 runs 38 tokens past its table
    qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, generator=g).to(torch.bfloat16)
    cos, sin = rotary_tables_ref(D, 2048)
    kcd, vcd = kc.to(DEV), vc.to(DEV)
    rope_store_prefill(qkv.to(DEV), B, S, Hq, Hkv, D, kcd, vcd, starts.to(DEV), bt.to(DEV), cos.to(DEV), sin.to(DEV))
    torch.cuda.synchronize()
    changed = (kcd.cpu() != kc).flatten(2).any(-1) | (vcd.cpu() != vc).flatten(2).any(-1)    # [page, slot]
    allowed = torch.zeros_like(changed)
    allowed[1, 450 - 256:] = True           # row 0: positions 450..511 on its second page
    allowed[3, :100] = True                 # row 1: positions 0..99 on its first page
    assert not bool((changed & ~allowed).any()), (changed & ~allowed).nonzero().tolist()
    assert bool(changed[1, 450 - 256:].all()) and bool(changed[3, :100].all())
    q = torch.randn(B, S, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(B, S, Hkv, D, generator=g).to(torch.bfloat16)
    out = flash_attn_with_kvcache(q.to(DEV), kcd, vcd, k=k.to(DEV), v=k.to(DEV), cache_seqlens=starts.to(DEV),
                                  block_table=bt.to(DEV), causal=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())
    # row 1 is a well-formed call: it still matches the oracle
    kc1, vc1 = kcd.cpu().clone(), vcd.cpu().clone()
    kc0, vc0 = kc1.clone(), vc1.clone()
    want = flash_attn_with_kvcache_ref(q[1:], kc0, vc0, k[1:], k[1:], None, None, starts[1:], bt[1:], causal=True,
                                       p_bf16=True)
    assert_close_bf16(out[1:].cpu(), want, atol=4e-3, rtol=2 ** -6, what="well-formed row")
