"""The headline configuration's own dimensions (Llama-3-8B: hidden 4096, 32/8 heads of 128, FFN 14336,
vocab 128 256; two blocks instead of 32 so that the CPU restatement finishes in seconds), context
4096 through a shuffled block table: decode logits against the oracle, the captured step against
the eager one, and the fused step against the step built from separate kernels (bit for bit)."""
import dataclasses
import random

import pytest
import torch

from conftest import assert_close_bf16
from oracle.ref_cache import RefKVCache
from oracle.ref_model import RefRunner

pytestmark = pytest.mark.gpu
DEV = "cuda"
CTX = 4096


def _build(bs, fused=True, seed=5):
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = dataclasses.replace(SHAPES["llama3-8b"], num_hidden_layers=2)
    cpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1).to(DEV)
    n_blocks, max_tokens = bs * (CTX // 256 + 2) + 2, CTX + 512
    runner = ModelGraphRunner(gpu_model, 2, cfg.num_attention_heads, cfg.head_dim, cfg.num_key_value_heads,
                              cfg.hidden_size, cfg.intermediate_size, DEV, max_batch_size=bs,
                              max_blocks=max_tokens // 256, fused=fused)
    mgr = KVCacheManager(n_blocks, 256, 2, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16,
                         device=DEV, max_tokens=max_tokens, max_batch_size=bs)
    ref_cache = RefKVCache(n_blocks, 256, 2, cfg.num_key_value_heads, cfg.head_dim, max_tokens=max_tokens,
                           max_batch_size=bs, strict_reference=False)
    order = list(mgr._free)
    random.Random(seed).shuffle(order)           # non-contiguous pages, the same on both sides
    for pool in (mgr._free, ref_cache.free_blocks):
        pool.clear()
        pool.extend(order)
    g = torch.Generator().manual_seed(seed)
    lens = [CTX - 3 * i for i in range(bs)]
    for sid, L in enumerate(lens):
        ok, slots = mgr.alloc(sid, L)
        ok2, slots2 = ref_cache.alloc(sid, L)
        assert ok and ok2 and slots.tolist() == slots2
    for li in range(2):                          # every slot random: used ones matter, the rest must not
        k = torch.randn(ref_cache.k_caches[li].shape, generator=g).to(torch.bfloat16)
        v = torch.randn(ref_cache.v_caches[li].shape, generator=g).to(torch.bfloat16)
        ref_cache.k_caches[li].copy_(k)
        ref_cache.v_caches[li].copy_(v)
        mgr.k_caches[li].copy_(k)
        mgr.v_caches[li].copy_(v)
    return cfg, runner, mgr, RefRunner(cpu_model, cfg.head_dim, world=1, max_blocks=max_tokens // 256), ref_cache, lens


@pytest.mark.parametrize("bs", [1, 8, 32])   # 32: the continuous-batching configuration's batch size
def test_llama3_8b_dimensions_decode_step(mv, bs):
    cfg, runner, mgr, ref, ref_cache, lens = _build(bs)
    sids = list(range(bs))
    ids = torch.randint(0, cfg.vocab_size, (bs,), generator=torch.Generator().manual_seed(bs))
    for sid in sids:
        mgr.append(sid)
        ref_cache.append(sid)
    pos = [L + 1 for L in lens]
    bt, sl = mgr.cache_batch_data(sids, pos)
    rbt, rsl = ref_cache.cache_batch_data(sids, pos)
    assert torch.equal(bt.cpu(), rbt) and torch.equal(sl.cpu(), rsl)            # indexing: bit exact
    runner._input_ids[:bs] = ids.to(DEV)
    kv_before = [c.clone() for c in mgr.k_caches]
    eager = runner.decode(runner._input_ids[:bs], bs, mgr).float().cpu()
    want = ref.decode(ids, [(ref_cache.k_caches, ref_cache.v_caches)], rsl, rbt).float()
    # Elementwise: the path's 1e-3 plus one bf16 unit at the largest logit (logits of this shape are
    # O(1), one unit in the last place is 2^-7).  On average: free-running logits of two CORRECT
    # implementations differ by rounding flips that every further bf16 rounding point amplifies,
    # so the yardstick is the oracle against itself with one legitimate choice changed (P rounded
    # to bf16 before P.V, as a matrix-core kernel does); the device path must stay within a small
    # multiple of that distance, and may only pick another token where the top two logits tie.
    assert_close_bf16(eager, want, atol=1e-3 + 2.0 ** -7 * float(want.abs().max()), rtol=2 ** -7,
                      what=f"8B dims bs={bs}")
    alt = RefRunner(ref.model, cfg.head_dim, world=1, max_blocks=ref_cache.max_seq_blocks, p_bf16=True)
    alt_l = alt.decode(ids, [(ref_cache.k_caches, ref_cache.v_caches)], rsl, rbt).float()
    floor = (alt_l - want).abs().mean().item()
    budget = 4 * floor + 1e-3
    assert (eager - want).abs().mean().item() <= budget, (floor, (eager - want).abs().mean().item())
    for b in range(bs):
        top2 = want[b].topk(2).values
        if float(top2[0] - top2[1]) > 6 * budget:
            assert int(eager[b].argmax()) == int(want[b].argmax())
    # KV: every slot but the appended one is untouched (bit exact); the appended rows come out of
    # this step's own projections (free running), so they are compared like the logits
    for li in range(2):
        for got_c, want_c in ((mgr.k_caches[li].cpu(), ref_cache.k_caches[li]),
                              (mgr.v_caches[li].cpu(), ref_cache.v_caches[li])):
            diff = (got_c != want_c).flatten(2).any(-1)          # [block, slot]
            assert int(diff.sum()) <= bs, "only the appended slots may differ at all"
            for blk, slot in diff.nonzero().tolist():
                hit = [(int(rbt[b, p // 256]), p % 256) for b, p in enumerate(pos)]   # the new key lands at cache_seqlens
                assert (blk, slot) in hit, (blk, slot, hit)
            assert_close_bf16(got_c.float(), want_c.float(), atol=1e-3 + 2.0 ** -6 * float(want_c.float().abs().max()),
                              rtol=2 ** -6, what=f"appended KV layer {li}")
    # captured step == eager step, bit for bit (the append rewrites the same slot with the same bits)
    runner.capture(mgr, batch_sizes=[bs])
    assert bs in runner._graphs
    replay = runner.forward(ids.to(DEV), mgr, bs).float().cpu()
    assert torch.equal(replay, eager)
    first_max = torch.stack([(row == row.max()).nonzero()[0, 0] for row in eager])   # lowest index on ties
    assert torch.equal(runner.greedy_ids(runner._logits[:bs]).cpu(), first_max)
    del kv_before


@pytest.mark.parametrize("bs", [2, 8])
def test_llama3_8b_dimensions_every_stage_on_identical_inputs(mv, bs):
    """The path's stated 1e-3 at the headline widths, in the form in which it is defined: every
    STAGE of a block (norm, QKV projection, RoPE + append + paged attention, O projection, add + norm,
    gate/up projection + SwiGLU, down projection + residual) gets the ORACLE's input, and its output is
    held to 1e-3 plus one bf16 unit in the last place (both sides round to bf16); the LM head on its
    un-rounded fp32 logits to 1e-3 absolute.

    Why per stage and not per block at these widths: with random weights the first block turns a
    residual stream of |x| ~ 0.04 into one of |x| ~ 2, and single-unit rounding flips multiply on the
    way - measured on MI355X: 0.1 % of the attention outputs differ by one unit, 1.6 % of the residual
    sum, 28 % of the SwiGLU outputs, and the block output then differs by 3.7e-3 on average although
    every stage, fed identical inputs, is within a unit.  The whole block is therefore compared on
    average against how far the oracle moves from ITSELF when one legitimate choice changes."""
    from micro_vllm_amd.core.paged_attention import paged_attn_decode
    from micro_vllm_amd.kernel.linear import wgemm
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm_
    from micro_vllm_amd.kernel.rmsnorm_residual import rmsnorm_residual_gemm
    from oracle.ref_attention import flash_attn_with_kvcache_ref
    from oracle.ref_ops import BF16, linear_ref, rmsnorm_ref, rmsnorm_residual_ref, swiglu_ref

    cfg, runner, mgr, ref, ref_cache, lens = _build(bs)
    sids = list(range(bs))
    ids = torch.randint(0, cfg.vocab_size, (bs,), generator=torch.Generator().manual_seed(40 + bs))
    for sid in sids:
        mgr.append(sid)
        ref_cache.append(sid)
    pos = [L + 1 for L in lens]
    bt, sl = mgr.cache_batch_data(sids, pos)
    rbt, rsl = ref_cache.cache_batch_data(sids, pos)
    hq, hkv, D = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    dev = lambda t: t.to(DEV).contiguous()
    empty = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=DEV)
    ws = runner._gemm_ws
    h = ref.model.transformer.wte.weight[ids]
    alt = RefRunner(ref.model, D, world=1, max_blocks=ref_cache.max_seq_blocks, p_bf16=True)
    for li in range(2):
        blk, gblk, w = ref.model.transformer.h[li], runner.model.transformer.h[li], ref.shards[0].layers[li]
        what = f"8B widths bs={bs} block {li}: "
        normed = rmsnorm_ref(h, blk.ln_1.weight, blk.ln_1.eps)
        got = rmsnorm_(dev(h), gblk.ln_1.weight, empty(bs, cfg.hidden_size), blk.ln_1.eps)
        assert_close_bf16(got, normed, atol=1e-3, rtol=2 ** -7, what=what + "norm")
        qkv = linear_ref(normed, w["qkv"], w["qkv_b"])
        got = wgemm(dev(normed), gblk.attn._qkv_w, empty(bs, qkv.shape[1]), bias=gblk.attn._qkv_b, workspace=ws)
        assert_close_bf16(got, qkv, atol=1e-3, rtol=2 ** -7, what=what + "qkv")
        q, k, v = ref._split_qkv(qkv, (bs,))
        kc_before = ref_cache.k_caches[li].clone()
        attn = flash_attn_with_kvcache_ref(q.unsqueeze(1), ref_cache.k_caches[li], ref_cache.v_caches[li],
                                           k.unsqueeze(1), v.unsqueeze(1), ref.cos, ref.sin, rsl, rbt,
                                           causal=True, p_bf16=False).squeeze(1)
        kc, vc = mgr.get(li)
        got = paged_attn_decode(dev(q), dev(k), dev(v), kc, vc, sl, bt, runner.attention._cos_pool,
                                runner.attention._sin_pool)
        assert_close_bf16(got, attn, atol=1e-5, rtol=2 ** -7, what=what + "attention")
        assert torch.equal(kc.cpu(), ref_cache.k_caches[li]) and torch.equal(vc.cpu(), ref_cache.v_caches[li])
        assert not torch.equal(kc_before, ref_cache.k_caches[li])                # the append happened
        o = linear_ref(attn.reshape(bs, -1), w["o"])
        got = wgemm(dev(attn.reshape(bs, -1)), gblk.attn._o, empty(bs, cfg.hidden_size), workspace=ws)
        assert_close_bf16(got, o, atol=1e-3, rtol=2 ** -7, what=what + "o")
        normed2, residual = rmsnorm_residual_ref(o, h, blk.ln_2.weight, blk.ln_2.eps)
        gn, gr = rmsnorm_residual_gemm(dev(o), dev(h), gblk.ln_2.weight, empty(bs, cfg.hidden_size),
                                       empty(bs, cfg.hidden_size), blk.ln_2.eps)
        assert torch.equal(gr.cpu(), residual), what + "residual sum"
        assert_close_bf16(gn, normed2, atol=1e-3, rtol=2 ** -7, what=what + "add+norm")
        act = swiglu_ref(linear_ref(normed2, w["gu"]))
        got = wgemm(dev(normed2), gblk.mlp._gu, empty(bs, act.shape[1]), swiglu=True, workspace=ws)
        assert_close_bf16(got, act, atol=1e-3, rtol=2 ** -7, what=what + "gate/up + swiglu")
        mlp = linear_ref(act, w["d"])
        out = (mlp.float() + residual.float()).to(BF16)
        got = wgemm(dev(act), gblk.mlp._d, empty(bs, cfg.hidden_size), residual=dev(residual), workspace=ws)
        assert_close_bf16(got, out, atol=1e-3, rtol=2 ** -7, what=what + "down + residual")
        # the block as the runner fuses it, on the same block input: on average within a small multiple
        # of the oracle's distance from itself with P rounded to bf16 before P.V
        runner._hidden0[:bs] = dev(h)
        fused_out = runner._layer(runner._hidden0[:bs], li, bs, mgr).float().cpu()
        alt_cache_k = [c.clone() for c in ref_cache.k_caches]
        alt_cache_v = [c.clone() for c in ref_cache.v_caches]
        alt_out = alt.decode_layer(h, li, [(alt_cache_k, alt_cache_v)], rsl, rbt).float()
        floor = (alt_out - out.float()).abs().mean().item()
        assert (fused_out - out.float()).abs().mean().item() <= 4 * floor + 1e-3, (floor, li)
        h = out
    runner._hidden0[:bs] = dev(h)
    f32 = torch.empty(bs, cfg.vocab_size, dtype=torch.float32, device=DEV)
    runner._head(runner._hidden0[:bs], bs, out=f32)
    want32 = ref.head(h, fp32=True)
    err = (f32.cpu() - want32).abs().max().item()
    assert err <= 1e-3, err
    logits = runner._head(runner._hidden0[:bs], bs)
    assert_close_bf16(logits, ref.head(h), atol=1e-3, rtol=2 ** -7, what="bf16 logits")


def test_llama3_8b_dimensions_fused_equals_separate_kernels(mv):
    """Norm-in-prologue / epilogue-fused step, the same with the QKV projection and the attention of
    a block issued as one launch (key/value rows requested before q exists), and the step made of the
    reference's separate ops: one set of bits."""
    outs = []
    for fused, one_launch in ((True, False), (True, True), (False, False)):
        cfg, runner, mgr, _, _, lens = _build(2, fused=fused)
        runner.qkv_attn = one_launch
        ids = torch.tensor([17, cfg.vocab_size - 2])
        for sid in range(2):
            mgr.append(sid)
        mgr.cache_batch_data([0, 1], [L + 1 for L in lens])
        runner._input_ids[:2] = ids.to(DEV)
        outs.append(runner.decode(runner._input_ids[:2], 2, mgr).clone().cpu())
        if one_launch:
            assert any(runner._qkv_attn_ok.values()), "the library refused to fuse the headline head shape"
            torch.cuda.synchronize()
            assert int(runner._qa_status[0]) == 0
        del runner, mgr
        torch.cuda.empty_cache()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


def test_llama3_8b_dimensions_prefill(mv):
    """Prefill of 600 tokens (two pages and a part of a third) at the same dimensions: each block
    teacher-forced on the oracle's input, K/V pages, and the last position's logits free running."""
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_prefill import ModelPrefillRunner
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    torch.set_num_threads(16)
    cfg = dataclasses.replace(SHAPES["llama3-8b"], num_hidden_layers=2)
    cpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1).to(DEV)
    prefill = ModelPrefillRunner(gpu_model, 2, cfg.num_attention_heads, cfg.head_dim, cfg.num_key_value_heads,
                                 cfg.hidden_size, cfg.intermediate_size, DEV, max_batch_size=1)
    mgr = KVCacheManager(6, 256, 2, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16, device=DEV,
                         max_tokens=1024, max_batch_size=1)
    ref_cache = RefKVCache(6, 256, 2, cfg.num_key_value_heads, cfg.head_dim, max_tokens=1024, max_batch_size=1,
                           strict_reference=False)
    for pool in (mgr._free, ref_cache.free_blocks):
        pool.clear()
        pool.extend([4, 1, 5, 0, 2, 3])
    # the matrix-core attention rounds P to bf16: compare with the oracle form that does the same
    ref = RefRunner(cpu_model, cfg.head_dim, world=1, p_bf16=True)
    S = 600
    ids = torch.randint(0, cfg.vocab_size, (1, S), generator=torch.Generator().manual_seed(9))
    assert mgr.alloc(0, S)[0] and ref_cache.alloc(0, S)[0]
    bt, _ = mgr.cache_batch_data([0], [S])
    rbt, _ = ref_cache.cache_batch_data([0], [S])
    assert torch.equal(bt.cpu(), rbt)
    trace = []
    want = ref.prefill(ids, [(ref_cache.k_caches, ref_cache.v_caches)], rbt, trace=trace)[:, -1].float()

    alt_cache = RefKVCache(6, 256, 2, cfg.num_key_value_heads, cfg.head_dim, max_tokens=1024, max_batch_size=1,
                           strict_reference=False)
    alt_cache.free_blocks.clear()
    alt_cache.free_blocks.extend([4, 1, 5, 0, 2, 3])
    alt_cache.alloc(0, S)
    alt = RefRunner(cpu_model, cfg.head_dim, world=1, p_bf16=False)
    alt_trace = []
    alt_l = alt.prefill(ids, [(alt_cache.k_caches, alt_cache.v_caches)], rbt, trace=alt_trace)[:, -1].float()
    floors = [(alt_trace[li + 1].float() - trace[li + 1].float()).abs().mean().item() for li in range(2)]

    lens = mgr._cache_seqlens_buffer[:1].zero_()
    for li in range(2):
        h_in = trace[li].reshape(S, -1).to(DEV)
        out = prefill._prefill_layer(h_in, li, 1, S, mgr, lens, mgr._block_table_buffer[:1])
        w = trace[li + 1].reshape(S, -1).float()
        # bound per element as in the small-model test: the path's 1e-3, one bf16 unit at the row's
        # largest magnitude and one at the element's own addends; the mean far below
        err = (out.float().cpu() - w).abs()
        hin = trace[li].reshape(S, -1).float().abs()
        tol = 1e-3 + 2 ** -7 * w.abs().amax(dim=-1, keepdim=True) + 2 ** -7 * torch.maximum(w.abs(), hin)
        over = err > tol
        assert not bool(over.any()), (
            f"prefill layer {li}: max err {err.max().item():.3e}, {int(over.sum())} of {err.numel()} over the bound, "
            f"worst ratio {(err / tol).max().item():.2f}")
        # on average: within a small multiple of what the oracle differs from ITSELF on this block
        # when P is kept in fp32 instead of being rounded to bf16 (hidden states of this width are
        # O(1..10), an absolute figure would say nothing)
        assert err.mean().item() <= 4 * floors[li] + 1e-3, \
            f"prefill layer {li}: mean err {err.mean().item():.3e}, oracle-vs-oracle {floors[li]:.3e}"
        kmax = float(ref_cache.k_caches[li].float().abs().max())
        assert_close_bf16(mgr.k_caches[li], ref_cache.k_caches[li], atol=1e-3 + 2 ** -7 * kmax, rtol=2 ** -7)
        assert_close_bf16(mgr.v_caches[li], ref_cache.v_caches[li], atol=1e-3 + 2 ** -8 * kmax, rtol=2 ** -7)

    got = prefill.forward(ids.to(DEV), mgr, 1, last_token_only=True)[:, -1].float().cpu()
    floor = (alt_l - want).abs().mean().item()
    budget = 4 * floor + 1e-3
    assert (got - want).abs().mean().item() <= budget, (floor, (got - want).abs().mean().item())
    top2 = want[0].topk(2).values
    if float(top2[0] - top2[1]) > 6 * budget:
        assert int(got.argmax()) == int(want.argmax())


def test_llama3_8b_dimensions_prefill_block_at_3072_tokens(mv):
    """One block at the same dimensions on 3072 tokens: long enough for the 256-row GEMM form in all its
    widths (QKV with its left-over tiles cut along k inside the launch, gate/up with the fused
    activation, O and down in whole rounds) and for the pipelined attention over twelve pages - the
    forms the 4096-token prefill of the bench line runs - teacher-forced on the oracle's input."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_prefill import ModelPrefillRunner
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    torch.set_num_threads(16)
    cfg = dataclasses.replace(SHAPES["llama3-8b"], num_hidden_layers=1)
    cpu_model = build_synthetic_model(cfg, "cpu", seed=4321, norm_jitter=0.1)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=4321, norm_jitter=0.1).to(DEV)
    prefill = ModelPrefillRunner(gpu_model, 1, cfg.num_attention_heads, cfg.head_dim, cfg.num_key_value_heads,
                                 cfg.hidden_size, cfg.intermediate_size, DEV, max_batch_size=1)
    S, pages = 3072, 14
    order = [7, 1, 10, 0, 13, 4, 3, 11, 6, 12, 2, 9, 5, 8]
    mgr = KVCacheManager(pages, 256, 1, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16, device=DEV,
                         max_tokens=pages * 256, max_batch_size=1)
    ref_cache = RefKVCache(pages, 256, 1, cfg.num_key_value_heads, cfg.head_dim, max_tokens=pages * 256,
                           max_batch_size=1, strict_reference=False)
    for pool in (mgr._free, ref_cache.free_blocks):
        pool.clear()
        pool.extend(order)
    L = nv.lib()
    hid, qkv_dim = cfg.hidden_size, (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
    if torch.cuda.get_device_properties(0).multi_processor_count == 256:
        assert L.mv_gemm_prefill_form(S, qkv_dim, hid) in (1, 2, 3) and L.mv_gemm_prefill_form(S, cfg.intermediate_size, hid) in (1, 2, 3)
        assert L.mv_gemm_prefill_workspace_bytes(S, qkv_dim, hid) == 4096 + 256 * 256 * 1024, "QKV is meant to cut its left-over tiles"
    assert L.mv_paged_attn_prefill_form(1, S, cfg.num_attention_heads, cfg.num_key_value_heads) == 8
    ref = RefRunner(cpu_model, cfg.head_dim, world=1, p_bf16=True)
    ids = torch.randint(0, cfg.vocab_size, (1, S), generator=torch.Generator().manual_seed(19))
    assert mgr.alloc(0, S)[0] and ref_cache.alloc(0, S)[0]
    bt, _ = mgr.cache_batch_data([0], [S])
    rbt, _ = ref_cache.cache_batch_data([0], [S])
    assert torch.equal(bt.cpu(), rbt)
    trace = []
    ref.prefill(ids, [(ref_cache.k_caches, ref_cache.v_caches)], rbt, trace=trace)

    lens = mgr._cache_seqlens_buffer[:1].zero_()
    h_in = trace[0].reshape(S, -1).to(DEV)
    out = prefill._prefill_layer(h_in, 0, 1, S, mgr, lens, mgr._block_table_buffer[:1])
    w = trace[1].reshape(S, -1).float()
    err = (out.float().cpu() - w).abs()
    hin = trace[0].reshape(S, -1).float().abs()
    tol = 1e-3 + 2 ** -7 * w.abs().amax(dim=-1, keepdim=True) + 2 ** -7 * torch.maximum(w.abs(), hin)
    # The bound of the 600-token test, whose 2.5 M elements all keep it.  Among the 12.6 M here a few sit
    # just past it (3, by at most 8 %, when the test was written: outputs well below 1 whose MLP term
    # and residual, each near the row's largest magnitude, cancel - one bf16 unit of that magnitude
    # is then more than the bound allows for the row's largest OUTPUT): no element may pass it by a
    # quarter, and no more than one in a million at all.
    over = err > tol
    assert not bool((err > 1.25 * tol).any()) and int(over.sum()) <= err.numel() // 1_000_000, (
        f"max err {err.max().item():.3e}, {int(over.sum())} of {err.numel()} over the bound, "
        f"worst ratio {(err / tol).max().item():.2f}")
    # on average: a small fraction of a bf16 unit at the outputs' own size (the 600-token test above holds
    # the same quantity against the oracle's own spread between two softmax roundings: 2e-3 of the mean size)
    assert err.mean().item() <= 2 ** -7 * w.abs().mean().item() + 1e-4, (err.mean().item(), w.abs().mean().item())
    kmax = float(ref_cache.k_caches[0].float().abs().max())
    assert_close_bf16(mgr.k_caches[0], ref_cache.k_caches[0], atol=1e-3 + 2 ** -7 * kmax, rtol=2 ** -7)
    assert_close_bf16(mgr.v_caches[0], ref_cache.v_caches[0], atol=1e-3 + 2 ** -8 * kmax, rtol=2 ** -7)
