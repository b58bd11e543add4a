"""End-to-end decode parity: the HIP runner (eager and hipGraph replay) against the CPU restatement of
core/layer/model_graph.py:192-223 on small random-weight models, over several tokens, with the
engine's position numbering and the product KVCacheManager driving the block table."""
import pytest
import torch

from conftest import assert_close_bf16
from oracle.ref_cache import RefKVCache
from oracle.ref_model import RefRunner

pytestmark = pytest.mark.gpu
DEV = "cuda"


def build(mv, shape, qkv_bias=False, fused=True, n_blocks=24, page=256, max_tokens=2048, max_bs=8):
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES[shape]
    cpu_model = build_synthetic_model(cfg, "cpu", seed=1234, qkv_bias=qkv_bias, norm_jitter=0.1)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=1234, qkv_bias=qkv_bias, norm_jitter=0.1).to(DEV)
    runner = ModelGraphRunner(gpu_model, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.head_dim,
                              cfg.num_key_value_heads, cfg.hidden_size, cfg.intermediate_size, DEV,
                              max_batch_size=max_bs, fused=fused)
    mgr = KVCacheManager(n_blocks, page, cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim,
                         dtype=torch.bfloat16, device=DEV, max_tokens=max_tokens, max_batch_size=max_bs)
    ref = RefRunner(cpu_model, cfg.head_dim, world=1)
    ref_cache = RefKVCache(n_blocks, page, cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim,
                           max_tokens=max_tokens, max_batch_size=max_bs, strict_reference=False)
    return cfg, runner, mgr, ref, ref_cache


def prefill_random_kv(mgr, ref_cache, seq_lens, seed):
    """Stand-in for a prefill: allocate pages and fill the used slots with random K/V, identically
    on both sides."""
    g = torch.Generator().manual_seed(seed)
    for sid, L in enumerate(seq_lens):
        ok, slots = mgr.alloc(sid, L)
        assert ok
        ok2, slots2 = ref_cache.alloc(sid, L)
        assert slots.tolist() == slots2
        for li in range(mgr.n_layers):
            kc, vc = ref_cache.k_caches[li], ref_cache.v_caches[li]
            k = torch.randn(L, kc.shape[2], kc.shape[3], generator=g).to(torch.bfloat16)
            v = torch.randn(L, kc.shape[2], kc.shape[3], generator=g).to(torch.bfloat16)
            for i, s in enumerate(slots2):
                kc[s // kc.shape[1], s % kc.shape[1]] = k[i]
                vc[s // kc.shape[1], s % kc.shape[1]] = v[i]
            mgr.k_caches[li].copy_(kc)
            mgr.v_caches[li].copy_(vc)


def step_both(cfg, runner, mgr, ref, ref_cache, seq_ids, positions, ids, use_graph):
    """One decode step as core/engine.py:241-272 drives it: append (deduped), cache_batch_data,
    forward."""
    seen = set()
    for sid in seq_ids:
        if sid not in seen:
            mgr.append(sid)
            ref_cache.append(sid)
            seen.add(sid)
    bt, sl = mgr.cache_batch_data(seq_ids, positions)
    rbt, rsl = ref_cache.cache_batch_data(seq_ids, positions)
    assert torch.equal(bt.cpu(), rbt) and torch.equal(sl.cpu(), rsl)   # bit exact indexing
    bs = len(seq_ids)
    if use_graph:
        logits = runner.forward(ids.to(DEV), mgr, bs)
    else:
        runner._input_ids[:bs] = ids.to(DEV)
        logits = runner.decode(runner._input_ids[:bs], bs, mgr)
    want = ref.decode(ids, [(ref_cache.k_caches, ref_cache.v_caches)], rsl, rbt)
    return logits.float().cpu(), want.float()


def FREE_RUN_ATOL(want):
    """One bf16 unit in the last place at the magnitude of the largest logit."""
    return 2.0 ** -7 * float(want.abs().max())


CASES = [("tiny-gqa", False, True), ("tiny-gqa", False, False), ("tiny-mha", True, True),
         ("tiny-mha", True, False), ("tiny-d128", False, True),
         ("tiny-gqa7", True, True), ("tiny-gqa6-d128", False, True), ("tiny-gqa6-d128", False, False)]


@pytest.mark.parametrize("shape,bias,fused", CASES)
def test_each_layer_and_the_head_on_identical_inputs(mv, shape, bias, fused):
    """Teacher-forced parity: every block gets the ORACLE's hidden state as input, so what is
    compared is one block's arithmetic (norm, QKV, RoPE + append + paged attention, O, add+norm,
    gate/up, SwiGLU, down, residual) and not the amplification of earlier rounding flips.
    Tolerance: the path's 1e-3 plus one bf16 unit in the last place (outputs are bf16 on both
    sides).  The LM head is compared on its UN-ROUNDED fp32 logits to 1e-3 absolute."""
    cfg, runner, mgr, ref, ref_cache = build(mv, shape, qkv_bias=bias, fused=fused)
    seq_lens = [5, 255, 300]
    prefill_random_kv(mgr, ref_cache, seq_lens, seed=7)
    pos = [L + 1 for L in seq_lens]
    for sid in range(3):
        mgr.append(sid)
        ref_cache.append(sid)
    bt, sl = mgr.cache_batch_data([0, 1, 2], pos)
    rbt, rsl = ref_cache.cache_batch_data([0, 1, 2], pos)
    assert torch.equal(bt.cpu(), rbt) and torch.equal(sl.cpu(), rsl)
    ids = torch.tensor([3, cfg.vocab_size - 1, 77])
    trace = []
    ref.decode(ids, [(ref_cache.k_caches, ref_cache.v_caches)], rsl, rbt, trace=trace)
    exact = []
    for li in range(cfg.num_hidden_layers):
        runner._hidden0[:3] = trace[li].to(DEV)
        out = runner._layer(runner._hidden0[:3], li, 3, mgr)
        assert_close_bf16(out, trace[li + 1], atol=1e-3, rtol=2 ** -7, what=f"{shape} layer {li}")
        exact.append(float((out.cpu() == trace[li + 1]).float().mean()))
        # the appended key/value rows of this layer
        assert_close_bf16(mgr.k_caches[li], ref_cache.k_caches[li], atol=1e-3, rtol=2 ** -7)
        assert_close_bf16(mgr.v_caches[li], ref_cache.v_caches[li], atol=1e-3, rtol=2 ** -7)
    assert min(exact) > 0.7, exact   # the rest are single-ulp flips from fp32 summation order
    runner._hidden0[:3] = trace[-1].to(DEV)
    f32 = torch.empty(3, cfg.vocab_size, dtype=torch.float32, device=DEV)
    runner._head(runner._hidden0[:3], 3, out=f32)
    want32 = ref.head(trace[-1], fp32=True)
    assert (f32.cpu() - want32).abs().max().item() <= 1e-3
    assert (f32.cpu() - want32).abs().max().item() <= 1e-4   # in fact fp32 ordering noise only
    logits = runner._head(runner._hidden0[:3], 3)
    assert_close_bf16(logits, ref.head(trace[-1]), atol=1e-5, rtol=2 ** -7, what="bf16 logits")


@pytest.mark.parametrize("shape,bias,fused", CASES)
def test_free_running_decode_vs_oracle(mv, shape, bias, fused):
    """Several tokens end to end, nothing teacher-forced.  bf16 logits are quantised in steps of
    up to 2^-7 |x| (7.8e-3 at |x| in [1,2)), and one flipped rounding early on perturbs every
    later value, so per-element agreement to 1e-3 is not defined for rounded logits.  Checked:
    the greedy token is identical, the mean absolute error stays below 2e-3 (measured 0-1e-3), and no
    logit is off by more than one bf16 unit at the magnitude of the largest logit."""
    cfg, runner, mgr, ref, ref_cache = build(mv, shape, qkv_bias=bias, fused=fused)
    seq_lens = [5, 255, 300]
    prefill_random_kv(mgr, ref_cache, seq_lens, seed=7)
    g = torch.Generator().manual_seed(0)
    pos = [L + 1 for L in seq_lens]      # the engine's first decode position (core/sequence.py:42)
    for step in range(3):
        ids = torch.randint(0, cfg.vocab_size, (3,), generator=g)
        got, want = step_both(cfg, runner, mgr, ref, ref_cache, [0, 1, 2], pos, ids, use_graph=False)
        assert_close_bf16(got, want, atol=FREE_RUN_ATOL(want), rtol=2 ** -7,
                          what=f"{shape} fused={fused} step {step} logits")
        assert (got - want).abs().mean().item() <= 2e-3
        assert (got.argmax(-1) == want.argmax(-1)).all()
        pos = [p + 1 for p in pos]


def test_fused_and_unfused_paths_are_bit_identical(mv):
    cfg, r1, m1, _, rc1 = build(mv, "tiny-gqa", fused=True)
    _, r2, m2, _, rc2 = build(mv, "tiny-gqa", fused=False)
    prefill_random_kv(m1, rc1, [40, 260], seed=3)
    prefill_random_kv(m2, rc2, [40, 260], seed=3)
    ids = torch.tensor([11, 400])
    for m, r in ((m1, r1), (m2, r2)):
        for sid in (0, 1):
            m.append(sid)
        m.cache_batch_data([0, 1], [41, 261])
        r._input_ids[:2] = ids.to(DEV)
    a = r1.decode(r1._input_ids[:2], 2, m1).clone()
    b = r2.decode(r2._input_ids[:2], 2, m2).clone()
    assert torch.equal(a, b)


@pytest.mark.parametrize("bs", [1, 2, 4, 8])
def test_graph_replay_equals_eager_and_oracle(mv, bs):
    """Captured once, replayed over changing block tables / positions / token ids."""
    cfg, runner, mgr, ref, ref_cache = build(mv, "tiny-gqa", n_blocks=40)
    seq_lens = [3, 250, 257, 31, 64, 500, 9, 129][:bs]
    prefill_random_kv(mgr, ref_cache, seq_lens, seed=bs)
    runner.capture(mgr, batch_sizes=[bs])
    assert bs in runner._graphs
    g = torch.Generator().manual_seed(bs)
    pos = [L + 1 for L in seq_lens]
    sids = list(range(bs))
    for step in range(4):
        ids = torch.randint(0, cfg.vocab_size, (bs,), generator=g)
        got, want = step_both(cfg, runner, mgr, ref, ref_cache, sids, pos, ids, use_graph=True)
        assert_close_bf16(got, want, atol=FREE_RUN_ATOL(want), rtol=2 ** -7, what=f"graph bs={bs} step {step}")
        assert (got - want).abs().mean().item() <= 2e-3
        assert (got.argmax(-1) == want.argmax(-1)).all()
        pos = [p + 1 for p in pos]


def test_uneven_rows_replay_the_finer_split_graph(mv):
    """Rows of very different lengths replay the second graph of their batch size (finer
    attention split), rows of similar lengths the first; both agree with the oracle.  The rule
    that decides when a finer split exists looks at the CU count, which no test-sized batch
    reaches, so the test supplies its own."""
    cfg, runner, mgr, ref, ref_cache = build(mv, "tiny-gqa", n_blocks=40)
    runner._splits_uneven = lambda bs, even: even * 2
    seq_lens = [3, 500, 9, 31]
    prefill_random_kv(mgr, ref_cache, seq_lens, seed=5)
    runner.capture(mgr, batch_sizes=[4])
    assert 4 in runner._graphs and 4 in runner._graphs_uneven
    replayed = []
    for graphs in (runner._graphs, runner._graphs_uneven):
        g, inner = graphs[4], graphs[4].replay
        g.replay = (lambda inner=inner, graphs=graphs: (replayed.append(graphs), inner())[1])
    gen = torch.Generator().manual_seed(4)
    pos = [L + 1 for L in seq_lens]
    for step in range(3):
        ids = torch.randint(0, cfg.vocab_size, (4,), generator=gen)
        got, want = step_both(cfg, runner, mgr, ref, ref_cache, [0, 1, 2, 3], pos, ids, use_graph=True)
        assert replayed[-1] is runner._graphs_uneven
        assert_close_bf16(got, want, atol=FREE_RUN_ATOL(want), rtol=2 ** -7, what=f"uneven step {step}")
        assert (got.argmax(-1) == want.argmax(-1)).all()
        pos = [p + 1 for p in pos]
    # the same sequence in every row: equal lengths, first graph
    ids = torch.full((4,), 7)
    got, want = step_both(cfg, runner, mgr, ref, ref_cache, [1, 1, 1, 1], [pos[1]] * 4, ids, use_graph=True)
    assert replayed[-1] is runner._graphs
    assert_close_bf16(got, want, atol=FREE_RUN_ATOL(want), rtol=2 ** -7, what="even step")


def test_duplicate_padded_batch(mv):
    """bs=4 graph fed 3 live sequences + 1 duplicate, as core/scheduler.py:88-92 pads."""
    cfg, runner, mgr, ref, ref_cache = build(mv, "tiny-gqa")
    prefill_random_kv(mgr, ref_cache, [10, 20, 30], seed=2)
    runner.capture(mgr, batch_sizes=[4])
    ids = torch.tensor([1, 2, 3, 1])
    got, want = step_both(cfg, runner, mgr, ref, ref_cache, [0, 1, 2, 0], [11, 21, 31, 11], ids, True)
    assert_close_bf16(got, want, atol=FREE_RUN_ATOL(want), rtol=2 ** -7)
    assert torch.equal(got[0], got[3])


@pytest.mark.parametrize("bs,shape,bias", [(5, "tiny-gqa", False), (8, "tiny-gqa", False), (8, "tiny-mha", True),
                                           (7, "tiny-gqa7", True), (8, "tiny-d128", False)])
def test_batched_path_with_norms_on_the_slab_reduction_equals_the_per_kernel_path(mv, bs, shape, bias):
    """bs > 4: the decode step whose O and down projections carry the following norm and whose
    attention adds up the QKV projection's slabs, bias included (`_decode_batched`), against the same
    step built from `_layer` + `_head` (separate norm kernels, QKV rows stored by the projection):
    same logits, bit for bit, and the same KV written."""
    from micro_vllm_amd.kernel.linear import embedding

    cfg, r1, m1, _, rc1 = build(mv, shape, qkv_bias=bias, n_blocks=40)
    _, r2, m2, _, rc2 = build(mv, shape, qkv_bias=bias, n_blocks=40)
    assert r1.qkv_slabs
    seq_lens = [3, 250, 257, 31, 64, 500, 9, 129][:bs]
    prefill_random_kv(m1, rc1, seq_lens, seed=7)
    prefill_random_kv(m2, rc2, seq_lens, seed=7)
    ids = torch.randint(0, cfg.vocab_size, (bs,), generator=torch.Generator().manual_seed(1))
    for m, r in ((m1, r1), (m2, r2)):
        for sid in range(bs):
            m.append(sid)
        m.cache_batch_data(list(range(bs)), [L + 1 for L in seq_lens])
        r._input_ids[:bs] = ids.to(DEV)
    assert r1.fused and r1.world_size == 1
    a = r1.decode(r1._input_ids[:bs], bs, m1).clone()
    h = embedding(r2.model.transformer.wte.weight, r2._input_ids[:bs], r2._hidden0[:bs])
    for layer_idx in range(r2.num_layers):
        h = r2._layer(h, layer_idx, bs, m2)
    b = r2._head(h, bs).clone()
    assert torch.equal(a, b)
    for li in range(r1.num_layers):
        assert torch.equal(m1.k_caches[li], m2.k_caches[li]) and torch.equal(m1.v_caches[li], m2.v_caches[li])


def test_pick_inside_the_captured_step(mv):
    """The decode graph ends with the token pick: rows with temperature 0 carry the argmax, the
    others the draw the stand-alone sampler makes from the same logits, parameters and uniform
    numbers; rewriting the static parameters between replays changes the next replay only."""
    from micro_vllm_amd.core.layer.sampler import sample

    bs = 4
    cfg, runner, mgr, ref, ref_cache = build(mv, "tiny-gqa", n_blocks=24)
    prefill_random_kv(mgr, ref_cache, [30, 100, 257, 5], seed=4)
    runner.capture(mgr, batch_sizes=[bs])
    ids = torch.tensor([1, 2, 3, 4])
    for sid in range(bs):
        mgr.append(sid)
    mgr.cache_batch_data(list(range(bs)), [31, 101, 258, 6])
    # all greedy (the buffers' initial state)
    logits = runner.forward(ids.to(DEV), mgr, bs)
    assert torch.equal(runner.picked_ids(logits), runner.greedy_ids(logits))
    # rows 1 and 3 draw
    runner.set_sampling([0.0, 0.9, 0.0, 1.5], [1.0, 0.8, 1.0, 0.95])
    gen = torch.Generator(device=DEV).manual_seed(3)
    seen = set()
    for _ in range(6):
        runner.draw_uniforms(bs, generator=gen)
        logits = runner.forward(ids.to(DEV), mgr, bs)     # same inputs: the new key lands on the same slot
        picked = runner.picked_ids(logits).clone()
        want = sample(logits, runner._samp_temp[:bs], runner._samp_topp[:bs], runner.decode_top_k,
                      uniforms=runner._samp_u[:bs].clone())
        assert torch.equal(picked, want)
        g = runner.greedy_ids(logits)
        assert picked[0] == g[0] and picked[2] == g[2]
        seen.add(int(picked[3]))
    assert len(seen) > 1, "six draws at temperature 1.5 over 50 candidates never differed"
    runner.set_sampling([], [])
    logits = runner.forward(ids.to(DEV), mgr, bs)
    assert torch.equal(runner.picked_ids(logits), runner.greedy_ids(logits))
