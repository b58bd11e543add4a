"""Batch 32 at the runner and at the engine level (continuous batching with block 256: prefill
batches and padded decode batches interleaved), against the CPU oracle: the batch size of the
"bs=32 mixed prefill+decode" configuration, which the other runner / engine tests (up to 8 rows) do
not reach."""
import pytest
import torch

from conftest import assert_close_bf16
from test_gpu_decode import FREE_RUN_ATOL, build, prefill_random_kv, step_both
from test_gpu_engine import Tok, _oracle_generate

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_runner_step_at_batch_32_with_duplicate_padding(mv):
    """Graph of 32 rows fed 20 live sequences + 12 duplicates, the way core/scheduler.py:85-92 pads a
    decode group to the next captured size; three steps, positions advancing."""
    cfg, runner, mgr, ref, ref_cache = build(mv, "tiny-gqa", n_blocks=64, max_bs=32)
    seq_lens = [3, 250, 257, 31, 64, 500, 9, 129, 255, 256, 300, 17, 1, 77, 410, 200, 511, 90, 33, 150]
    prefill_random_kv(mgr, ref_cache, seq_lens, seed=32)
    runner.capture(mgr, batch_sizes=[32])
    assert 32 in runner._graphs
    sids = list(range(20)) + [i % 20 for i in range(12)]
    g = torch.Generator().manual_seed(32)
    pos = [L + 1 for L in seq_lens]
    for step in range(3):
        live = torch.randint(0, cfg.vocab_size, (20,), generator=g)
        ids = torch.stack([live[s] for s in sids])
        got, want = step_both(cfg, runner, mgr, ref, ref_cache, sids, [pos[s] for s in sids], ids, use_graph=True)
        assert_close_bf16(got, want, atol=FREE_RUN_ATOL(want), rtol=2 ** -7, what=f"bs=32 step {step}")
        assert (got - want).abs().mean().item() <= 2e-3
        assert (got.argmax(-1) == want.argmax(-1)).all()
        for row, s in enumerate(sids[20:], start=20):
            assert torch.equal(got[row], got[s])            # a duplicate row repeats its original
        pos = [p + 1 for p in pos]
    # the batched step (norms on the slab reduction) is what ran: it equals the per-kernel step
    assert runner.fused and runner.world_size == 1


def test_engine_continuous_batching_at_max_batch_32(mv):
    """28 requests of mixed lengths through an engine with max_batch_size 32 and graphs up to 32:
    prefill batches by length bucket, then decode groups padded to 1..32 rows, sequences finishing
    at different times.  Same greedy tokens as the loop that drives the oracle with the same policy."""
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.core.inference_context import BatchInferenceContext
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-gqa"]
    tok = Tok(cfg.vocab_size)
    cpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1).to(DEV)
    # decode groups are sequences at the same position: 20 prompts of one length (a group of 20,
    # padded to 32 rows), 5 of another (padded to 8), 3 of a third (padded to 4)
    prompts = [(chr(97 + i % 26) * 200)[:n - 2] + f"{i:02d}"
               for i, n in enumerate([30] * 20 + [70] * 5 + [150] * 3)]
    max_new = 5
    sizes = [1, 2, 4, 8, 16, 32]
    eng = InferenceEngine(model=gpu_model, tokenizer=tok, max_batch_size=32, n_blocks=80, max_tokens=1024,
                          graph_batch_sizes=sizes, prefill_timeout=0.0)
    for p in prompts:
        eng.add_request(p, max_tokens=max_new, temperature=0.0)
    by_id = {s.seq_id: s for s in eng.scheduler.waiting_queue}
    kinds = []
    for _ in range(2000):
        batch, kind = eng.get_next_batch()
        if not batch:
            break
        kinds.append((kind, len(batch)))
        ctx = BatchInferenceContext(len(batch), kind, batch)
        assert eng.step(ctx)
        eng.update_sequences(ctx.sequences)
    got = {s.prompt: s.output_ids for s in by_id.values()}
    want, _ = _oracle_generate(cfg, cpu_model, tok, prompts, max_new, 80, 1024, 32, graph_sizes=sizes)
    assert any(k == "decode" and n >= 16 for k, n in kinds), kinds     # large padded decode batches ran
    assert sum(1 for k, _ in kinds if k == "prefill") >= 2
    agree = 0
    for p in prompts:
        assert 1 <= len(got[p]) <= max_new
        assert got[p][0] == want[p][0], f"first token differs for {p!r}: {got[p]} vs {want[p]}"
        agree += int(got[p] == want[p])
    # greedy continuations may part legitimately at a near-tie of the top two logits
    assert agree >= 24, (agree, {p: (got[p], want[p]) for p in prompts if got[p] != want[p]})
    assert eng.cache_manager.stats["free"] == 80
