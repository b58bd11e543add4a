"""InferenceEngine end to end on the GPU (stub tokenizer, tiny random-weight model, greedy):
scheduler -> prefill -> decode with the reference's position numbering, against a loop that drives
the CPU oracle (RefRunner + RefKVCache) with the same scheduler decisions."""
import pytest
import torch

from oracle.ref_cache import RefKVCache
from oracle.ref_model import RefRunner

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Tok:
    eos_token_id = 2

    def __init__(self, vocab):
        self.vocab = vocab

    def encode(self, prompt, add_special_tokens=True):
        return [5 + (ord(c) * 7) % (self.vocab - 5) for c in prompt]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)


def _oracle_generate(cfg, cpu_model, tok, prompts, max_new, n_blocks, max_tokens, max_bs,
                     graph_sizes=(1, 2, 4, 8)):
    """The same policy by hand: one prefill batch per scheduler decision, then decode groups."""
    from micro_vllm_amd.core.scheduler import Scheduler
    from micro_vllm_amd.core.sequence import Sequence

    ref = RefRunner(cpu_model, cfg.head_dim, world=1, max_blocks=max(32, (max_tokens + 255) // 256))
    cache = RefKVCache(n_blocks, 256, cfg.num_hidden_layers, cfg.num_key_value_heads, cfg.head_dim,
                       max_tokens=max_tokens, max_batch_size=max_bs, strict_reference=False)
    sch = Scheduler(max_bs, 2048, tok, prefill_timeout=0.0)
    sch.batch_sizes = [b for b in graph_sizes if b <= max_bs]
    seqs = [Sequence(i, p, tok, max_new) for i, p in enumerate(prompts)]
    for s in seqs:
        sch.add_request(s)
    caches = [(cache.k_caches, cache.v_caches)]
    for _ in range(10000):
        batch, kind = sch.get_next_batch()
        if not batch:
            break
        ids_list = [s.get_next_input_ids() for s in batch]
        if kind == "prefill":
            S = len(ids_list[0])
            for s in batch:
                assert cache.alloc(s.seq_id, S)[0]
            bt, _ = cache.cache_batch_data([s.seq_id for s in batch], [S] * len(batch))
            logits = ref.prefill(torch.tensor(ids_list), caches, bt)[:, -1]
        else:
            seen = set()
            for s in batch:
                if s.seq_id not in seen:
                    cache.append(s.seq_id)
                    seen.add(s.seq_id)
            bt, sl = cache.cache_batch_data([s.seq_id for s in batch], [s.current_position for s in batch])
            logits = ref.decode(torch.tensor(ids_list).squeeze(1), caches, sl, bt)
        nxt = logits.float().argmax(-1).tolist()
        done = set()
        for s, t in zip(batch, nxt):
            if s.seq_id in done:
                continue
            done.add(s.seq_id)
            s.update_state(t, None)
            if s.is_finished():
                cache.free(s.seq_id)
                sch.mark_finished(s)
    return {s.prompt: s.output_ids for s in seqs}, logits


def test_engine_generate_matches_the_oracle_loop(mv):
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-gqa"]
    tok = Tok(cfg.vocab_size)
    cpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1).to(DEV)
    prompts = ["a" * 12, "bc" * 9, "hello world, this is a longer prompt " * 2, "zz"]
    max_new = 6
    eng = InferenceEngine(model=gpu_model, tokenizer=tok, max_batch_size=4, n_blocks=24, max_tokens=1024,
                          graph_batch_sizes=[1, 2, 4], prefill_timeout=0.0)
    collected = {}
    for p in prompts:
        sid = eng.add_request(p, max_tokens=max_new, temperature=0.0)
        collected[sid] = []
        eng.register_stream_callback(sid, lambda t, txt, sid=sid: collected[sid].append(t))
    by_id = {s.seq_id: s for s in eng.scheduler.waiting_queue}
    from micro_vllm_amd.core.inference_context import BatchInferenceContext
    kinds = []
    for _ in range(1000):
        batch, kind = eng.get_next_batch()
        if not batch:
            break
        kinds.append((kind, len(batch)))
        ctx = BatchInferenceContext(len(batch), kind, batch)
        assert eng.step(ctx)
        eng.update_sequences(ctx.sequences)
    got = {s.prompt: s.output_ids for s in by_id.values()}
    want, _ = _oracle_generate(cfg, cpu_model, tok, prompts, max_new, 24, 1024, 4)
    assert kinds[0][0] == "prefill" and any(k == "decode" for k, _ in kinds)
    # Greedy decoding is discontinuous in the logits: a near-tie between the top two can flip
    # and the continuations then differ legitimately.  Require identical first tokens for every
    # prompt (prefill parity) and full agreement for at least three of the four continuations.
    agree = 0
    for p in prompts:
        assert len(got[p]) >= 1 and len(got[p]) <= max_new
        assert got[p][0] == want[p][0], f"first token differs for {p!r}: {got[p]} vs {want[p]}"
        agree += int(got[p] == want[p])
    assert agree >= 3, (got, want)
    for sid, toks in collected.items():
        assert toks == by_id[sid].output_ids          # every token streamed exactly once
    assert eng.cache_manager.stats["free"] == 24      # every page returned
    assert not eng.scheduler.running_sequences or all(s.is_finished() for s in eng.scheduler.running_sequences)


def test_engine_generate_api_and_sampling_path(mv):
    """`generate` drains its prompts; with temperature > 0 the stochastic sampler runs on the device
    and returns valid ids."""
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-mha"]
    tok = Tok(cfg.vocab_size)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=7, qkv_bias=True).to(DEV)
    eng = InferenceEngine(model=gpu_model, tokenizer=tok, max_batch_size=2, n_blocks=12, max_tokens=512,
                          graph_batch_sizes=[1, 2], prefill_timeout=0.0)
    torch.manual_seed(0)
    out = eng.generate(["first prompt", "second"], max_tokens=5, temperature=0.8, top_p=0.9)
    assert set(out) == {"first prompt", "second"}
    for text in out.values():
        ids = [int(t) for t in text.split()]
        assert all(0 <= t < cfg.vocab_size for t in ids)


def test_engine_soak_many_ragged_requests(mv):
    """120 requests of ragged prompt lengths (1..700 tokens, crossing zero to two page boundaries),
    1..24 new tokens each, greedy and sampled mixed, batches of up to 8 with every captured graph
    size in play: every request finishes with exactly the tokens it asked for (no end-of-sequence
    id in the vocabulary the stub tokenizer emits), every token is streamed once, no id leaves the
    vocabulary, and every page is back in the pool at the end - repeatedly, so pages freed by one
    wave of requests are reused by the next."""
    import random

    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.core.inference_context import BatchInferenceContext
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-gqa"]
    tok = Tok(cfg.vocab_size)
    gpu_model = build_synthetic_model(cfg, "cpu", seed=11, norm_jitter=0.1).to(DEV)
    n_blocks = 200   # the scheduler admits prefills without looking at the pool (as the reference): room for all
    eng = InferenceEngine(model=gpu_model, tokenizer=tok, max_batch_size=8, n_blocks=n_blocks, max_tokens=1024,
                          graph_batch_sizes=[1, 2, 4, 8], prefill_timeout=0.0)
    eng.eos_token_id = None
    rnd = random.Random(5)
    torch.manual_seed(5)
    for wave in range(3):
        want_len, streamed = {}, {}
        for i in range(40):
            plen = rnd.choice([1, 2, 17, 255, 256, 257, 300, 511, 513, 700])
            prompt = "".join(chr(33 + rnd.randrange(90)) for _ in range(plen))
            new = rnd.randint(1, 24)
            sid = eng.add_request(prompt + f"#{wave}.{i}", max_tokens=new, temperature=rnd.choice([0.0, 0.0, 0.7, 1.3]),
                                  top_p=rnd.choice([0.5, 0.9, 1.0]))
            want_len[sid] = new
            streamed[sid] = []
            eng.register_stream_callback(sid, lambda t, txt, sid=sid: streamed[sid].append(t))
        seqs = {s.seq_id: s for s in eng.scheduler.waiting_queue}
        assert len(seqs) == 40, "request ids collided"
        sizes = set()
        for _ in range(20000):
            batch, kind = eng.get_next_batch()
            if kind == "waiting":
                continue
            if not batch:
                break
            sizes.add((kind, len(batch)))
            ctx = BatchInferenceContext(len(batch), kind, batch)
            assert eng.step(ctx)
            eng.update_sequences(ctx.sequences)
        assert not eng.scheduler.waiting_queue
        for sid, s in seqs.items():
            assert s.is_finished()
            assert len(s.output_ids) == want_len[sid], (len(s.output_ids), want_len[sid])
            assert streamed[sid] == s.output_ids
            assert all(0 <= t < cfg.vocab_size for t in s.output_ids)
        assert eng.cache_manager.stats["free"] == n_blocks
        assert {n for k, n in sizes if k == "decode"} >= {1, 2, 4, 8} or wave > 0
        eng.scheduler.get_finished_results()
        eng.scheduler.running_sequences.clear()


@pytest.mark.parametrize("scheme,shape,bias", [("llama", "tiny-d128", False), ("qwen", "tiny-mha", True),
                                               # Qwen2-like: q / k / v biases, 7 query heads on one kv head
                                               ("llama", "tiny-gqa7", True)])
def test_engine_from_a_checkpoint_directory_equals_engine_from_the_model_object(mv, tmp_path, scheme, shape, bias):
    """`InferenceEngine(model_path=...)` (core/engine.py:45-47, core/model_loader.py:6): the
    safetensors directory is read parameter by parameter and must give the very same greedy
    continuations as the model object it was written from; the rotary base of the checkpoint's
    config reaches the rotary tables."""
    import dataclasses
    import os, sys
    sys.path.insert(0, os.path.dirname(__file__))
    from ckpt_util import write_checkpoint
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.core.paged_attention import rotary_tables
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    theta = 500000.0 if scheme == "llama" else 10000.0
    cfg = dataclasses.replace(SHAPES[shape], rope_theta=theta)
    tok = Tok(cfg.vocab_size)
    src = build_synthetic_model(cfg, "cpu", seed=11, qkv_bias=bias, norm_jitter=0.1)
    path = write_checkpoint(src, str(tmp_path / "ckpt"), scheme, shards=2)
    prompts = ["a" * 12, "bc" * 9, "hello world, this is a longer prompt " * 2]
    kw = dict(tokenizer=tok, max_batch_size=4, n_blocks=24, max_tokens=1024, graph_batch_sizes=[1, 2, 4],
              prefill_timeout=0.0)
    a = InferenceEngine(model_path=path, **kw)
    cos, sin = rotary_tables(cfg.head_dim, a.graph_runner.attention._cos_pool.shape[0], theta, DEV)
    assert torch.equal(a.graph_runner.attention._cos_pool, cos.to(torch.bfloat16))
    assert torch.equal(a.graph_runner.attention._sin_pool, sin.to(torch.bfloat16))
    got = a.generate(prompts, max_tokens=6, temperature=0.0)
    b = InferenceEngine(model=src.to(DEV), **kw)
    want = b.generate(prompts, max_tokens=6, temperature=0.0)
    assert got == want and len(got) == len(prompts)
    assert all(len(v.split()) >= 1 for v in got.values())


def test_http_front_end_over_the_real_engine(mv):
    """api_server.py:96-184 over the device path: the engine runs on the server's own thread;
    greedy /generate, /batch_generate and the event stream give the tokens `engine.generate` gives."""
    import json
    from starlette.testclient import TestClient
    from micro_vllm_amd.api_server import EngineLoop, create_app
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-gqa"]
    tok = Tok(cfg.vocab_size)
    kw = dict(tokenizer=tok, max_batch_size=4, n_blocks=24, max_tokens=1024, graph_batch_sizes=[1, 2, 4],
              prefill_timeout=0.0)
    prompts = ["a" * 12, "bc" * 9, "hello world, this is a longer prompt " * 2]
    want = InferenceEngine(model=build_synthetic_model(cfg, "cpu", seed=3, norm_jitter=0.1).to(DEV), **kw) \
        .generate(prompts, max_tokens=6, temperature=0.0)
    eng = InferenceEngine(model=build_synthetic_model(cfg, "cpu", seed=3, norm_jitter=0.1).to(DEV), **kw)
    loop = EngineLoop(eng)
    loop.start()
    try:
        with TestClient(create_app(loop)) as c:
            assert c.get("/health").json()["device"] == "cuda"
            r = c.post("/generate", json={"prompt": prompts[0], "max_tokens": 6, "temperature": 0.0}).json()
            assert r["text"] == want[prompts[0]] and 1 <= r["tokens"] <= 6
            r = c.post("/batch_generate", json={"prompts": prompts, "max_tokens": 6, "temperature": 0.0}).json()
            assert [x["text"] for x in r["results"]] == [want[p] for p in prompts]
            with c.stream("POST", "/generate_stream", json={"prompt": prompts[1], "max_tokens": 6,
                                                            "temperature": 0.0, "stream": True}) as resp:
                events = [json.loads(l[5:]) for l in resp.iter_lines() if l.startswith("data:")]
            streamed = [e["token"] for e in events]
            assert tok.decode(tok.encode(prompts[1]) + streamed) == want[prompts[1]]
            # sampled requests go through the in-graph pick and come back with valid ids
            r = c.post("/generate", json={"prompt": "zz", "max_tokens": 5, "temperature": 0.8}).json()
            assert 1 <= r["tokens"] <= 5
    finally:
        loop.stop()
    assert eng.cache_manager.stats["free"] == 24


def test_checkpoint_directory_with_its_tokenizer_end_to_end(mv, tmp_path):
    """Directory in, text out: weights from safetensors, tokenizer from the same directory, greedy
    continuation equal to the one an engine over the source model object produces."""
    import os, sys
    sys.path.insert(0, os.path.dirname(__file__))
    from ckpt_util import write_byte_tokenizer, write_checkpoint
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.core.model_loader import load_tokenizer
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    src = build_synthetic_model(SHAPES["tiny-gqa"], "cpu", seed=21, norm_jitter=0.1)
    path = write_byte_tokenizer(write_checkpoint(src, str(tmp_path / "ckpt"), "llama"))
    kw = dict(max_batch_size=2, n_blocks=12, max_tokens=512, graph_batch_sizes=[1, 2], prefill_timeout=0.0)
    prompts = ["The quick brown fox", "jumps over"]
    got = InferenceEngine(path, **kw).generate(prompts, max_tokens=8, temperature=0.0)
    want = InferenceEngine(model=src.to(DEV), tokenizer=load_tokenizer(path), **kw) \
        .generate(prompts, max_tokens=8, temperature=0.0)
    assert got == want and set(got) == set(prompts)
    assert all(got[p].startswith(p) for p in prompts)     # the text holds the prompt, then the continuation


def test_kv_pool_sized_from_free_device_memory(mv):
    """n_blocks="auto": the pool takes a share of what is free after the weights are in place."""
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-gqa"]
    tok = Tok(cfg.vocab_size)
    free, _ = torch.cuda.mem_get_info()
    eng = InferenceEngine(model=build_synthetic_model(cfg, "cpu", seed=5).to(DEV), tokenizer=tok,
                          max_batch_size=2, n_blocks="auto", kv_fraction=0.002, max_tokens=512,
                          graph_batch_sizes=[1, 2], prefill_timeout=0.0)
    per_block = 2 * 256 * cfg.num_key_value_heads * cfg.head_dim * 2 * cfg.num_hidden_layers
    total = eng.cache_manager.stats["free"]
    assert total >= 81 and abs(total - int(free * 0.002) // per_block) <= max(81, total // 10)
    out = eng.generate(["abc", "defg"], max_tokens=4, temperature=0.0)
    assert len(out) == 2 and eng.cache_manager.stats["free"] == total


def test_requests_that_cannot_be_scheduled_are_refused_at_intake(mv):
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-gqa"]
    eng = InferenceEngine(model=build_synthetic_model(cfg, "cpu", seed=5).to(DEV), tokenizer=Tok(cfg.vocab_size),
                          max_batch_size=2, max_prefill_tokens=64, n_blocks=12, max_tokens=512,
                          graph_batch_sizes=[1, 2], prefill_timeout=0.0)
    with pytest.raises(ValueError, match="max_prefill_tokens=64"):
        eng.add_request("x" * 65, max_tokens=4)
    with pytest.raises(ValueError, match="512 positions"):
        eng.add_request("x" * 60, max_tokens=452)          # 60 + 452 + 1 > 512
    with pytest.raises(ValueError):
        eng.add_request("", max_tokens=4)
    assert not eng.scheduler.waiting_queue
    assert len(eng.generate(["x" * 64], max_tokens=447, temperature=0.0)) == 1   # exactly at both limits


def test_outputs_do_not_depend_on_what_recycled_pages_held(mv):
    """The position numbering leaves one slot per sequence unwritten and attended (index S after a
    prompt of S tokens).  The engine zeroes it on the first decode step, so the same greedy prompt
    gives the same tokens on a fresh pool and on a pool whose pages a previous request filled; with
    the zeroing off (the reference's behaviour) the dirty pool may change them."""
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-gqa"]
    tok = Tok(cfg.vocab_size)
    prompts = ["the quick brown fox " * 3, "pack my box with five dozen liquor jugs", "xy"]

    def run(dirty):
        gpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1).to(DEV)
        eng = InferenceEngine(model=gpu_model, tokenizer=tok, max_batch_size=4, n_blocks=6, max_tokens=512,
                              graph_batch_sizes=[1, 2, 4], prefill_timeout=0.0)
        if dirty:   # what recycled pages look like: every slot of the pool holds somebody's keys
            g = torch.Generator(device=DEV).manual_seed(9)
            eng.cache_manager._k_store.normal_(0.0, 3.0, generator=g)
            eng.cache_manager._v_store.normal_(0.0, 3.0, generator=g)
        out = eng.generate(prompts, max_tokens=8, temperature=0.0)
        assert eng.cache_manager.stats["free"] == 6
        return out

    clean, dirty = run(False), run(True)
    assert clean == dirty


def test_first_token_from_the_last_real_position(mv):
    """`sample_last_real_token=True`: a short prompt batched with a longer one gets its first token
    from its own last token's logits - the token it gets when it is prefilled alone - instead of
    from the pad position (the reference's behaviour, which stays the default)."""
    from micro_vllm_amd.core.engine import InferenceEngine
    from micro_vllm_amd.core.inference_context import BatchInferenceContext
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES["tiny-gqa"]
    tok = Tok(cfg.vocab_size)
    short, long_ = "abcdefg", "hello world, this is a longer prompt!!"

    def first_tokens(prompts, flag):
        gpu_model = build_synthetic_model(cfg, "cpu", seed=1234, norm_jitter=0.1).to(DEV)
        eng = InferenceEngine(model=gpu_model, tokenizer=tok, max_batch_size=4, n_blocks=8, max_tokens=512,
                              graph_batch_sizes=[1, 2, 4], prefill_timeout=0.0, sample_last_real_token=flag)
        for p in prompts:
            eng.add_request(p, max_tokens=2, temperature=0.0)
        batch, kind = eng.get_next_batch()
        assert kind == "prefill" and len(batch) == len(prompts)
        assert eng.step(BatchInferenceContext(len(batch), kind, batch))
        return {s.prompt: s._next_token for s in batch}

    alone = first_tokens([short], True)[short]
    assert first_tokens([short], False)[short] == alone              # no padding: the flag changes nothing
    together = first_tokens([short, long_], True)
    assert together[short] == alone
    assert together[long_] == first_tokens([long_], False)[long_]
