"""HTTP front end and load generator on the CPU (reference: api_server.py:96-184, benchmark.py):
the real Scheduler / Sequence / update_sequences bookkeeping, with the device step replaced by a
rule (next token = last token + 1), so every response is predictable."""
import asyncio
import json
import threading

import pytest

import micro_vllm_amd  # noqa: F401
from micro_vllm_amd.api_server import EngineLoop, create_app
from micro_vllm_amd.core.engine import InferenceEngine
from micro_vllm_amd.core.scheduler import Scheduler

VOCAB = 1000


class Tok:
    eos_token_id = 2

    def encode(self, prompt, add_special_tokens=True):
        return [5 + (ord(c) * 7) % (VOCAB - 5) for c in prompt]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)


class Pages:
    block_size, max_seq_blocks = 256, 16        # what the engine's intake check reads

    def __init__(self):
        self.freed = []

    def free(self, sid):
        self.freed.append(sid)


class RuleEngine(InferenceEngine):
    """InferenceEngine's request intake and token bookkeeping over a step that needs no device."""

    def __init__(self, fail_on=None):
        self.tokenizer = Tok()
        self.scheduler = Scheduler(8, 2048, self.tokenizer, prefill_timeout=0.0)
        self.scheduler.batch_sizes = [1, 2, 4, 8]
        self.stream_callbacks = {}
        self.cache_manager = Pages()
        self.eos_token_id = self.tokenizer.eos_token_id
        self.device = "rule"
        self.fail_on = fail_on
        self.kinds = []

    def step(self, ctx):
        self.kinds.append((ctx.batch_type, len(ctx.sequences)))
        if self.fail_on and any(self.fail_on in s.prompt for s in ctx.sequences):
            raise RuntimeError("OOM")
        for s in ctx.sequences:
            s._next_token = 5 + (s.full_ids[-1] + 1 - 5) % (VOCAB - 5)
        return True


def expected(prompt, n):
    ids = Tok().encode(prompt)
    out = []
    for _ in range(n):
        out.append(5 + ((out[-1] if out else ids[-1]) + 1 - 5) % (VOCAB - 5))
    return ids, out


@pytest.fixture
def client():
    from starlette.testclient import TestClient
    eng = RuleEngine()
    loop = EngineLoop(eng)
    loop.start()
    with TestClient(create_app(loop, "rule-model")) as c:
        c.engine, c.loop = eng, loop
        yield c
    loop.stop()


def test_health_generate_and_batch(client):
    h = client.get("/health").json()
    assert h == {"status": "healthy", "model": "rule-model", "device": "rule",
                 "running_sequences": 0, "waiting_sequences": 0}
    r = client.post("/generate", json={"prompt": "hello", "max_tokens": 5}).json()
    ids, out = expected("hello", 5)
    assert r["text"] == Tok().decode(ids + out) and r["tokens"] == 5 and r["time_ms"] > 0
    prompts = ["ab", "a much longer prompt than the others " * 3, "xyz"]
    r = client.post("/batch_generate", json={"prompts": prompts, "max_tokens": 4}).json()
    assert [x["text"] for x in r["results"]] == [Tok().decode(sum(expected(p, 4), [])) for p in prompts]
    assert len(client.engine.cache_manager.freed) == 4            # every sequence returned its pages
    assert not client.loop._requests and not client.engine.stream_callbacks
    assert not client.engine.scheduler.finished_sequences       # drained by the loop, no growth
    assert client.engine.kinds[0][0] == "prefill" and any(k == "decode" for k, _ in client.engine.kinds)


def test_stream_emits_one_event_per_token(client):
    with client.stream("POST", "/generate_stream",
                       json={"prompt": "stream me", "max_tokens": 6, "stream": True}) as resp:
        assert resp.headers["content-type"].startswith("text/event-stream")
        events = [json.loads(l[5:]) for l in resp.iter_lines() if l.startswith("data:")]
    _, out = expected("stream me", 6)
    assert [e["token"] for e in events] == out
    assert events[-1]["full_text"] == "".join(str(t) for t in out)
    assert all(e["finished"] == (e["token"] == 2) for e in events)
    # stream=False falls back to the plain response, as the reference does
    r = client.post("/generate_stream", json={"prompt": "x", "max_tokens": 2}).json()
    assert r["tokens"] == 2


def test_prompts_that_can_never_be_scheduled_are_refused(client):
    too_long = "x" * 2049                      # the scheduler's max_prefill_tokens is 2048
    assert client.post("/generate", json={"prompt": too_long, "max_tokens": 2}).status_code == 400
    assert client.post("/generate_stream", json={"prompt": too_long, "max_tokens": 2, "stream": True}).status_code == 400
    assert client.post("/generate", json={"prompt": "", "max_tokens": 2}).status_code == 400
    assert not client.engine.scheduler.waiting_queue and not client.loop._requests
    assert client.post("/generate", json={"prompt": "x" * 2048, "max_tokens": 2}).json()["tokens"] == 2


def test_a_failing_step_fails_its_requests_and_the_loop_goes_on():
    from starlette.testclient import TestClient
    eng = RuleEngine(fail_on="boom")
    loop = EngineLoop(eng)
    loop.start()
    try:
        with TestClient(create_app(loop)) as c:
            assert c.post("/generate", json={"prompt": "boom", "max_tokens": 3}).status_code == 500
            assert c.post("/generate", json={"prompt": "fine", "max_tokens": 3}).json()["tokens"] == 3
            assert c.get("/health").json()["running_sequences"] == 0
    finally:
        loop.stop()


def test_a_decode_step_without_pages_fails_only_the_starved_sequence():
    """A decode step that found no KV page for one sequence undoes itself (KVPoolExhausted names the
    sequence): that request fails, the other of the same batch finishes with the tokens it would have
    had anyway."""
    from starlette.testclient import TestClient

    from micro_vllm_amd.core.engine import KVPoolExhausted

    class Starving(RuleEngine):
        def step(self, ctx):
            if ctx.batch_type == "decode":
                hit = [s.seq_id for s in ctx.sequences if "starve" in s.prompt]
                if hit:
                    raise KVPoolExhausted(hit)
            return super().step(ctx)

    eng = Starving()
    loop = EngineLoop(eng)
    loop.start()
    try:
        with TestClient(create_app(loop)) as c:
            import concurrent.futures as cf
            with cf.ThreadPoolExecutor(2) as ex:
                bad = ex.submit(lambda: c.post("/generate", json={"prompt": "starve me", "max_tokens": 6}))
                good = ex.submit(lambda: c.post("/generate", json={"prompt": "feed me", "max_tokens": 6}))
                bad, good = bad.result(), good.result()
            assert bad.status_code == 500 and "OOM" in bad.text
            ids, out = expected("feed me", 6)
            assert good.json()["text"] == Tok().decode(ids + out)
            assert c.get("/health").json()["running_sequences"] == 0 and not loop._requests
    finally:
        loop.stop()


def test_with_several_ranks_a_failing_step_answers_every_request_and_ends_the_process(monkeypatch):
    """Rank 0 of a group cannot get the other ranks back into step after a failed step: every open
    request gets the error, new ones are refused, and the process is told to exit non-zero."""
    import time

    import micro_vllm_amd.api_server as srv

    monkeypatch.setattr(srv, "get_world_size", lambda: 2)
    monkeypatch.setattr(srv.BatchInferenceContext, "broadcast", lambda self: None)
    eng = RuleEngine(fail_on="boom")
    loop = EngineLoop(eng)
    exits, done = [], []
    loop.exit_process = exits.append
    loop.submit("a boom prompt", 4, 0.0, 1.0, on_done=lambda seq, err: done.append(("boom", err)))
    loop.submit("an innocent one that is much longer and sits in another bucket " * 4, 4, 0.0, 1.0,
                on_done=lambda seq, err: done.append(("other", err)))
    loop.start()
    t0 = time.time()
    while not exits and time.time() - t0 < 10:
        time.sleep(0.01)
    loop.stop()
    assert exits == [1]
    assert sorted(k for k, _ in done) == ["boom", "other"] and all(isinstance(e, RuntimeError) for _, e in done)
    assert not loop._requests
    with pytest.raises(RuntimeError, match="unrecoverable"):
        loop.submit("late", 2, 0.0, 1.0)


def test_load_generator_against_a_live_server():
    """The load generator over real sockets: uvicorn in a thread, aiohttp client."""
    import socket
    import time
    import uvicorn
    from micro_vllm_amd import benchmark

    eng = RuleEngine()
    loop = EngineLoop(eng)
    loop.start()
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    server = uvicorn.Server(uvicorn.Config(create_app(loop), log_level="warning"))
    th = threading.Thread(target=lambda: server.run(sockets=[sock]), daemon=True)
    th.start()
    try:
        for _ in range(200):
            if server.started:
                break
            time.sleep(0.05)
        assert server.started
        out = asyncio.run(benchmark.run(f"http://127.0.0.1:{port}", 6, 7))
        assert out["requests"] == 6 and out["tokens"] == 6 * 7
        assert out["tokens_per_s"] > 0 and out["ttft_s_median"] is not None
        assert out["latency_s_p99"] >= out["latency_s_median"] >= out["ttft_s_median"]
    finally:
        server.should_exit = True
        th.join(timeout=10)
        loop.stop()


def _rank_worker(rank, world, port, out_q):
    import os, sys, time
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MVLLM_DIST_BACKEND="gloo")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch
    torch.set_num_threads(1)
    from micro_vllm_amd.core import parallel_config as pc
    pc.setup()
    eng = RuleEngine()
    loop = EngineLoop(eng)
    if rank == 0:
        done = {}
        loop.start()
        prompts = ["ab", "a much longer prompt than the others " * 3, "xyz", "q"]
        for p in prompts:
            loop.submit(p, 5, 0.0, 0.9, on_done=lambda seq, err, p=p: done.__setitem__(p, list(seq.output_ids)))
        for _ in range(2000):
            if len(done) == len(prompts):
                break
            time.sleep(0.005)
        loop.stop()
        out_q.put((rank, {p: done.get(p) for p in prompts}, sorted(eng.cache_manager.freed), eng.kinds))
    else:
        loop.run_follower()                 # returns when rank 0 stops
        out_q.put((rank, None, sorted(eng.cache_manager.freed), eng.kinds))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_ranks_run_the_same_steps_and_retire_the_same_sequences():
    """world_size 2 over gloo: rank 0's loop broadcasts every step before and after it, the
    follower runs the same batches (api_server.py:61-90), applies the same tokens, returns the
    same pages, and leaves its loop when rank 0 stops."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, outs, freed, kinds = q.get(timeout=120)
        got[rank] = (outs, freed, kinds)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs, freed0, kinds0 = got[0]
    for prompt, ids in outs.items():
        assert ids == expected(prompt, 5)[1], prompt
    assert got[1][1] == freed0 and len(freed0) == 4
    assert got[1][2] == kinds0 and kinds0[0][0] == "prefill"


def test_identical_prompts_in_one_clock_tick_get_distinct_ids(monkeypatch):
    import time as _time
    eng = RuleEngine()
    loop = EngineLoop(eng)
    frozen = _time.time()
    monkeypatch.setattr("micro_vllm_amd.core.engine.time.time", lambda: frozen)
    done = []
    ids = [loop.submit("same prompt", 3, 0.0, 0.9, on_done=lambda s, e: done.append(s.seq_id)) for _ in range(3)]
    monkeypatch.undo()
    assert len(set(ids)) == 3
    loop.start()
    try:
        for _ in range(1000):
            if len(done) == 3:
                break
            _time.sleep(0.005)
    finally:
        loop.stop()
    assert sorted(done) == sorted(ids) and sorted(eng.cache_manager.freed) == sorted(ids)
