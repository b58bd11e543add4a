"""HTTP front end over InferenceEngine: the reference's four endpoints (api_server.py:96-184:
GET /health, POST /generate, /batch_generate, /generate_stream with server-sent events) and its
two loops (rank 0 drives the scheduler and tells the other tensor-parallel ranks what to run,
api_server.py:61-90).

What is arranged differently, on purpose:

  * The engine is driven by ONE dedicated thread (`EngineLoop`), not by a coroutine that yields
    between steps: a decode step blocks its caller for milliseconds (graph replay + the read-back
    of the picked ids), and on the event loop that would stall every open connection.  Requests
    enter through `submit`, tokens and completions leave through callbacks that the HTTP side
    bridges into asyncio queues with `call_soon_threadsafe`.
  * /generate and /batch_generate go through the same loop as the streams (the reference calls
    `engine.generate` from the handler while its background loop is pulling batches from the same
    scheduler, api_server.py:112).
  * `tokens` in a response counts generated tokens (the reference reports `len(text)`).
  * An idle loop sleeps a fraction of the scheduler's prefill timeout instead of spinning.
"""
import asyncio
import json
import logging
import threading
import time
from typing import Callable, List, Optional

from .core.inference_context import BatchInferenceContext
from .core.parallel_config import get_rank, get_world_size, rank0

logger = logging.getLogger("api_server")


def _hard_exit(code: int):
    """End the process without unwinding: other threads may be blocked in a collective."""
    import os
    import sys
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    time.sleep(0.5)          # let the event loop write the error responses that were just queued
    os._exit(code)


class _Request:
    __slots__ = ("on_done", "seq")

    def __init__(self, on_done):
        self.on_done, self.seq = on_done, None


class EngineLoop:
    """Owns the thread that talks to the device.  Rank 0: `run()` pulls batches from the
    scheduler, broadcasts each step's context before and after it (so the other ranks run the same
    step and then apply the same tokens), and reports tokens / completions.  Other ranks:
    `run_follower()`."""

    def __init__(self, engine):
        self.engine = engine
        self._lock = threading.Lock()       # the scheduler's queues are touched from two threads
        self._requests = {}
        self._running = False
        self._thread: Optional[threading.Thread] = None
        self.steps = 0
        self.fatal: Optional[BaseException] = None
        # what ends the process after a failure the ranks cannot recover from together (a
        # supervisor starts a fresh one); tests put a recorder here
        self.exit_process: Callable[[int], None] = _hard_exit

    # ---- called from any thread
    def submit(self, prompt: str, max_tokens: int, temperature: float, top_p: float,
               on_token: Optional[Callable] = None, on_done: Optional[Callable] = None) -> int:
        if self.fatal is not None:
            raise RuntimeError(f"the engine stopped after an unrecoverable failure: {self.fatal!r}")
        req = _Request(on_done)
        with self._lock:
            # raises ValueError for a request that can never be scheduled (the handlers answer 400)
            seq_id = self.engine.add_request(prompt, max_tokens, temperature, top_p)
            req.seq = next(s for s in reversed(self.engine.scheduler.waiting_queue) if s.seq_id == seq_id)
            # the id is a hash of prompt and arrival time (core/engine.py:139-146): the same prompt
            # twice within one clock tick would share it, and with it KV pages and completion
            while seq_id in self._requests:
                seq_id = (seq_id + 1) % (2 ** 32)
            req.seq.seq_id = seq_id
            self._requests[seq_id] = req
            if on_token is not None:
                self.engine.register_stream_callback(seq_id, on_token)
        return seq_id

    def start(self, switch_interval: Optional[float] = 2e-4):
        """`switch_interval`: the interpreter hands the GIL from a busy thread to a waiting one only
        every 5 ms by default - longer than a decode step, so the engine thread coming back from
        its wait on the device would sit behind the event loop's response writing.  A short
        interval keeps the device fed; None leaves the setting alone."""
        if switch_interval is not None:
            import sys
            sys.setswitchinterval(switch_interval)
        self._running = True
        self._thread = threading.Thread(target=self.run, name="engine-loop", daemon=True)
        self._thread.start()

    def stop(self):
        self._running = False
        if self._thread is not None:
            self._thread.join(timeout=10)
            self._thread = None

    # ---- the device-facing thread
    def run(self):
        eng = self.engine
        idle = max(getattr(eng.scheduler, "prefill_timeout", 0.02) / 4, 2e-4)
        self._running = True
        if getattr(eng, "device_index", None) is not None:
            import torch
            torch.cuda.set_device(eng.device_index)     # the current device is a per-thread setting
        while self._running:
            with self._lock:
                batch, kind = eng.get_next_batch()
            if kind == "waiting" or not batch:
                BatchInferenceContext(0, "waiting").broadcast()
                time.sleep(idle)
                continue
            ctx = BatchInferenceContext(len(batch), kind, batch)
            ctx.broadcast()
            try:
                eng.step(ctx)
            except Exception as e:            # e.g. out of KV pages: fail these requests, keep serving
                logger.exception("step failed")
                if get_world_size() > 1:
                    # the other ranks are inside the same step and expect its second message: there
                    # is no way back into lockstep from here.  Every open request is answered with
                    # the error and the process ends non-zero (the followers, blocked in their
                    # receive, go down with the process group; a supervisor starts all ranks afresh)
                    self._fatal(e)
                    return
                starved = getattr(e, "seq_ids", None)
                if starved is not None:
                    # a decode step found no page for some sequences and undid itself: only those
                    # fail, the rest of the batch is scheduled again as if the step had not been tried
                    self._fail([s for s in ctx.sequences if s.seq_id in set(starved)], e)
                else:
                    self._fail(ctx.sequences, e)
                continue
            ctx.broadcast()
            with self._lock:
                eng.update_sequences(ctx.sequences)
                eng.scheduler.get_finished_results()      # drained here: completions go out below
            self.steps += 1
            self._report(ctx.sequences)
        # tell the other ranks that no further step will come (the reference's followers have no
        # way out of their loop, api_server.py:79-90)
        BatchInferenceContext(0, "idle").broadcast()

    def _report(self, sequences, error=None):
        seen = set()
        for seq in sequences:
            if seq.seq_id in seen or not (error or seq.is_finished()):
                continue
            seen.add(seq.seq_id)
            req = self._requests.pop(seq.seq_id, None)
            self.engine.unregister_stream_callback(seq.seq_id)
            if req is not None and req.on_done is not None:
                req.on_done(seq, error)

    def _fail(self, sequences, error):
        with self._lock:
            for seq in sequences:
                seq.state = "finished"
                seq.max_tokens = len(seq.output_ids)     # is_finished() from here on
                self.engine.cache_manager.free(seq.seq_id)
                self.engine.scheduler.mark_finished(seq)
            self.engine.scheduler.get_finished_results()
        self._report(sequences, error)

    def _fatal(self, error):
        self.fatal = error
        self._running = False
        with self._lock:
            open_ids = list(self._requests)
        for seq_id in open_ids:
            req = self._requests.pop(seq_id, None)
            self.engine.unregister_stream_callback(seq_id)
            if req is not None and req.on_done is not None:
                try:
                    req.on_done(req.seq, error)
                except Exception:
                    logger.exception("reporting the failure of request %s", seq_id)
        self.exit_process(1)

    def run_follower(self):
        """api_server.py:79-90: receive the step, run it, receive the tokens, apply them."""
        eng = self.engine
        self._running = True
        while self._running:
            ctx = BatchInferenceContext.receive(eng.tokenizer)
            if ctx.batch_type == "idle":
                break                       # rank 0 has stopped serving
            if ctx.batch_type == "waiting" or ctx.batch_size == 0:
                continue
            eng.step(ctx)
            ctx = BatchInferenceContext.receive(eng.tokenizer)
            eng.update_sequences(ctx.sequences)


def create_app(loop: EngineLoop, model_name: str = "model"):
    """The FastAPI application over a started (or to be started) EngineLoop."""
    from fastapi import FastAPI, HTTPException
    from fastapi.responses import StreamingResponse
    from pydantic import BaseModel

    class GenerateReq(BaseModel):
        prompt: str
        max_tokens: int = 128
        temperature: float = 0.7
        top_p: float = 0.9
        stream: bool = False

    class BatchGenerateReq(BaseModel):
        prompts: List[str]
        max_tokens: int = 128
        temperature: float = 0.7
        top_p: float = 0.9

    class GenerateResp(BaseModel):
        text: str
        tokens: int
        time_ms: float

    class BatchGenerateResp(BaseModel):
        results: List[GenerateResp]

    class HealthResp(BaseModel):
        status: str = "healthy"
        model: str = model_name
        device: str
        running_sequences: int
        waiting_sequences: int

    app = FastAPI(title="micro-vllm MI355X API server", version="0.1.0")
    engine = loop.engine

    async def _complete(prompt, max_tokens, temperature, top_p):
        aio = asyncio.get_running_loop()
        fut = aio.create_future()
        t0 = time.time()

        def done(seq, error):
            aio.call_soon_threadsafe(fut.set_result, (seq, error))

        try:
            loop.submit(prompt, max_tokens, temperature, top_p, on_done=done)
        except ValueError as e:
            raise HTTPException(400, str(e))
        seq, error = await fut
        if error is not None:
            raise HTTPException(500, f"generation failed: {error}")
        text = engine.tokenizer.decode(seq.full_ids, skip_special_tokens=True)
        return GenerateResp(text=text, tokens=len(seq.output_ids), time_ms=(time.time() - t0) * 1000)

    @app.get("/health", response_model=HealthResp)
    async def health():
        return HealthResp(device=str(engine.device),
                          running_sequences=len(engine.scheduler.running_sequences),
                          waiting_sequences=len(engine.scheduler.waiting_queue))

    @app.post("/generate", response_model=GenerateResp)
    async def generate(req: GenerateReq):
        return await _complete(req.prompt, req.max_tokens, req.temperature, req.top_p)

    @app.post("/batch_generate", response_model=BatchGenerateResp)
    async def batch_generate(req: BatchGenerateReq):
        results = await asyncio.gather(*[
            _complete(p, req.max_tokens, req.temperature, req.top_p) for p in req.prompts])
        return BatchGenerateResp(results=list(results))

    @app.post("/generate_stream")
    async def generate_stream(req: GenerateReq):
        if not req.stream:
            return await generate(req)
        aio = asyncio.get_running_loop()
        queue: asyncio.Queue = asyncio.Queue()

        def on_token(token, text):
            aio.call_soon_threadsafe(queue.put_nowait, ("token", token, text))

        def on_done(seq, error):
            aio.call_soon_threadsafe(queue.put_nowait, ("done", seq, error))

        try:
            loop.submit(req.prompt, req.max_tokens, req.temperature, req.top_p, on_token, on_done)
        except ValueError as e:
            raise HTTPException(400, str(e))

        async def events():
            full_text = ""
            while True:
                kind, a, b = await queue.get()
                if kind == "done":
                    if b is not None:
                        yield f"data: {json.dumps({'error': str(b), 'finished': True})}\n\n"
                    break
                full_text += b
                data = {"token": a, "text": b, "full_text": full_text,
                        "finished": a == engine.eos_token_id}
                yield f"data: {json.dumps(data, ensure_ascii=False)}\n\n"

        return StreamingResponse(events(), media_type="text/event-stream")

    return app


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="Serve a safetensors checkpoint directory over HTTP")
    ap.add_argument("model_path")
    ap.add_argument("--host", default="0.0.0.0")
    ap.add_argument("--port", type=int, default=8000)
    ap.add_argument("--max-batch-size", type=int, default=40)
    ap.add_argument("--max-prefill-tokens", type=int, default=2048)
    ap.add_argument("--max-tokens", type=int, default=4352, help="longest sequence a block-table row describes")
    ap.add_argument("--kv-blocks", type=int, default=None,
                    help="KV pages of 256 tokens (default: sized from the free device memory)")
    ap.add_argument("--kv-fraction", type=float, default=0.9,
                    help="share of the free device memory the KV pool takes when --kv-blocks is not given")
    ap.add_argument("--exact-rope", action="store_true")
    ap.add_argument("--trust-remote-code", action="store_true", help="for tokenizers shipped as Python")
    args = ap.parse_args(argv)

    from .core.engine import InferenceEngine
    engine = InferenceEngine(args.model_path, args.max_batch_size, args.max_prefill_tokens,
                             n_blocks=args.kv_blocks if args.kv_blocks is not None else "auto",
                             kv_fraction=args.kv_fraction, max_tokens=args.max_tokens,
                             exact_rope=args.exact_rope, trust_remote_code=args.trust_remote_code)
    loop = EngineLoop(engine)
    logger.info("rank %d: model loaded on %s", get_rank(), engine.device)
    if not rank0():
        loop.run_follower()
        return
    import uvicorn
    app = create_app(loop, model_name=args.model_path)
    loop.start()
    try:
        uvicorn.run(app, host=args.host, port=args.port)
    finally:
        loop.stop()
        engine.shutdown()


if __name__ == "__main__":
    main()
