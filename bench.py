"""Decode-path benchmark: tokens/s of paged-KV single-token decode on a random-weight
Llama-shaped model, with the HBM roofline of the dominant kernel and the CPU restatement of
the reference path timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

With --gpus N > 1 the model is cut over N tensor-parallel ranks, one process per GPU.  Started
under a launcher (WORLD_SIZE set) the process is one of those ranks; started plainly it becomes
the launcher itself: it starts N fresh rank processes of this file (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT in their environment), passes rank 0's JSON line through,
ends every rank when one fails, and never touches a GPU itself.  More ranks than GPUs is refused
unless MVLLM_SHARE_GPU=1 (a rehearsal of the several-ranks path on one card).

A step = one token for every sequence of the batch: KVCacheManager.append + cache_batch_data
(host), hipGraph replay of the whole N-layer forward (device), greedy argmax on the device
feeding the next step.  Inputs (weights, KV pages, block tables) are resident in HBM before
the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import random
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, vendor peak


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--bs", type=int, default=1)
    ap.add_argument("--ctx", type=int, default=4096)
    ap.add_argument("--layers", type=int, default=0, help="build only this many layers (debug)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prefill", action="store_true", help="skip the prefill leg of the report")
    ap.add_argument("--no-bs32", action="store_true", help="skip the batch-32 legs of the report")
    ap.add_argument("--no-engine", action="store_true",
                    help="skip the InferenceEngine leg of the report (32 requests, prefill and decode mixed)")
    ap.add_argument("--as-rank-of", type=int, default=0, metavar="N",
                    help="one process, no peers: build and run rank 0 of a tensor-parallel group of N "
                         "(the several-ranks launch sequence; its reductions sum one partial)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--tail-steps", type=int, default=50,
                    help="steps timed one by one AFTER the timed region, for the median / p99 step")
    ap.add_argument("--ragged", action="store_true",
                    help="context lengths uniform in [256, ctx] (seed 3) instead of ctx for every row")
    ap.add_argument("--one-split-rule", action="store_true",
                    help="A/B switch: never replay the finer-split graph for rows of uneven length")
    ap.add_argument("--shapes", action="store_true", help="per-projection GB/s table on stderr")
    return ap.parse_args()


def build(args, dev, world):
    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner
    from micro_vllm_amd.models import SHAPES, build_synthetic_model

    cfg = SHAPES[args.model]
    layers = args.layers or cfg.num_hidden_layers
    model = build_synthetic_model(cfg, dev, seed=1234, num_layers=layers)
    alone = {}
    if args.as_rank_of:
        from micro_vllm_amd.kernel.allreduce import IpcAllReduce
        assert world == args.as_rank_of
        alone = dict(reducer=IpcAllReduce(0, 1, max(args.bs, 1), cfg.hidden_size), tp_rank=0, tp_world=world)
    total_tokens = args.ctx + args.steps + args.warmup + args.tail_steps + 8
    blocks_per_seq = (total_tokens + 255) // 256 + 1
    runner = ModelGraphRunner(
        model, layers, cfg.num_attention_heads // world, cfg.head_dim,
        cfg.num_key_value_heads // world, cfg.hidden_size, cfg.intermediate_size // world, dev,
        max_batch_size=args.bs, max_blocks=max(32, blocks_per_seq), release_source_weights=True,
        **alone)
    mgr = KVCacheManager(args.bs * blocks_per_seq + 2, 256, layers, cfg.num_key_value_heads // world,
                         cfg.head_dim, dtype=torch.bfloat16, device=dev,
                         max_tokens=blocks_per_seq * 256, max_batch_size=args.bs)
    # non-contiguous pages (seed 2) and N(0,1) KV contents (seed 1): pure-bandwidth timing
    order = list(mgr._free)
    random.Random(2).shuffle(order)
    mgr._free.clear()
    mgr._free.extend(order)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    mgr._k_store.normal_(0.0, 1.0, generator=g)
    mgr._v_store.normal_(0.0, 1.0, generator=g)
    rng = random.Random(3)
    ctxs = [rng.randint(min(256, args.ctx), args.ctx) if args.ragged else args.ctx
            for _ in range(args.bs)]
    for sid in range(args.bs):
        ok, _ = mgr.alloc(sid, ctxs[sid])
        assert ok, "KV pool too small"
    return cfg, layers, runner, mgr, ctxs


MFMA_PEAK_TFLOPS = 2500.0  # dense bf16, MI355X


def prefill_leg(args, cfg, layers, runner, dev, reps=3):
    """Second half of the metric: one fixed-length prefill of `ctx` tokens (bs 1) through
    ModelPrefillRunner on the same weight image, KV pages written, last-token logits.  Timed with
    HIP events on the current stream; flops = 2 * params * tokens + causal attention."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_prefill import ModelPrefillRunner

    S = args.ctx
    blocks = (S + 255) // 256 + 1
    pre = ModelPrefillRunner(runner.model, layers, runner.num_heads, runner.head_size, runner.kv_num_heads,
                             runner.hidden_dim, runner.intermediate_size, dev, max_batch_size=1,
                             max_blocks=max(32, blocks), release_source_weights=True)
    mgr = KVCacheManager(blocks + 1, 256, layers, runner.kv_num_heads, runner.head_size,
                         dtype=torch.bfloat16, device=dev, max_tokens=blocks * 256, max_batch_size=1)
    order = list(mgr._free)
    random.Random(2).shuffle(order)
    mgr._free.clear()
    mgr._free.extend(order)
    assert mgr.alloc(0, S)[0]
    mgr.cache_batch_data([0], [S])
    ids = torch.randint(0, cfg.vocab_size, (1, S), generator=torch.Generator().manual_seed(0)).to(dev)
    for _ in range(2):
        logits = pre.forward(ids, mgr, 1, last_token_only=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        logits = pre.forward(ids, mgr, 1, last_token_only=True)
    e1.record()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(logits.float()).all()), "non-finite prefill logits"
    t = e0.elapsed_time(e1) * 1e-3 / reps
    hid, inter = cfg.hidden_size, cfg.intermediate_size // 2
    qkv_dim = (cfg.num_attention_heads + 2 * cfg.num_key_value_heads) * cfg.head_dim
    flops = 2.0 * S * layers * (hid * qkv_dim + hid * hid + 3 * hid * inter) + 2.0 * hid * cfg.vocab_size \
        + 4.0 * layers * cfg.num_attention_heads * cfg.head_dim * S * S / 2
    return {"metric": "prefill tokens/sec", "value": round(S / t, 1), "unit": "tokens/s",
            "ms": round(t * 1e3, 3), "seq_len": S, "batch": 1, "bound": "mfma",
            "achieved_TFLOPs": round(flops / t / 1e12, 1), "peak_TFLOPs": MFMA_PEAK_TFLOPS,
            "frac": round(flops / t / 1e12 / MFMA_PEAK_TFLOPS, 4),
            "attention": {"waves_per_workgroup": int(nv.lib().mv_paged_attn_prefill_form(
                              1, S, cfg.num_attention_heads, cfg.num_key_value_heads)),
                          "running_maximum_may_lag_by_pow2": float(os.environ.get("MVLLM_PREFILL_ATTN_RAISE") or 0)},
            "logits": "last position only (what the engine samples from)"}


def projection_timing(runner, bs, layers, reps=20):
    """Average duration of the projection launches of one decode step, issued exactly as the step
    issues them - with the norm prologues (bs <= 4), or with the slab reduction + norm launches
    (bs > 4), bias / SwiGLU / residual epilogues - back to back as one hipGraph, bracketed by HIP
    events on
    the stream the graph is launched on.  Attention, embedding and the token pick are left out, so the
    figure is the dominant kernel's own; the ~0.2 us between graph nodes is inside it."""
    import ctypes
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.linear import wgemm, wgemm_norm, wgemm_postnorm, wgemm_slabs

    tr = runner.model.transformer
    ws = runner._gemm_ws
    h, normed, resid = runner._hidden0[:bs], runner._normed[:bs], runner._residual[:bs]
    attn = runner._attn[:bs].reshape(bs, -1)

    def tp_pass():          # ModelGraphRunner._decode_tp without its attention and reduction launches
        slabs = getattr(runner._reducer, "takes_slabs", False)
        for li in range(layers):
            b = tr.h[li]
            if bs > 4 and runner.qkv_slabs:
                wgemm_slabs(normed, b.attn._qkv_w, ws)
            else:
                wgemm(normed, b.attn._qkv_w, runner._qkv[:bs], bias=b.attn._qkv_b, workspace=ws, plan_slabs=bs > 4)
            if slabs:
                wgemm_slabs(attn, b.attn._o, ws)
            else:
                wgemm(attn, b.attn._o, runner._attn_out[:bs], workspace=ws)
            wgemm(normed, b.mlp._gu, runner._swiglu_out[:bs], swiglu=True, workspace=ws)
            if slabs:
                wgemm_slabs(runner._swiglu_out[:bs], b.mlp._d, ws)
            else:
                wgemm(runner._swiglu_out[:bs], b.mlp._d, runner._attn_out[:bs], workspace=ws)
        if getattr(runner, "_gatherer", None) is not None:   # head cut by vocabulary rows
            wgemm(normed, runner._lm_head_shard, runner._logits_local[:bs], workspace=ws)
        else:
            wgemm(normed, runner._lm_head, runner._logits[:bs], workspace=ws)
        return 4 * layers + 1

    def one_pass():
        if runner.world_size > 1:
            return tp_pass()
        n = 0
        for li in range(layers):
            b = tr.h[li]
            if bs <= 4:     # ModelGraphRunner._layer_norm_fused
                wgemm_norm(h, b.ln_1.weight, b.ln_1.eps, b.attn._qkv_w, runner._qkv[:bs], bias=b.attn._qkv_b,
                           workspace=ws)
                wgemm(attn, b.attn._o, runner._attn_out[:bs], workspace=ws)
                wgemm_norm(runner._attn_out[:bs], b.ln_2.weight, b.ln_2.eps, b.mlp._gu, runner._swiglu_out[:bs],
                           add_r=h, sum_out=resid, swiglu=True, workspace=ws)
                wgemm(runner._swiglu_out[:bs], b.mlp._d, runner._mlp_out[:bs], residual=resid, workspace=ws)
            else:           # ModelGraphRunner._decode_batched
                if runner.qkv_slabs:    # the slabs are added up by the attention kernel (not timed here)
                    wgemm_slabs(normed, b.attn._qkv_w, ws)
                else:
                    wgemm(normed, b.attn._qkv_w, runner._qkv[:bs], bias=b.attn._qkv_b, workspace=ws, plan_slabs=True)
                wgemm_postnorm(attn, b.attn._o, resid, b.ln_2.weight, b.ln_2.eps, normed, residual=h, workspace=ws)
                wgemm(normed, b.mlp._gu, runner._swiglu_out[:bs], swiglu=True, workspace=ws)
                wgemm_postnorm(runner._swiglu_out[:bs], b.mlp._d, runner._mlp_out[:bs], b.ln_1.weight, b.ln_1.eps,
                               normed, residual=resid, var_from_rounded=True, workspace=ws)
            n += 4
        if getattr(runner, "_gatherer", None) is not None:   # several ranks, head cut by vocabulary rows
            wgemm(normed, runner._lm_head_shard, runner._logits_local[:bs], workspace=ws)
        elif bs <= 4:
            wgemm_norm(h, tr.ln_f.weight, tr.ln_f.eps, runner._lm_head, runner._logits[:bs], workspace=ws)
        else:
            wgemm(normed, runner._lm_head, runner._logits[:bs], workspace=ws)
        return n + 1

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        n_launch = one_pass()
        side.synchronize()
        nv.check("begin", nv.lib().mv_graph_begin_capture(side.cuda_stream))
        one_pass()
        h_ = ctypes.c_void_p()
        nv.check("end", nv.lib().mv_graph_end_capture(side.cuda_stream, ctypes.byref(h_)))
        for _ in range(3):
            nv.lib().mv_graph_launch(h_, side.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(reps):
            nv.lib().mv_graph_launch(h_, side.cuda_stream)
        e1.record(side)
        side.synchronize()
        nv.lib().mv_graph_destroy(h_)
    torch.cuda.current_stream().wait_stream(side)
    return e0.elapsed_time(e1) * 1e-3 / (reps * n_launch), n_launch


def roofline_block(runner, cfg, bs, layers, world, weight_bytes, step_bytes, step_seconds):
    """The `roofline` object of the bench line: the dominant kernel's achieved bytes per second from
    the in-run graph timing, the whole step's next to it, and the HBM traffic per launch from the
    committed counter passes (rocprofv3 --pmc cannot run inside this process), labelled with the file
    it comes from."""
    t_k, n_launch = projection_timing(runner, bs, layers)
    per_launch = weight_bytes / n_launch
    ach = per_launch / t_k / 1e9
    kname = "wgemm_dot_kernel" if bs <= 4 else "wgemm_mfma_kernel"
    traffic, source = None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            d = json.load(open(tf))
            key = f"{kname}:{cfg.name}:bs{bs}:tp{world}"
            traffic = d.get(key)
            if traffic is not None:
                source = (d.get(key + ":detail") or {}).get(
                    "source", "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, tools/pmc_traffic.py)")
        except Exception:
            traffic = None
    return {
        "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": source,
        "kernel": kname,
        "bytes_per_launch": int(per_launch),
        "avg_launch_us": round(t_k * 1e6, 3),
        "launches_per_step": n_launch,
        "timing": ("HIP events around a hipGraph of this rank's projection launches as the several-ranks step "
                   "issues them (k-slabs left for the reductions; the reduction + add + norm launches and "
                   "attention are not in it), measured in this run" if world > 1 else
                   "HIP events around a hipGraph of the step's projection launches as the step issues them "
                   "(norm prologues / slab-reduction norms and epilogues included), measured in this run"),
        "step_frac": round(step_bytes / step_seconds / 1e9 / HBM_PEAK_GBS, 4),
    }


def batch_leg(args, cfg, layers, model, dev, bs, ragged, steps=60, warmup=10):
    """Decode at another batch size on the same weights, inside the same run: its own runner
    (sharing the weight image), KV pool, captured graph, timed region and roofline object."""
    from micro_vllm_amd.core.cache_manager import KVCacheManager
    from micro_vllm_amd.core.layer.model_graph import ModelGraphRunner

    total_tokens = args.ctx + steps + warmup + 8
    blocks_per_seq = (total_tokens + 255) // 256 + 1
    runner = ModelGraphRunner(model, layers, cfg.num_attention_heads, cfg.head_dim, cfg.num_key_value_heads,
                              cfg.hidden_size, cfg.intermediate_size, dev, max_batch_size=bs,
                              max_blocks=max(32, blocks_per_seq), release_source_weights=True)
    mgr = KVCacheManager(bs * blocks_per_seq + 2, 256, layers, cfg.num_key_value_heads, cfg.head_dim,
                         dtype=torch.bfloat16, device=dev, max_tokens=blocks_per_seq * 256, max_batch_size=bs)
    order = list(mgr._free)
    random.Random(2).shuffle(order)
    mgr._free.clear()
    mgr._free.extend(order)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    mgr._k_store.normal_(0.0, 1.0, generator=g)
    mgr._v_store.normal_(0.0, 1.0, generator=g)
    rng = random.Random(3)
    ctxs = [rng.randint(min(256, args.ctx), args.ctx) if ragged else args.ctx for _ in range(bs)]
    for sid in range(bs):
        assert mgr.alloc(sid, ctxs[sid])[0], "KV pool too small"
    runner.capture(mgr, batch_sizes=[bs])
    seq_ids = list(range(bs))
    ids = torch.randint(0, cfg.vocab_size, (bs,), generator=torch.Generator().manual_seed(0)).to(dev)

    def step(ids, pos):
        for sid in seq_ids:
            mgr.append(sid)
        mgr.cache_batch_data(seq_ids, [c + pos for c in ctxs])
        return runner.greedy_ids(runner.forward(ids, mgr, bs))

    pos = 1
    for _ in range(warmup):
        ids = step(ids, pos)
        pos += 1
    torch.cuda.synchronize()
    pos0 = pos
    t0 = time.perf_counter()
    for _ in range(steps):
        ids = step(ids, pos)
        pos += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert bool(torch.isfinite(runner._logits[:bs].float()).all()), "non-finite logits"
    wbytes = cfg.weight_bytes(1)
    if layers != cfg.num_hidden_layers:
        import dataclasses
        wbytes = dataclasses.replace(cfg, num_hidden_layers=layers).weight_bytes(1)
    kv_tok = cfg.kv_bytes_per_token(1) * layers // cfg.num_hidden_layers
    mean_keys = sum(ctxs) / bs + (pos0 + pos0 + steps - 1) / 2.0 + 1.0
    step_bytes = wbytes + bs * (mean_keys + 1.0) * kv_tok
    out = {
        "metric": "decode tokens/sec", "value": round(bs * steps / dt, 1), "unit": "tokens/s",
        "batch": bs, "seq_len": args.ctx, "contexts": "uniform in [256, seq_len] (seed 3)" if ragged else "seq_len for every row",
        "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4),
        "graph": bs in runner._graphs,
        "uneven_split_graph": bs in runner._graphs_uneven and runner._rows_uneven(mgr, bs),
        "step_algorithmic_GB": round(step_bytes / 1e9, 3),
        "step_hbm_frac": round(step_bytes / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4),
    }
    try:
        out["roofline"] = roofline_block(runner, cfg, bs, layers, 1, wbytes, step_bytes, dt / steps)
    except Exception as e:
        log(f"[bench] bs={bs} roofline measurement failed: {e!r}")
        out["roofline"] = None
    del runner, mgr
    torch.cuda.empty_cache()
    return out


def shape_table(runner, bs, layers):
    """Per-projection streaming rate, cycling over the layers' distinct weights (no cache reuse)."""
    from micro_vllm_amd.kernel.linear import wgemm

    tr = runner.model.transformer
    rows = []
    specs = [
        ("qkv", lambda b: wgemm(runner._normed[:bs], b.attn._qkv_w, runner._qkv[:bs], workspace=runner._gemm_ws), lambda b: b.attn._qkv_w),
        ("o", lambda b: wgemm(runner._attn[:bs].reshape(bs, -1), b.attn._o, runner._attn_out[:bs], workspace=runner._gemm_ws), lambda b: b.attn._o),
        ("gate_up+swiglu", lambda b: wgemm(runner._normed[:bs], b.mlp._gu, runner._swiglu_out[:bs], swiglu=True, workspace=runner._gemm_ws), lambda b: b.mlp._gu),
        ("down", lambda b: wgemm(runner._swiglu_out[:bs], b.mlp._d, runner._mlp_out[:bs], workspace=runner._gemm_ws), lambda b: b.mlp._d),
    ]
    for name, fn, wof in specs:
        for li in range(layers):
            fn(tr.h[li])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 5
        for _ in range(reps):
            for li in range(layers):
                fn(tr.h[li])
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / (reps * layers)
        nbytes = wof(tr.h[0]).nbytes
        rows.append((name, nbytes, t))
        log(f"[shapes] {name:16s} {nbytes/1e6:8.1f} MB  {t*1e6:8.2f} us  {nbytes/t/1e9:8.1f} GB/s (eager launches, incl. gaps)")
    return rows


class _CharTokenizer:
    """One token per character, no end-of-sequence token among them (the engine leg needs a tokenizer
    object; there is no vocabulary file offline)."""
    eos_token_id = 1

    def __init__(self, vocab):
        self.vocab = vocab

    def encode(self, prompt, add_special_tokens=True):
        return [5 + (ord(c) * 7919) % (self.vocab - 5) for c in prompt]

    def decode(self, ids, skip_special_tokens=True):
        return " ".join(str(i) for i in ids)


def engine_leg(cfg, model, requests=32, prompt_tokens=512, new_tokens=128):
    """The continuous-batching configuration as the reference runs it: `requests` prompts through
    InferenceEngine.generate - scheduler, bucketed prefill batches under max_prefill_tokens, padded
    decode batches replaying captured graphs, greedy pick on the device, update_sequences on the host.
    Generated tokens over the wall time of the whole job (second of two repetitions)."""
    from micro_vllm_amd.core.engine import InferenceEngine

    blocks = requests * ((prompt_tokens + new_tokens + 255) // 256 + 1) + 4
    eng = InferenceEngine(model=model, tokenizer=_CharTokenizer(cfg.vocab_size), max_batch_size=requests,
                          max_prefill_tokens=max(2048, prompt_tokens), n_blocks=blocks,
                          max_tokens=prompt_tokens + new_tokens + 256,
                          graph_batch_sizes=[b for b in (1, 2, 4, 8, 16, 32) if b <= requests] or [requests],
                          prefill_timeout=0.0)
    prompts = ["".join(chr(33 + (i * 31 + j * 7) % 90) for j in range(prompt_tokens)) for i in range(requests)]
    steps = {"prefill": [0, 0.0], "decode": [0, 0.0]}
    inner = eng.step

    def timed_step(ctx):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = inner(ctx)
        torch.cuda.synchronize()
        k = steps[ctx.batch_type if ctx.batch_type in steps else "decode"]
        k[0] += 1
        k[1] += time.perf_counter() - t
        return r

    eng.step = timed_step
    for _ in range(2):
        for k in steps.values():
            k[0], k[1] = 0, 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.generate(prompts, max_tokens=new_tokens, temperature=0.0, top_p=0.9)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    ntok = sum(len(v.split()) for v in out.values()) - requests * prompt_tokens
    assert len(out) == requests and ntok > 0
    return {"metric": "generated tokens/sec through InferenceEngine.generate (scheduler + prefill batches + "
                      "padded decode batches, whole job)",
            "value": round(ntok / dt, 1), "unit": "tokens/s", "requests": requests,
            "prompt_tokens": prompt_tokens, "new_tokens": new_tokens, "max_batch_size": requests,
            "job_ms": round(dt * 1e3, 1), "generated": ntok,
            "prefill_steps": steps["prefill"][0], "prefill_ms": round(steps["prefill"][1] * 1e3, 1),
            "decode_steps": steps["decode"][0],
            "decode_ms_per_step": round(steps["decode"][1] / max(1, steps["decode"][0]) * 1e3, 3),
            "sampling": "greedy"}


def cpu_baseline(args, cfg, runner, dev):
    """The CPU restatement of the reference's decode path (oracle/ref_model.py) on the host cores:
    2 transformer layers + final norm + LM head of the same random model at the same context,
    median step time, extrapolated to the full depth."""
    from micro_vllm_amd.models.synthetic import SyntheticModel, _Attn, _Block, _Mlp, _Param, _Transformer
    from micro_vllm_amd.models import build_synthetic_model
    from oracle.ref_cache import RefKVCache
    from oracle.ref_model import RefRunner
    import dataclasses

    n_sample = 2
    torch.set_num_threads(os.cpu_count() or 1)
    cores = torch.get_num_threads()
    # same generator recipe as the device model, CPU side, sample layers only (seeded; the values
    # differ from the device model's, the shapes and work do not)
    small_vocab_cfg = cfg
    model = build_synthetic_model(small_vocab_cfg, "cpu", seed=1234, num_layers=n_sample)
    ref = RefRunner(model, cfg.head_dim, world=1, max_blocks=max(32, (args.ctx + 300) // 256 + 1),
                    fp32_weights=True)
    pages = (args.ctx + 2 + 255) // 256 + 1
    cache = RefKVCache(args.bs * pages + 1, 256, n_sample, cfg.num_key_value_heads, cfg.head_dim,
                       max_tokens=pages * 256, max_batch_size=args.bs, strict_reference=False)
    g = torch.Generator().manual_seed(1)
    for li in range(n_sample):
        cache.k_caches[li].normal_(0.0, 1.0, generator=g)
        cache.v_caches[li].normal_(0.0, 1.0, generator=g)
    for sid in range(args.bs):
        cache.alloc(sid, args.ctx)
    ids = torch.randint(0, cfg.vocab_size, (args.bs,), generator=torch.Generator().manual_seed(0))
    pos = args.ctx + 1
    t_layers, t_head = [], []
    for it in range(2 + 5):
        for sid in range(args.bs):
            cache.append(sid)
        bt, sl = cache.cache_batch_data(list(range(args.bs)), [pos] * args.bs)
        t0 = time.perf_counter()
        h = model.transformer.wte.weight[ids]
        for li in range(n_sample):
            h = ref.decode_layer(h, li, [(cache.k_caches, cache.v_caches)], sl, bt)
        t1 = time.perf_counter()
        ref.head(h)
        t2 = time.perf_counter()
        if it >= 2:
            t_layers.append((t1 - t0) / n_sample)
            t_head.append(t2 - t1)
        pos += 1
    step = statistics.median(t_layers) * cfg.num_hidden_layers + statistics.median(t_head)
    return {
        "value": round(args.bs / step, 4),
        "unit": "tokens/s",
        "cores": cores,
        "kind": "port",
        "sample": (f"CPU restatement of the reference decode path (oracle/ref_model.py, fp32 math on "
                   f"bf16-valued weights): {n_sample} of {cfg.num_hidden_layers} layers + LM head at "
                   f"ctx {args.ctx}, bs {args.bs}, median of 5 steps after 2 warm-up; per-layer time "
                   f"x{cfg.num_hidden_layers} + head = {step*1e3:.0f} ms/step"),
    }


def launch_ranks(args) -> int:
    """The parent of a several-GPU run started without a launcher: N child processes of this file,
    one per rank, as the reference is started (one process per GPU from RANK / WORLD_SIZE,
    core/parallel_config.py:21-32).  Nothing here initialises a GPU (counting devices does not),
    and the children are fresh processes, not re-executions of this one.  Rank 0's standard output
    (the JSON line) is passed through; everything else the ranks print goes to standard error.
    Returns the exit code: 0 only if every rank ended with 0; the first failure ends the others."""
    import socket
    import subprocess
    import threading

    n = args.gpus
    ndev = torch.cuda.device_count()
    share = os.environ.get("MVLLM_SHARE_GPU", "0") == "1"
    if ndev < 1:
        log("bench.py needs a GPU (the HIP path has no fallback)")
        return 2
    if n > ndev and not share:
        log(f"--gpus {n} but only {ndev} GPU(s) are visible; set MVLLM_SHARE_GPU=1 to rehearse "
            f"{n} ranks on fewer cards (no measure of scaling)")
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs, lines = [], []

    def pump(stream):
        for line in stream:
            lines.append(line)

    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if n > ndev:
            # RCCL refuses two ranks on one device: the rehearsal meets over gloo; the data path's
            # own exchange (mapped peer buffers) is the same either way
            env.setdefault("MVLLM_DIST_BACKEND", "gloo")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=True))
    reader = threading.Thread(target=pump, args=(procs[0].stdout,), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("MVLLM_BENCH_LAUNCH_TIMEOUT_S", "1700"))
    code = 0
    while True:
        states = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(states) if c not in (None, 0)]
        if bad:
            code = bad[0][1] if bad[0][1] > 0 else 1
            log(f"[bench] rank {bad[0][0]} ended with {bad[0][1]}; ending the other ranks")
            break
        if all(c == 0 for c in states):
            break
        if time.time() > deadline:
            log("[bench] the ranks did not finish in time; ending them")
            code = 124
            break
        time.sleep(0.2)
    if code:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5)
    sys.stdout.write("".join(lines))
    sys.stdout.flush()
    return code


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.as_rank_of:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus) and not args.as_rank_of:
        log(f"[bench] --gpus {args.gpus} under a launcher of {world} rank(s): running {world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
    local %= torch.cuda.device_count()  # rehearsals put several ranks on one GPU (gloo backend)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"

    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core import parallel_config as pc
    from micro_vllm_amd.core.layer.sampler import greedy

    pc.setup()
    peers = world > 1                 # ranks to synchronise with
    if args.as_rank_of:
        assert world == 1, "--as-rank-of runs one process"
        world = args.as_rank_of       # from here on: the group size the model is cut for
    t_build = time.time()
    cfg, layers, runner, mgr, ctxs = build(args, dev, world)
    bs = args.bs
    if not args.no_graph:
        if args.one_split_rule:
            runner.uneven_ratio = float("inf")
        captured = 1
        try:
            runner.capture(mgr, batch_sizes=[bs])
        except Exception as e:
            if not peers:
                raise
            log(f"[bench] rank {rank}: capture failed ({e!r})")
            captured = 0
        if peers:
            # every rank replays a graph or none does: a rank that decodes eagerly beside ranks that
            # replay would still issue the same reductions, but the timing would be of a mixture
            t = torch.tensor([captured], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                log(f"[bench] rank {rank}: a rank could not capture its step; every rank decodes eagerly")
                runner._graphs.clear()
                runner._graphs_uneven.clear()
                runner._is_graph_ready = True
    torch.cuda.synchronize()
    log(f"[bench] rank {rank}/{world}: {cfg.name} x{layers} layers built + captured in {time.time()-t_build:.1f}s")

    seq_ids = list(range(bs))
    ids = torch.randint(0, cfg.vocab_size, (bs,), generator=torch.Generator().manual_seed(0)).to(dev)
    next_ids = torch.zeros(bs, dtype=torch.long, device=dev)
    pos = 1  # added to a row's context: the engine's first decode position after a prompt

    def step(ids, pos):
        for sid in seq_ids:
            mgr.append(sid)
        mgr.cache_batch_data(seq_ids, [c + pos for c in ctxs])
        logits = runner.forward(ids, mgr, bs)
        return runner.greedy_ids(logits)    # the argmax is the last kernel of the captured step

    def sync_all():
        torch.cuda.synchronize()
        if peers:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ids = step(ids, pos)
        pos += 1
    sync_all()
    pos0 = pos
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ids = step(ids, pos)
        pos += 1
    torch.cuda.synchronize()
    if peers:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if peers:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(runner._logits[:bs].float()).all()), "non-finite logits"

    # ---- outside the timed region: single steps between HIP events (median / p99 latency) ----
    lat = []
    for _ in range(args.tail_steps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ids = step(ids, pos)
        e1.record()
        e1.synchronize()
        lat.append(e0.elapsed_time(e1))
        pos += 1
    lat.sort()
    # ---- and, with more than one rank, what the 2 x layers reductions of a step cost alone: the
    # runner's own reducer (one-shot all-reduce fused with add + RMSNorm, or torch.distributed + norm
    # kernels), issued back to back the way the step issues them ----
    allreduce_us, allreduce_kind = None, None
    if world > 1:
        try:
            red = runner._reducer
            allreduce_kind = type(red).__name__
            n = cfg.hidden_size
            part = torch.zeros(bs, n, dtype=torch.bfloat16, device=dev)
            resid, y, normed = (torch.zeros(bs, n, dtype=torch.bfloat16, device=dev) for _ in range(3))
            lnw = torch.ones(n, dtype=torch.bfloat16, device=dev)

            def ops(k):
                for _ in range(k):
                    red.addnorm(bs, n, partial=part, residual=resid, y=y, ln_weight=lnw, eps=1e-5, normed=normed)

            ops(4)
            sync_all()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops(2 * layers)
            e1.record()
            e1.synchronize()
            allreduce_us = e0.elapsed_time(e1) * 1e3 / (2 * layers)
            sync_all()
        except Exception as e:
            log(f"[bench] reduction timing failed: {e!r}")

    ms_per_step = dt / args.steps * 1e3
    tok_s = bs * args.steps / dt

    # ---- algorithmic bytes of one step on one GPU ---------------------------------------
    head_sharded = getattr(runner, "_gatherer", None) is not None   # LM head cut by vocabulary rows
    full_cfg_bytes = cfg.weight_bytes(world, head_sharded)
    if layers != cfg.num_hidden_layers:  # debug slices: count what was built
        import dataclasses
        full_cfg_bytes = dataclasses.replace(cfg, num_hidden_layers=layers).weight_bytes(world, head_sharded)
    kv_tok = cfg.kv_bytes_per_token(world) * layers // cfg.num_hidden_layers
    mean_keys = sum(ctxs) / bs + (pos0 + pos0 + args.steps - 1) / 2.0 + 1.0
    kv_bytes = bs * (mean_keys + 1.0) * kv_tok
    step_bytes = full_cfg_bytes + kv_bytes
    step_gbs = step_bytes / (dt / args.steps) / 1e9

    out = {
        "metric": "decode tokens/sec (paged-KV single-token decode, whole job)",
        "value": round(tok_s, 2),
        "unit": "tokens/s",
        "n_gpus": 1 if args.as_rank_of else world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic",
        "config": {
            "workload": (f"{cfg.name}-shaped random weights, TP={world} bf16, bs={bs}, "
                         f"seq_len={args.ctx}{' ragged: uniform in [256, seq_len]' if args.ragged else ''}"
                         f" paged decode (page 256, shuffled block table)"),
            "global_batch": bs,
            "seq_len": args.ctx,
            "layers": layers,
            "parallelism": (f"rank 0 of tp{world} alone on one GPU: no peers, every reduction sums one partial"
                            if args.as_rank_of else f"tp{world}"),
            "lm_head": ("cut by vocabulary rows over the ranks, logits all-gathered" if head_sharded
                        else "whole on every rank"),
            "graph": bs in runner._graphs,   # False when capture was skipped or fell back to eager
            # the step replayed the graph with the finer attention split (rows of uneven length)
            "uneven_split_graph": bs in runner._graphs_uneven and runner._rows_uneven(mgr, bs),
            "sampling": "greedy argmax on device",
        },
        "step_algorithmic_GB": round(step_bytes / 1e9, 3),
        "step_GBps": round(step_gbs, 1),
        "step_hbm_frac": round(step_gbs / HBM_PEAK_GBS, 4),
    }
    if lat:
        out["step_ms_median"] = round(lat[len(lat) // 2], 4)
        out["step_ms_p99"] = round(lat[min(len(lat) - 1, int(len(lat) * 0.99))], 4)
    if allreduce_us is not None:
        out["allreduce"] = {"us_per_op": round(allreduce_us, 2), "ops_per_step": 2 * layers,
                            "share_of_step": round(allreduce_us * 2 * layers / (ms_per_step * 1e3), 4),
                            "reducer": allreduce_kind,
                            "what": "[bs, hidden] sum over the ranks + residual add + RMSNorm, eager launches back to back"}

    if rank == 0:
        try:
            out["roofline"] = roofline_block(runner, cfg, bs, layers, world, full_cfg_bytes, step_bytes,
                                             dt / args.steps)
        except Exception as e:  # the headline number must still be printed
            log(f"[bench] roofline measurement failed: {e!r}")
            out["roofline"] = None
        if world == 1 and bs == 1 and not args.no_bs32 and args.model == "llama3-8b":
            # the other batch size the metric names, on the record of the same run
            out["bs32"] = {}
            for name, ragged in (("even", False), ("ragged", True)):
                try:
                    out["bs32"][name] = batch_leg(args, cfg, layers, runner.model, dev, 32, ragged)
                except Exception as e:
                    log(f"[bench] bs=32 {name} leg failed: {e!r}")
                    out["bs32"][name] = None
        if args.shapes:
            shape_table(runner, bs, layers)
        if world == 1 and not args.no_prefill:
            try:
                out["prefill"] = prefill_leg(args, cfg, layers, runner, dev)
            except Exception as e:
                log(f"[bench] prefill leg failed: {e!r}")
                out["prefill"] = None
        if world == 1 and bs == 1 and not args.no_engine and args.model == "llama3-8b":
            try:
                out["engine_bs32"] = engine_leg(cfg, runner.model)
            except Exception as e:
                log(f"[bench] engine leg failed: {e!r}")
                out["engine_bs32"] = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, cfg, runner, dev)
            except Exception as e:
                log(f"[bench] cpu baseline failed: {e!r}")
                out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if peers:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
