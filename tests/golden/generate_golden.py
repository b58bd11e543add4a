"""Generate the golden fixtures in this directory by RUNNING THE REFERENCE'S OWN CODE on CPU.

Needs the micro-vllm checkout at /root/reference (or $MVLLM_REFERENCE); it exists only in the build
container, so the outputs are committed and this script is never run by the tests.  What it
executes from the reference:
  * the Triton kernels of kernel/rmsnorm_add.py, kernel/rmsnorm_residual.py, kernel/swiglu.py,
    called directly under TRITON_INTERPRET=1 with CPU tensors;
  * core/paged_attention.py PrecomputedRotaryEmbedding, core/layer/rope.py RoPE.forward;
  * core/cache_manager.py KVCacheManager (device="cpu");
  * core/layer/model_graph.py ModelGraphRunner._mlp (static, eager);
  * core/scheduler.py Scheduler + core/sequence.py Sequence with a stub tokenizer;
  * core/layer/sampler.py Sampler._sample_impl (static, eager) under torch.manual_seed.
Only inputs and outputs (data) are stored; no reference source text.

Usage:  cd /tmp && python /root/repo/tests/golden/generate_golden.py
"""
import hashlib
import json
import os
import sys
import tempfile

os.environ["TRITON_INTERPRET"] = "1"
REF = os.environ.get("MVLLM_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
os.chdir(tempfile.mkdtemp(prefix="golden_"))  # the reference writes a log file into the cwd

import numpy as np  # noqa: E402
import torch  # noqa: E402
import triton  # noqa: E402
import triton.language as tl  # noqa: E402
from triton.runtime import interpreter as _interp  # noqa: E402

# The Triton interpreter narrows fp32 -> bf16 by TRUNCATING the mantissa (its cast_impl passes no
# rounding mode), which no GPU build of these kernels does: Triton's GPU lowering rounds to
# nearest even.  Replace that one conversion of the interpreter (a tool, not the reference) so
# the recorded outputs are what the reference's kernels compute on a GPU.
_orig_cast_impl = _interp.InterpreterBuilder.cast_impl


def _cast_impl_rne(self, src, dst_type):
    if src.dtype.scalar == tl.float32 and dst_type.scalar == tl.bfloat16:
        f = np.ascontiguousarray(src.data, dtype=np.float32)
        bits = torch.from_numpy(f).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
        return _interp.TensorHandle(bits.reshape(f.shape), dst_type.scalar)
    return _orig_cast_impl(self, src, dst_type)


_interp.InterpreterBuilder.cast_impl = _cast_impl_rne


class unpatched_interpreter:
    """The interpreter as it ships (fp32 -> bf16 by truncation) for the duration of a block: the
    same kernels are recorded a second time with it, so that the fixtures do not rest on the
    replaced conversion alone."""

    def __enter__(self):
        _interp.InterpreterBuilder.cast_impl = _orig_cast_impl

    def __exit__(self, *exc):
        _interp.InterpreterBuilder.cast_impl = _cast_impl_rne


def u16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def gen_norms():
    from kernel.rmsnorm_add import _rmsnorm_gemm_kernel, _rmsnorm_kernel
    from kernel.rmsnorm_residual import _rmsnorm_residual_fused_for_gemm_kernel

    out = {}
    cases = [(1, 128, 1e-6), (4, 2048, 1e-5), (2, 4096, 1e-6), (3, 320, 1e-6), (32, 256, 1e-5)]
    for ci, (rows, n, eps) in enumerate(cases):
        g = torch.Generator().manual_seed(ci)
        x = (torch.randn(rows, n, generator=g) * 1.5).to(torch.bfloat16)
        r = torch.randn(rows, n, generator=g).to(torch.bfloat16)
        w = (1.0 + 0.1 * torch.randn(n, generator=g)).to(torch.bfloat16)
        # decode rmsnorm_: kernel/rmsnorm_add.py:170-206 launch parameters
        y = torch.empty_like(x)
        _rmsnorm_gemm_kernel[(rows,)](x, y, w, x.stride(0), y.stride(0), n, eps,
                                      BLOCK_SIZE=min(triton.next_power_of_2(n), 2048))
        # prefill rmsnorm: kernel/rmsnorm_add.py:37-53
        y2 = torch.empty_like(x)
        _rmsnorm_kernel[(rows,)](x, y2, w, x.stride(0), y2.stride(0), n, eps,
                                 BLOCK_SIZE=min(triton.next_power_of_2(n), 8192))
        # add+rmsnorm: kernel/rmsnorm_residual.py:127-166
        yn, res = torch.empty_like(x), torch.empty_like(x)
        _rmsnorm_residual_fused_for_gemm_kernel[(rows,)](
            x, r, yn, res, w, x.stride(0), r.stride(0), yn.stride(0), res.stride(0), n, eps,
            BLOCK_SIZE=min(triton.next_power_of_2(n), 8192))
        out.update({f"c{ci}_x": u16(x), f"c{ci}_r": u16(r), f"c{ci}_w": u16(w),
                    f"c{ci}_eps": np.float64(eps), f"c{ci}_rmsnorm_decode": u16(y),
                    f"c{ci}_rmsnorm_prefill": u16(y2), f"c{ci}_addnorm_y": u16(yn),
                    f"c{ci}_addnorm_res": u16(res)})
        with unpatched_interpreter():   # the literal outputs of the interpreter as it ships
            ty, tyn, tres = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
            _rmsnorm_gemm_kernel[(rows,)](x, ty, w, x.stride(0), ty.stride(0), n, eps,
                                          BLOCK_SIZE=min(triton.next_power_of_2(n), 2048))
            _rmsnorm_residual_fused_for_gemm_kernel[(rows,)](
                x, r, tyn, tres, w, x.stride(0), r.stride(0), tyn.stride(0), tres.stride(0), n, eps,
                BLOCK_SIZE=min(triton.next_power_of_2(n), 8192))
        out.update({f"c{ci}_rmsnorm_decode_unpatched": u16(ty), f"c{ci}_addnorm_y_unpatched": u16(tyn),
                    f"c{ci}_addnorm_res_unpatched": u16(tres)})
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "norms.npz"), **out)


def gen_swiglu():
    from kernel.swiglu import _swiglu_kernel

    out = {}
    cases = [(1, 256), (4, 5632), (3, 1000), (32, 128)]
    for ci, (rows, d) in enumerate(cases):
        g = torch.Generator().manual_seed(100 + ci)
        gu = (torch.randn(rows, 2 * d, generator=g) * 2.0).to(torch.bfloat16)
        o = torch.empty(rows, d, dtype=torch.bfloat16)
        n_el = rows * d
        # kernel/swiglu.py:131-146 launch parameters
        _swiglu_kernel[(triton.cdiv(n_el, 4096),)](gu.view(-1), o.view(-1), n_el, stride_d=2 * d,
                                                  BLOCK_SIZE=4096)
        out.update({f"c{ci}_gate_up": u16(gu), f"c{ci}_out": u16(o)})
        with unpatched_interpreter():
            to = torch.empty(rows, d, dtype=torch.bfloat16)
            _swiglu_kernel[(triton.cdiv(n_el, 4096),)](gu.view(-1), to.view(-1), n_el, stride_d=2 * d,
                                                      BLOCK_SIZE=4096)
        out[f"c{ci}_out_unpatched"] = u16(to)
    out["n_cases"] = np.int64(len(cases))
    np.savez_compressed(os.path.join(OUT, "swiglu.npz"), **out)


def gen_rotary():
    from core.layer.rope import RoPE
    from core.paged_attention import PrecomputedRotaryEmbedding

    out = {}
    meta = {}
    for dim in (64, 128):
        emb = PrecomputedRotaryEmbedding(dim, max_position=32 * 256)
        cos = emb.cos_cache[0, 0, :, :dim // 2].contiguous()  # core/paged_attention.py:187-190
        sin = emb.sin_cache[0, 0, :, :dim // 2].contiguous()
        rows = list(range(0, 300)) + [511, 512, 1000, 2047, 2048, 4095, 4096, 8191]
        out[f"d{dim}_rows"] = np.array(rows, dtype=np.int64)
        out[f"d{dim}_cos_rows"] = u16(cos[rows])
        out[f"d{dim}_sin_rows"] = u16(sin[rows])
        meta[f"d{dim}"] = {
            "cos_sha256": hashlib.sha256(u16(cos).tobytes()).hexdigest(),
            "sin_sha256": hashlib.sha256(u16(sin).tobytes()).hexdigest(),
            "distinct_rows": int(torch.unique(cos.view(torch.int16), dim=0).shape[0]),
        }
        g = torch.Generator().manual_seed(200 + dim)
        q = torch.randn(2, 7, 4, dim, generator=g).to(torch.bfloat16)
        k = torch.randn(2, 7, 2, dim, generator=g).to(torch.bfloat16)
        qr, kr = RoPE().forward(q, k, cos, sin)
        out.update({f"d{dim}_q": u16(q), f"d{dim}_k": u16(k), f"d{dim}_q_rot": u16(qr),
                    f"d{dim}_k_rot": u16(kr)})
    np.savez_compressed(os.path.join(OUT, "rotary.npz"), **out)
    with open(os.path.join(OUT, "rotary_meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


def gen_cache_traces():
    from core.cache_manager import KVCacheManager

    traces = []

    def run(name, n_blocks, block_size, max_tokens, max_bs, script):
        mgr = KVCacheManager(n_blocks, block_size, n_layers=1, n_heads=1, head_size=8,
                             dtype=torch.float32, device="cpu", max_tokens=max_tokens,
                             max_batch_size=max_bs)
        events = []
        for op in script:
            kind = op[0]
            if kind == "alloc":
                ok, slots = mgr.alloc(op[1], op[2])
                events.append({"op": op, "ok": bool(ok),
                               "slots_head": slots[:4].tolist() if ok else None,
                               "slots_tail": slots[-4:].tolist() if ok else None})
            elif kind == "append":
                res = [mgr.append(op[1]) for _ in range(op[2])]
                events.append({"op": op, "slots": res if len(res) <= 8 else res[:4] + res[-4:],
                               "n_blocks": len(mgr.get_blocks(op[1]))})
            elif kind == "batch":
                bt, sl = mgr.cache_batch_data(op[1], op[2])
                events.append({"op": op, "block_table": bt.tolist(), "seqlens": sl.tolist()})
            elif kind == "free":
                mgr.free(op[1])
                events.append({"op": op, "free": len(mgr._free)})
            elif kind == "slots":
                events.append({"op": op, "slots": mgr.get_slots(op[1], op[2])})
            elif kind == "stats":
                events.append({"op": op, "stats": mgr.stats})
        traces.append({"name": name, "n_blocks": n_blocks, "block_size": block_size,
                       "max_tokens": max_tokens, "max_batch_size": max_bs, "events": events})

    for L in (1, 32, 255, 257, 4095):
        run(f"single_L{L}", 40, 256, 8192, 4, [
            ("alloc", 7, L), ("batch", [7], [L]), ("append", 7, 1), ("batch", [7], [L + 1]),
            ("append", 7, 300), ("batch", [7], [L + 301]), ("slots", 7, [0, L - 1, L, L + 5]),
            ("stats",), ("free", 7), ("stats",)])
    # lengths that are exact multiples of the block size: recorded as the reference behaves
    for L in (256, 4096):
        run(f"degenerate_L{L}", 40, 256, 8192, 4, [
            ("alloc", 3, L), ("batch", [3], [L]), ("append", 3, 2), ("batch", [3], [L + 2]),
            ("append", 3, 300), ("batch", [3], [L + 302])])
    # several sequences, free + reuse (non-contiguous block ids), duplicate rows, small pages
    run("interleaved", 12, 16, 128, 8, [
        ("alloc", 1, 20), ("alloc", 2, 5), ("alloc", 3, 33), ("batch", [1, 2, 3], [20, 5, 33]),
        ("append", 2, 12), ("append", 1, 3), ("batch", [2, 1, 2, 2], [17, 23, 17, 17]),
        ("free", 1), ("alloc", 4, 40), ("batch", [4, 3, 2], [40, 33, 17]), ("append", 4, 9),
        ("append", 3, 20), ("batch", [3, 4], [53, 49]), ("stats",), ("alloc", 5, 100),
        ("free", 2), ("free", 3), ("free", 4), ("stats",), ("alloc", 6, 17), ("batch", [6], [17])])
    with open(os.path.join(OUT, "cache_traces.json"), "w") as f:
        json.dump(traces, f)


def gen_mlp():
    from core.layer.model_graph import ModelGraphRunner

    g = torch.Generator().manual_seed(300)
    x = torch.randn(4, 128, generator=g).to(torch.bfloat16)
    gu = (torch.randn(128, 512, generator=g) * 0.05).to(torch.bfloat16)  # [in, 2I]: up | gate
    d = (torch.randn(256, 128, generator=g) * 0.05).to(torch.bfloat16)   # [I, hid]
    y = ModelGraphRunner._mlp(x, gu, d)
    np.savez_compressed(os.path.join(OUT, "mlp.npz"), x=u16(x), gu=u16(gu), d=u16(d), y=u16(y))


def gen_scheduler():
    import core.scheduler as sched_mod
    from core.scheduler import Scheduler
    from core.sequence import Sequence

    class Tok:
        eos_token_id = 2

        def encode(self, prompt, add_special_tokens=True):
            return [5 + (ord(c) % 50) for c in prompt]

    clock = [1000.0]
    sched_mod.time.time = lambda: clock[0]
    import core.sequence as seq_mod
    seq_mod.time.time = lambda: clock[0]

    tok = Tok()
    sch = Scheduler(max_batch_size=4, max_prefill_tokens=200, tokenizer=tok)
    log = []

    def snap(tag):
        batch, kind = sch.get_next_batch()
        log.append({"tag": tag, "t": clock[0], "type": kind, "seq_ids": [s.seq_id for s in batch],
                    "lens": [len(s.input_ids) for s in batch],
                    "padded_tail": [s.input_ids[-3:] for s in batch],
                    "running": [s.seq_id for s in sch.running_sequences],
                    "waiting": [s.seq_id for s in sch.waiting_queue]})
        return batch, kind

    lens = [10, 45, 52, 98, 103, 12]
    for i, L in enumerate(lens):
        sch.add_request(Sequence(i, "a" * L, tok, max_tokens=3))
        clock[0] += 0.001
    snap("before_timeout")
    clock[0] += 0.05
    b, kind = snap("after_timeout_bucket0")
    for s in b:
        s.update_state(7, None)
    b, kind = snap("next_bucket")
    for s in b:
        s.update_state(7, None)
    b, kind = snap("bucket100")
    for s in b:
        s.update_state(7, None)
    for step in range(4):
        b, kind = snap(f"decode{step}")
        seen = set()
        for s in b:
            if s.seq_id in seen:
                continue
            seen.add(s.seq_id)
            s.update_state(7 if step < 1 else 2, None)
    snap("drained")
    with open(os.path.join(OUT, "scheduler_trace.json"), "w") as f:
        json.dump({"lens": lens, "max_batch_size": 4, "max_prefill_tokens": 200, "log": log}, f)


def gen_sampler():
    """core/layer/sampler.py:21-43 run as it stands (the static method, eagerly): 40 000 draws per
    row of logits.  Individual draws depend on torch's generator, so what is stored is the count
    per vocabulary index - the empirical distribution - for rows that exercise each stage: an
    ordinary row, a row with equal logits inside the kept set and across the top-k boundary, and
    a top_p so small that only the first candidate survives."""
    from core.layer.sampler import Sampler

    n = 40000
    g = torch.Generator().manual_seed(400)
    rows = [
        ((torch.randn(64, generator=g) * 2).to(torch.bfloat16), 0.8, 0.9, 20),
        (torch.tensor([0.5, 3.0, 3.0, -1.0, 2.0, 2.0, 2.0, 0.25, 1.0, 1.0, 1.0, 1.0, -4.0, 0.0, 2.5, -2.0] * 2)
         .to(torch.bfloat16), 1.0, 0.97, 12),
        ((torch.randn(48, generator=g) * 1.5).to(torch.bfloat16), 0.7, 1e-6, 10),
    ]
    out = {"n_cases": np.int64(len(rows)), "draws": np.int64(n)}
    for ci, (logits, temp, top_p, top_k) in enumerate(rows):
        torch.manual_seed(500 + ci)
        ids = Sampler._sample_impl(logits.float()[None].expand(n, -1).contiguous(), torch.full((n,), temp),
                                   torch.full((n,), top_p), top_k)
        counts = torch.bincount(ids, minlength=logits.numel())
        out.update({f"c{ci}_logits": u16(logits), f"c{ci}_temperature": np.float64(temp),
                    f"c{ci}_top_p": np.float64(top_p), f"c{ci}_top_k": np.int64(top_k),
                    f"c{ci}_counts": counts.numpy().astype(np.int64),
                    f"c{ci}_support": torch.nonzero(counts).flatten().numpy().astype(np.int64)})
    np.savez_compressed(os.path.join(OUT, "sampler.npz"), **out)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for fn in (gen_norms, gen_swiglu, gen_rotary, gen_cache_traces, gen_mlp, gen_scheduler, gen_sampler):
        if only and fn.__name__ not in only:
            continue
        print("generating", fn.__name__, flush=True)
        fn()
    print("done ->", OUT)
