"""Pin the CPU oracle against outputs of the reference's own code (tests/golden/, produced by
tests/golden/generate_golden.py from the micro-vllm checkout).  CPU only."""
import hashlib
import json
import os

import numpy as np
import torch

from conftest import bf16_from_u16, u16_of, ulp_diff
from oracle.ref_cache import RefKVCache
from oracle.ref_ops import (mlp_ref, rmsnorm_ref, rmsnorm_residual_ref, rope_prefill_ref,
                            rotary_tables_ref, swiglu_ref)

# The fixtures hold the reference kernels' results rounded to nearest even (see the generator).
# The interpreter sums with numpy in a different order than torch, so a value that sits on a
# rounding boundary may land one unit in the last place away; nearly all are bit-identical.
ULP_TOL = 1


def _frac_exact(a, b):
    return float((ulp_diff(a, b) == 0).float().mean())


def test_rmsnorm_matches_reference_triton(golden_dir):
    z = np.load(os.path.join(golden_dir, "norms.npz"))
    for c in range(int(z["n_cases"])):
        x, w = bf16_from_u16(z[f"c{c}_x"]), bf16_from_u16(z[f"c{c}_w"])
        eps = float(z[f"c{c}_eps"])
        got = rmsnorm_ref(x, w, eps)
        for variant in ("rmsnorm_decode", "rmsnorm_prefill"):
            want = bf16_from_u16(z[f"c{c}_{variant}"])
            assert ulp_diff(got, want).max() <= ULP_TOL, (c, variant)
            assert _frac_exact(got, want) > 0.995, (c, variant)


def test_add_rmsnorm_matches_reference_triton(golden_dir):
    z = np.load(os.path.join(golden_dir, "norms.npz"))
    for c in range(int(z["n_cases"])):
        x, r, w = (bf16_from_u16(z[f"c{c}_{k}"]) for k in ("x", "r", "w"))
        y, res = rmsnorm_residual_ref(x, r, w, float(z[f"c{c}_eps"]))
        # the residual sum is a single fp32 add + rounding: must be bit exact
        assert np.array_equal(u16_of(res), z[f"c{c}_addnorm_res"]), c
        want = bf16_from_u16(z[f"c{c}_addnorm_y"])
        assert ulp_diff(y, want).max() <= ULP_TOL, c
        assert _frac_exact(y, want) > 0.995, c


def test_swiglu_matches_reference_triton(golden_dir):
    z = np.load(os.path.join(golden_dir, "swiglu.npz"))
    for c in range(int(z["n_cases"])):
        gu = bf16_from_u16(z[f"c{c}_gate_up"])
        want = bf16_from_u16(z[f"c{c}_out"])
        got = swiglu_ref(gu)
        assert ulp_diff(got, want).max() <= ULP_TOL, c
        assert _frac_exact(got, want) > 0.99, c


def _truncate_bf16(t32):
    """fp32 -> bf16 by dropping the low 16 bits, what the Triton interpreter does as it ships."""
    return (t32.contiguous().view(torch.int32) & -65536).view(torch.float32).to(torch.bfloat16)


def test_oracle_against_the_unpatched_interpreter_outputs(golden_dir):
    """The fixtures above were recorded with the interpreter's fp32 -> bf16 conversion replaced by
    round-to-nearest-even (what a GPU build of the kernels does).  The same kernels' outputs from the
    interpreter AS IT SHIPS (it truncates) are stored next to them, so the pin does not rest on the
    replaced conversion: the oracle's rounded results are within one unit in the last place of them
    (three for add+norm's normalised output, whose input - the stored residual sum - is itself
    truncated), and the oracle's fp32 values, truncated the same way, reproduce them."""
    z = np.load(os.path.join(golden_dir, "norms.npz"))
    for c in range(int(z["n_cases"])):
        x, r, w = (bf16_from_u16(z[f"c{c}_{k}"]) for k in ("x", "r", "w"))
        eps, n = float(z[f"c{c}_eps"]), x.shape[-1]
        lit = bf16_from_u16(z[f"c{c}_rmsnorm_decode_unpatched"])
        assert ulp_diff(rmsnorm_ref(x, w, eps), lit).max() <= 1, c
        xf = x.float()
        y32 = (xf * torch.rsqrt((xf * xf).sum(-1, keepdim=True) / n + eps)) * w.float()
        t = _truncate_bf16(y32)
        assert ulp_diff(t, lit).max() <= 1 and _frac_exact(t, lit) > 0.995, c
        lit_y = bf16_from_u16(z[f"c{c}_addnorm_y_unpatched"])
        lit_res = bf16_from_u16(z[f"c{c}_addnorm_res_unpatched"])
        y, res = rmsnorm_residual_ref(x, r, w, eps)
        assert ulp_diff(res, lit_res).max() <= 1 and ulp_diff(y, lit_y).max() <= 3, c
        s32 = x.float() + r.float()
        res_t = _truncate_bf16(s32)
        assert torch.equal(res_t, lit_res), c                      # one fp32 add, truncated: bit exact
        rrms = torch.rsqrt((s32 * s32).sum(-1, keepdim=True) / n + eps)
        y_t = _truncate_bf16((res_t.float() * rrms) * w.float())
        assert ulp_diff(y_t, lit_y).max() <= 1 and _frac_exact(y_t, lit_y) > 0.995, c
    z = np.load(os.path.join(golden_dir, "swiglu.npz"))
    for c in range(int(z["n_cases"])):
        gu = bf16_from_u16(z[f"c{c}_gate_up"])
        lit = bf16_from_u16(z[f"c{c}_out_unpatched"])
        assert ulp_diff(swiglu_ref(gu), lit).max() <= 1, c
        d = gu.shape[-1] // 2
        up, gate = gu[..., :d].float(), gu[..., d:].float()
        t = _truncate_bf16((gate * torch.sigmoid(gate)) * up)
        assert ulp_diff(t, lit).max() <= 1 and _frac_exact(t, lit) > 0.99, c


def test_rotary_tables_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, "rotary.npz"))
    meta = json.load(open(os.path.join(golden_dir, "rotary_meta.json")))
    for dim in (64, 128):
        cos, sin = rotary_tables_ref(dim, 32 * 256)
        assert hashlib.sha256(u16_of(cos).tobytes()).hexdigest() == meta[f"d{dim}"]["cos_sha256"]
        assert hashlib.sha256(u16_of(sin).tobytes()).hexdigest() == meta[f"d{dim}"]["sin_sha256"]
        rows = z[f"d{dim}_rows"]
        assert np.array_equal(u16_of(cos[rows]), z[f"d{dim}_cos_rows"])
        assert np.array_equal(u16_of(sin[rows]), z[f"d{dim}_sin_rows"])
        # the bf16-position quirk: far fewer distinct rows than positions
        distinct = int(torch.unique(cos.view(torch.int16), dim=0).shape[0])
        assert distinct == meta[f"d{dim}"]["distinct_rows"] < 8192


def test_rope_prefill_bit_exact(golden_dir):
    z = np.load(os.path.join(golden_dir, "rotary.npz"))
    for dim in (64, 128):
        cos, sin = rotary_tables_ref(dim, 32 * 256)
        q = bf16_from_u16(z[f"d{dim}_q"])
        k = bf16_from_u16(z[f"d{dim}_k"])
        qr, kr = rope_prefill_ref(q, k, cos, sin)
        assert np.array_equal(u16_of(qr), z[f"d{dim}_q_rot"])
        assert np.array_equal(u16_of(kr), z[f"d{dim}_k_rot"])


def test_mlp_close_to_reference_eager(golden_dir):
    """The reference's eager `_mlp` rounds silu(gate) to bf16 before multiplying by up; the
    kernel route (and this oracle) round once.  Outputs agree within bf16 resolution."""
    z = np.load(os.path.join(golden_dir, "mlp.npz"))
    x, gu, d, y = (bf16_from_u16(z[k]) for k in ("x", "gu", "d", "y"))
    got = mlp_ref(x, gu.t().contiguous(), d.t().contiguous())
    err = (got.float() - y.float()).abs().max().item()
    assert err <= 1e-3 + 2 ** -7 * y.float().abs().max().item(), err


def _replay(trace, mgr, is_ref):
    """Drive a manager with a recorded script and compare every observable with the recording."""
    degenerate = trace["name"].startswith("degenerate")
    for ev in trace["events"]:
        op = ev["op"]
        kind = op[0]
        if kind == "alloc":
            ok, slots = mgr.alloc(op[1], op[2])
            assert bool(ok) == ev["ok"], (trace["name"], op)
            if ok:
                sl = slots if isinstance(slots, list) else slots.tolist()
                assert sl[:4] == ev["slots_head"] and sl[-4:] == ev["slots_tail"], (trace["name"], op)
        elif kind == "append":
            res = [mgr.append(op[1]) for _ in range(op[2])]
            want = ev["slots"]
            got = res if len(res) <= 8 else res[:4] + res[-4:]
            assert got == want, (trace["name"], op)
            nblk = len(mgr.blocks[op[1]]) if is_ref else len(mgr.get_blocks(op[1]))
            assert nblk == ev["n_blocks"], (trace["name"], op)
        elif kind == "batch":
            bt, sl = mgr.cache_batch_data(op[1], op[2])
            assert bt.tolist() == ev["block_table"], (trace["name"], op)
            assert sl.tolist() == ev["seqlens"], (trace["name"], op)
        elif kind == "free":
            mgr.free(op[1])
            nfree = len(mgr.free_blocks) if is_ref else len(mgr._free)
            assert nfree == ev["free"], (trace["name"], op)
        elif kind == "slots" and not is_ref:
            assert mgr.get_slots(op[1], op[2]) == ev["slots"], (trace["name"], op)
        elif kind == "stats" and not is_ref:
            assert mgr.stats == ev["stats"], (trace["name"], op)
    return degenerate


def test_ref_cache_traces_bit_exact(golden_dir):
    traces = json.load(open(os.path.join(golden_dir, "cache_traces.json")))
    assert len(traces) >= 8
    for tr in traces:
        mgr = RefKVCache(tr["n_blocks"], tr["block_size"], 1, 1, 8, max_tokens=tr["max_tokens"],
                         max_batch_size=tr["max_batch_size"], strict_reference=True,
                         dtype=torch.float32)
        _replay(tr, mgr, is_ref=True)
