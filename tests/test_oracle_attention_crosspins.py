"""The attention oracle (oracle/ref_attention.py) is a restatement of a third-party contract and
has no vector of its own to be checked against ("parity unpinned").  Its PIECES, however, can each
be tied to something that is pinned or independently implemented:

  * the NeoX rotation        -> outputs of the reference's own RoPE.forward (core/layer/rope.py:10-39)
                                recorded in tests/golden/rotary.npz;
  * where a token's K/V land -> slot numbers the reference's own KVCacheManager reported
                                (alloc / get_slots, core/cache_manager.py:351-399, :563-614) recorded in
                                tests/golden/cache_traces.json;
  * gather + softmax(QK^T)V  -> torch.nn.functional.scaled_dot_product_attention in fp32 on rows
                                gathered with plain indexing, bottom-right causal mask, GQA by
                                repeating kv heads.
What stays unpinned is the reading of the contract itself (which positions are rotated, where the
mask is anchored, which kv head a query head reads); the third check restates those choices a
second time in different code, which guards against slips, not against a shared misreading."""
import json
import math
import os

import numpy as np
import torch

from oracle.ref_attention import _rotate_neox, flash_attn_with_kvcache_ref
from oracle.ref_cache import RefKVCache

BF16 = torch.bfloat16


def _bf(a):
    return torch.from_numpy(a.view(np.int16).copy()).view(BF16)


def _ulps(a, b):
    """Distance in bf16 units of the larger magnitude's binade."""
    a, b = a.float(), b.float()
    scale = torch.maximum(a.abs(), b.abs()).clamp_min(2.0 ** -126)
    unit = torch.exp2(torch.floor(torch.log2(scale)) - 7)
    return (a - b).abs() / unit


def test_rotation_equals_the_references_rope_forward(golden_dir):
    """Both rotate NeoX halves; the reference multiplies in bf16 tensors (each product and the sum
    rounded), the oracle in fp32 with one rounding (what flash-attn's kernel does): at most one bf16
    unit apart, and mostly equal."""
    d = np.load(os.path.join(golden_dir, "rotary.npz"))
    for dim in (64, 128):
        rows = d[f"d{dim}_rows"].tolist()
        cos, sin = _bf(d[f"d{dim}_cos_rows"]), _bf(d[f"d{dim}_sin_rows"])
        assert rows[:7] == list(range(7))
        for name in ("q", "k"):
            x, want = _bf(d[f"d{dim}_{name}"]), _bf(d[f"d{dim}_{name}_rot"])
            S = x.shape[1]
            got = _rotate_neox(x, cos[:S][None, :, None, :], sin[:S][None, :, None, :])
            u = _ulps(got, want)
            # products of two bf16 numbers rounded before the subtraction can cancel: allow the unit of
            # the operands, not of the (smaller) result, where that happens
            big = x.float().abs().amax(dim=-1, keepdim=True) * 2.0 ** -7
            assert bool(((u <= 1.0) | ((got.float() - want.float()).abs() <= big)).all()), (dim, name)
            assert float((got != want).float().mean()) < 0.25, (dim, name)
            assert torch.equal(got[:, 0], x[:, 0]), "position 0 is the identity (cos 1, sin 0)"


def test_append_lands_on_the_slot_the_references_cache_manager_reports(golden_dir):
    """Replay the reference's allocator traces on the oracle's allocator (itself pinned bit-exact to
    them), then let the attention oracle append a token at position p through the block table and
    look it up at the flat slot number the REFERENCE reported for p."""
    traces = json.load(open(os.path.join(golden_dir, "cache_traces.json")))
    checked = 0
    for tr in traces:
        page = tr["block_size"]
        mgr = RefKVCache(tr["n_blocks"], page, 1, 1, 8, max_tokens=tr["max_tokens"],
                         max_batch_size=tr["max_batch_size"], strict_reference=True, dtype=BF16)
        kc, vc = mgr.k_caches[0], mgr.v_caches[0]

        def put_and_find(seq_id, pos, slot, tag):
            bt, _ = mgr.cache_batch_data([seq_id], [pos])
            if slot < 0 or int(bt[0, pos // page]) < 0:
                return 0
            k = torch.full((1, 1, 1, 8), float(1 + (tag % 200)), dtype=BF16)
            v = -k
            q = torch.zeros(1, 1, 1, 8, dtype=BF16)
            flash_attn_with_kvcache_ref(q, kc, vc, k, v, cache_seqlens=torch.tensor([pos], dtype=torch.int32),
                                        block_table=bt.clone(), p_bf16=False)
            assert torch.equal(kc.view(-1, 1, 8)[slot], k[0, 0]), (tr["name"], seq_id, pos, slot)
            assert torch.equal(vc.view(-1, 1, 8)[slot], v[0, 0]), (tr["name"], seq_id, pos, slot)
            return 1

        for ev in tr["events"]:
            op = ev["op"]
            if op[0] == "alloc":
                ok, _ = mgr.alloc(op[1], op[2])
                if ok and ev["ok"]:
                    L = op[2]
                    for pos, slot in list(zip(range(4), ev["slots_head"])) + \
                            list(zip(range(max(L - 4, 0), L), ev["slots_tail"][-min(4, L):])):
                        if pos < L:
                            checked += put_and_find(op[1], pos, slot, checked)
            elif op[0] == "append":
                for _ in range(op[2]):
                    mgr.append(op[1])
            elif op[0] == "free":
                mgr.free(op[1])
            elif op[0] == "slots":
                for pos, slot in zip(op[2], ev["slots"]):
                    checked += put_and_find(op[1], pos, slot, checked)
    assert checked >= 60


def test_gather_and_softmax_equal_sdpa_on_plainly_gathered_rows():
    """flash_attn_with_kvcache_ref without rotation against torch's own attention: rows gathered by
    flat slot = page * page_size + offset, kv heads repeated G times, mask anchored bottom right."""
    g = torch.Generator().manual_seed(11)
    page, Hq, Hkv, D = 256, 4, 2, 64
    for B, Sq, lens in ((2, 1, [300, 17]), (2, 5, [250, 511]), (1, 7, [0])):
        nblk = 8
        kc = torch.randn(nblk, page, Hkv, D, generator=g).to(BF16)
        vc = torch.randn(nblk, page, Hkv, D, generator=g).to(BF16)
        perm = torch.randperm(nblk, generator=g).to(torch.int32)
        bt = torch.full((B, 4), -1, dtype=torch.int32)
        for b in range(B):
            need = (lens[b] + Sq + page - 1) // page
            bt[b, :need] = perm[b * 3:b * 3 + need]
        q = torch.randn(B, Sq, Hq, D, generator=g).to(BF16)
        k = torch.randn(B, Sq, Hkv, D, generator=g).to(BF16)
        v = torch.randn(B, Sq, Hkv, D, generator=g).to(BF16)
        sl = torch.tensor(lens, dtype=torch.int32)
        out = flash_attn_with_kvcache_ref(q, kc, vc, k, v, cache_seqlens=sl, block_table=bt, causal=True,
                                          p_bf16=False)
        for b in range(B):
            total = lens[b] + Sq
            t = torch.arange(total)
            slots = bt[b, t // page].long() * page + t % page
            K = kc.view(-1, Hkv, D)[slots].float()            # [T, Hkv, D], new rows included
            V = vc.view(-1, Hkv, D)[slots].float()
            assert torch.equal(K[lens[b]:], k[b].float()) and torch.equal(V[lens[b]:], v[b].float())
            Kh = K.repeat_interleave(Hq // Hkv, dim=1).permute(1, 0, 2)   # [Hq, T, D]
            Vh = V.repeat_interleave(Hq // Hkv, dim=1).permute(1, 0, 2)
            mask = t[None, :] <= (lens[b] + torch.arange(Sq))[:, None]     # bottom-right aligned
            want = torch.nn.functional.scaled_dot_product_attention(
                q[b].float().permute(1, 0, 2), Kh, Vh, attn_mask=mask, scale=1.0 / math.sqrt(D))
            want = want.permute(1, 0, 2)
            err = (out[b].float() - want).abs()
            assert bool((err <= 1e-5 + 2.0 ** -8 * want.abs()).all()), (B, Sq, b, float(err.max()))
