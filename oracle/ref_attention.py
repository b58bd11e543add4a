"""CPU restatement of flash_attn_with_kvcache as the reference calls it (test infrastructure).

PARITY UNPINNED: the arithmetic belongs to the third-party flash-attn package
(requirements.txt:21 `flash-attn>=2.0.0`, not installed here, not vendored in the reference),
and the reference holds no golden vector for it.  This file restates the package's published
contract for the arguments used at core/layer/model_graph.py:161-168 and
core/layer/model_prefill.py:49-52:

  * k_cache / v_cache: [num_blocks, page, Hkv, D], addressed through block_table [B, max_blocks]
    (int32; logical block i of row b lives in physical page block_table[b, i]);
  * the Sq new tokens of row b get positions cache_seqlens[b] + 0..Sq-1; when rotary tables are
    given, q and the new k are rotated (rotary_interleaved=False: NeoX halves) at those positions,
    in fp32, and rounded to bf16; k (rotated) and v are written into the cache in place;
  * causal=True aligns the mask to the bottom right: query i sees keys 0..cache_seqlens[b]+i;
  * query head h reads kv head h // (Hq / Hkv);
  * S = q.k^T in fp32, softmax in fp32 with scale 1/sqrt(D) by default, output bf16.
    `p_bf16=True` (default) additionally rounds P to bf16 before P.V while the normaliser sums
    the unrounded P - what the package's tensor-core kernels do internally (P is an MMA operand).
    That rounding is an implementation detail, not part of the contract: its realisation depends
    on the kernel's tile order, so two correct kernels differ by up to
    2^-9 * sum_t p_t |v_t| per element.  `p_bf16=False` is the exact-softmax form used for the
    tight kernel checks.
"""
import math

import torch

BF16 = torch.bfloat16


def _rotate_neox(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """x: [..., D] bf16; cos/sin: broadcastable [..., D/2] bf16 -> bf16."""
    d = x.shape[-1]
    x1, x2 = x[..., :d // 2].float(), x[..., d // 2:].float()
    c, s = cos.float(), sin.float()
    return torch.cat([x1 * c - x2 * s, x2 * c + x1 * s], dim=-1).to(BF16)


def flash_attn_with_kvcache_ref(q, k_cache, v_cache, k, v, rotary_cos=None, rotary_sin=None,
                                cache_seqlens=None, block_table=None, softmax_scale=None,
                                causal=True, p_bf16=True, return_p_bound=False):
    """q: [B,Sq,Hq,D]; k, v: [B,Sq,Hkv,D]; caches mutated in place; returns [B,Sq,Hq,D] bf16."""
    B, Sq, Hq, D = q.shape
    Hkv = k.shape[2]
    G = Hq // Hkv
    page = k_cache.shape[1]
    scale = softmax_scale if softmax_scale is not None else 1.0 / math.sqrt(D)
    out = torch.empty(B, Sq, Hq, D, dtype=BF16)
    bound = torch.zeros(B, Sq, Hq, D) if return_p_bound else None
    for b in range(B):
        p0 = int(cache_seqlens[b])
        pos = torch.arange(p0, p0 + Sq)
        qb, kb, vb = q[b], k[b], v[b]  # [Sq,H,D]
        if rotary_cos is not None:
            c = rotary_cos[pos][:, None, :]
            s = rotary_sin[pos][:, None, :]
            qb = _rotate_neox(qb, c, s)
            kb = _rotate_neox(kb, c, s)
        # append the new tokens
        for i in range(Sq):
            t = p0 + i
            pg = int(block_table[b, t // page])
            k_cache[pg, t % page] = kb[i]
            v_cache[pg, t % page] = vb[i]
        total = p0 + Sq
        nblk = (total + page - 1) // page
        pages = block_table[b, :nblk].long()
        K = k_cache[pages].reshape(nblk * page, Hkv, D)[:total].float()  # [T,Hkv,D]
        V = v_cache[pages].reshape(nblk * page, Hkv, D)[:total].float()
        for h in range(Hq):
            kh, vh = K[:, h // G], V[:, h // G]
            s_ = (qb[:, h].float() @ kh.t()) * scale  # [Sq,T]
            if causal:
                keys = torch.arange(total)[None, :]
                limit = (p0 + torch.arange(Sq))[:, None]
                s_ = s_.masked_fill(keys > limit, float("-inf"))
            m = s_.max(dim=-1, keepdim=True).values
            p = torch.exp(s_ - m)
            l = p.sum(dim=-1, keepdim=True)
            pv = p.to(BF16).float() if p_bf16 else p
            o = (pv @ vh) / l
            out[b, :, h] = o.to(BF16)
            if return_p_bound:
                bound[b, :, h] = (p @ vh.abs()) / l * 2.0 ** -9
    return (out, bound) if return_p_bound else out
