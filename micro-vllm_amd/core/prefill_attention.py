"""Prefill-side attention ops: rotate + append a chunk's keys/values to the paged cache and causal
attention of the chunk over the cache.  Together they stand in for `RoPE.forward` plus the
multi-token `flash_attn_with_kvcache` call of core/layer/model_prefill.py:45-52."""
from typing import Optional

import torch

from .. import _native as nv


def rope_store_prefill(qkv: torch.Tensor, B: int, S: int, num_heads: int, kv_num_heads: int,
                       head_size: int, k_cache: torch.Tensor, v_cache: torch.Tensor,
                       cache_seqlens: torch.Tensor, block_table: torch.Tensor,
                       rotary_cos: Optional[torch.Tensor], rotary_sin: Optional[torch.Tensor]):
    """qkv: [B*S, (Hq+2Hkv)*D] rows `q heads | k heads | v heads`.  Rotates q and k in place at
    positions cache_seqlens[b] + i and writes k, v into their pages.  Returns qkv."""
    assert qkv.is_cuda and qkv.dtype == torch.bfloat16 and qkv.dim() == 2 and qkv.stride(1) == 1
    assert qkv.shape == (B * S, (num_heads + 2 * kv_num_heads) * head_size)
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape
    NB, page, ch, cd = k_cache.shape
    assert ch == kv_num_heads and cd == head_size, "cache heads/dim mismatch"
    assert cache_seqlens.dtype == torch.int32 and block_table.dtype == torch.int32
    assert cache_seqlens.numel() >= B and block_table.shape[0] >= B and block_table.stride(1) == 1
    rot_max = 0
    if rotary_cos is not None:
        assert rotary_cos.dtype == torch.bfloat16 and rotary_cos.shape[1] == head_size // 2
        assert rotary_cos.is_contiguous() and rotary_sin.is_contiguous()
        rot_max = rotary_cos.shape[0]
    nv.check("mv_rope_store_prefill", nv.lib().mv_rope_store_prefill(
        qkv.data_ptr(), qkv.stride(0), nv.ptr(rotary_cos), nv.ptr(rotary_sin), rot_max,
        cache_seqlens.data_ptr(), block_table.data_ptr(), block_table.stride(0),
        k_cache.data_ptr(), v_cache.data_ptr(), B, S, num_heads, kv_num_heads, head_size, page, NB,
        nv.stream_ptr()))
    return qkv


def paged_attn_prefill_rows(q_rows: torch.Tensor, B: int, S: int, num_heads: int, head_size: int,
                            k_cache: torch.Tensor, v_cache: torch.Tensor, cache_seqlens: torch.Tensor,
                            block_table: torch.Tensor, out_rows: torch.Tensor,
                            softmax_scale: Optional[float] = None) -> torch.Tensor:
    """q_rows: [B*S, >=Hq*D] (a column slice of the fused qkv buffer is fine); the cache already
    holds the chunk's keys and values.  out_rows: [B*S, Hq*D]."""
    assert q_rows.is_cuda and q_rows.dtype == torch.bfloat16 and q_rows.stride(1) == 1
    assert out_rows.dtype == torch.bfloat16 and out_rows.stride(1) == 1
    assert q_rows.shape[0] == B * S and out_rows.shape[0] == B * S
    NB, page, hkv, cd = k_cache.shape
    assert cd == head_size and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert cache_seqlens.dtype == torch.int32 and block_table.dtype == torch.int32
    scale = float(softmax_scale) if softmax_scale is not None else float(head_size) ** -0.5
    nv.check("mv_paged_attn_prefill", nv.lib().mv_paged_attn_prefill(
        q_rows.data_ptr(), q_rows.stride(0), k_cache.data_ptr(), v_cache.data_ptr(),
        cache_seqlens.data_ptr(), block_table.data_ptr(), block_table.stride(0), out_rows.data_ptr(),
        out_rows.stride(0), B, S, num_heads, hkv, head_size, page, NB, scale, nv.stream_ptr()))
    return out_rows


def paged_attn_prefill(q, k, v, k_cache, v_cache, cache_seqlens, block_table, softmax_scale=None,
                       causal=True, out=None):
    """The multi-token form of flash_attn_with_kvcache: q [B,S,Hq,D], k/v [B,S,Hkv,D] (already
    rotated); appends k, v at cache_seqlens[b] + i and returns causal attention [B,S,Hq,D]."""
    assert causal, "the prefill path is causal (core/layer/model_prefill.py:51)"
    B, S, Hq, D = q.shape
    Hkv = k.shape[2]
    assert k.shape == (B, S, Hkv, D) and v.shape == (B, S, Hkv, D)
    NB, page, _, _ = k_cache.shape
    # slot of every new token: page table lookup on the device (no host round trip)
    pos = cache_seqlens[:B].to(torch.long)[:, None] + torch.arange(S, device=q.device)[None, :]
    # positions past the last block-table column have no page: their tokens are dropped (slot -1),
    # never looked up (an index past the row would fault inside gather)
    col = pos // page
    inside = (col < block_table.shape[1]) & (pos >= 0)
    pages = torch.gather(block_table[:B].to(torch.long), 1, col.clamp(0, block_table.shape[1] - 1))
    slots = torch.where(inside & (pages >= 0), pages * page + pos % page, torch.full_like(pages, -1))
    from .cache_manager import store_kvcache
    store_kvcache(k.reshape(B * S, Hkv, D).contiguous(), v.reshape(B * S, Hkv, D).contiguous(),
                  k_cache, v_cache, slots.reshape(-1).to(torch.int32), page)
    q_rows = q.reshape(B * S, Hq * D)
    if q_rows.stride(1) != 1 or q_rows.stride(0) % 8:
        q_rows = q_rows.contiguous()
    if out is None:
        out = torch.empty(B, S, Hq, D, dtype=torch.bfloat16, device=q.device)
    assert out.is_contiguous() and out.shape == (B, S, Hq, D)
    paged_attn_prefill_rows(q_rows, B, S, Hq, D, k_cache, v_cache, cache_seqlens[:B].contiguous(),
                            block_table[:B], out.view(B * S, Hq * D), softmax_scale)
    return out
