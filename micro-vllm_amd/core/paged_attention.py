"""Rotary tables and the paged-attention call of the decode path.

Mirrors the pieces of the reference's core/paged_attention.py that its runners consume:
`PrecomputedRotaryEmbedding` (:42-63), `PagedAttention._cos_pool/_sin_pool/max_blocks`
(:151-190), and the `flash_attn_with_kvcache(...)` call (:281, core/layer/model_graph.py:161).
`flash_attn_with_kvcache` here has the third-party function's keyword signature for the
arguments the reference passes, and runs mv_paged_attn_decode (csrc/attn_decode.hip).
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import _native as nv


def rotary_tables(dim: int, max_position: int, base: float = 10000.0, device="cpu",
                  exact_positions: bool = False):
    """cos/sin tables [max_position, dim/2].

    Default reproduces the reference bit for bit: inverse frequencies AND positions are
    built in bfloat16 (core/paged_attention.py:50-52), so positions above 256 collapse onto
    the nearest representable bf16 integer.  `exact_positions=True` builds them in fp32
    and rounds only the final cos/sin to bf16.
    """
    # always built on the CPU and then moved: the table is an input of the parity contract,
    # and CPU/GPU transcendental ulps must not leak into it
    if exact_positions:
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
        t = torch.arange(max_position, dtype=torch.float32)
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        cos, sin = freqs.cos().to(torch.bfloat16), freqs.sin().to(torch.bfloat16)
    else:
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).to(torch.bfloat16) / dim))
        t = torch.arange(max_position, dtype=inv_freq.dtype)
        freqs = torch.einsum("i,j->ij", t, inv_freq)
        cos, sin = freqs.cos(), freqs.sin()
    return cos.contiguous().to(device), sin.contiguous().to(device)


class PrecomputedRotaryEmbedding(nn.Module):
    """cos_cache / sin_cache of shape [1, 1, max_position, dim] (both halves repeated)."""

    def __init__(self, dim, max_position=8192, base=10000, device=None, exact_positions=False):
        super().__init__()
        self.dim = dim
        self.device = device or torch.device("cpu")
        self.max_position = max_position
        cos, sin = rotary_tables(dim, max_position, base, self.device, exact_positions)
        self.register_buffer("cos_cache", torch.cat((cos, cos), dim=-1)[None, None].contiguous())
        self.register_buffer("sin_cache", torch.cat((sin, sin), dim=-1)[None, None].contiguous())

    def forward(self, x, positions):
        batch_size, num_heads, seq_len, head_size = x.shape
        positions = positions.to(self.device)
        cos = self.cos_cache[:, :, positions].view(batch_size, 1, seq_len, head_size)
        sin = self.sin_cache[:, :, positions].view(batch_size, 1, seq_len, head_size)
        x1, x2 = x[..., :self.dim // 2], x[..., self.dim // 2:]
        return x * cos + torch.cat((-x2, x1), dim=-1) * sin


class PagedAttention(nn.Module):
    """Owner of the rotary pools the runners pass to the attention kernel."""

    def __init__(self, num_heads: int, head_size: int, kv_num_heads: int, device: str = "auto",
                 max_batch_size=16, max_blocks=32, max_position=4096, max_tokens=8192,
                 block_size=256, exact_positions: bool = False, rope_base: float = 10000.0):
        super().__init__()
        self.block_size = block_size
        self.max_tokens = max_tokens
        self.max_seq_blocks = (max_tokens + block_size - 1) // block_size
        self.num_heads = num_heads
        self.kv_num_heads = kv_num_heads
        self.head_size = head_size
        self.scale = head_size ** -0.5
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if device != "auto":
            self.device = torch.device(device)
        self.max_batch_size = max_batch_size
        self.max_blocks = max_blocks
        max_kv_capacity = max_blocks * 256
        cos, sin = rotary_tables(head_size, max_kv_capacity, rope_base, self.device, exact_positions)
        self._cos_pool = cos.to(torch.bfloat16).contiguous()
        self._sin_pool = sin.to(torch.bfloat16).contiguous()


def _attn_workspace(device, nbytes):
    """Split-KV workspace of the current stream (zero-initialised: its head holds arrival tickets)."""
    return nv.stream_workspace("attn_decode", device, nbytes)


def paged_attn_decode(q, k_new, v_new, k_cache, v_cache, cache_seqlens, block_table,
                      rotary_cos=None, rotary_sin=None, softmax_scale=None, num_splits=0,
                      out=None, workspace=None):
    """q: [B,Hq,D]; k_new/v_new: [B,Hkv,D] (views into a fused qkv row are fine);
    caches [NB,page,Hkv,D]; returns out [B,Hq,D].  `workspace` (float32, only used with more than
    one split) must have been zero-filled when it was allocated: it starts with the arrival tickets
    of the split merge, which every launch returns to zero."""
    assert q.is_cuda and q.dtype == torch.bfloat16
    B, Hq, D = q.shape
    Hkv = k_new.shape[1]
    assert k_new.shape == (B, Hkv, D) and v_new.shape == (B, Hkv, D)
    assert q.stride(2) == 1 and q.stride(1) == D, "q heads must be contiguous"
    assert k_new.stride(2) == 1 and k_new.stride(1) == D and v_new.stride(1) == D
    assert k_new.stride(0) == v_new.stride(0), "k and v must share a row stride"
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape
    NB, page, ch, cd = k_cache.shape
    assert ch == Hkv and cd == D, "cache heads/dim mismatch"
    assert cache_seqlens.dtype == torch.int32 and block_table.dtype == torch.int32
    assert block_table.stride(1) == 1 and cache_seqlens.is_contiguous()
    L = nv.lib()
    if num_splits <= 0:
        num_splits = L.mv_paged_attn_decode_pick_splits(B, Hkv, D, block_table.shape[1] * page)
    if out is None:
        out = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=q.device)
    assert out.is_contiguous() and out.shape == (B, Hq, D)
    if workspace is None:
        workspace = _attn_workspace(q.device, L.mv_paged_attn_decode_workspace_bytes(B, Hq, D, num_splits))
    ws_ptr = workspace.data_ptr() if workspace is not None else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    scale = float(softmax_scale) if softmax_scale is not None else float(D) ** -0.5
    rot_max = rotary_cos.shape[0] if rotary_cos is not None else 0
    if rotary_cos is not None:
        assert rotary_cos.dtype == torch.bfloat16 and rotary_cos.shape[1] == D // 2
        assert rotary_cos.is_contiguous() and rotary_sin.is_contiguous()
    qsb = q.stride(0) if B > 1 else Hq * D
    ksb = k_new.stride(0) if B > 1 else Hkv * D
    nv.check("mv_paged_attn_decode", L.mv_paged_attn_decode(
        q.data_ptr(), qsb, k_new.data_ptr(), v_new.data_ptr(), ksb, k_cache.data_ptr(),
        v_cache.data_ptr(), nv.ptr(rotary_cos), nv.ptr(rotary_sin), rot_max,
        cache_seqlens.data_ptr(), block_table.data_ptr(), block_table.stride(0), out.data_ptr(),
        ws_ptr, ws_bytes, B, Hq, Hkv, D, page, NB, scale, num_splits, nv.stream_ptr()))
    return out


def qkv_attn_decode(x, ln_weight, eps, weight, bias, qkv, num_heads, k_cache, v_cache, cache_seqlens,
                    block_table, rotary_cos, rotary_sin, num_splits, out, workspace, sync_workspace,
                    status_word, softmax_scale=None):
    """RMSNorm + QKV projection + paged attention of a decode step in one launch
    (mv_qkv_attn_decode): `qkv` [B, (Hq + 2 Hkv) D] and `out` [B, Hq, D] end up holding what
    `wgemm_norm(x, ln_weight, eps, weight, qkv, bias=bias)` followed by `paged_attn_decode` on the
    columns of `qkv` would have written, bit for bit.  Returns False, having launched nothing, when
    the library does not fuse this shape (the caller then issues the two launches).
    `sync_workspace`: int32 tensor of mv_qkv_attn_workspace_bytes() bytes, zeroed once;
    `status_word`: pinned int32 host tensor the kernel writes if a wait inside it times out."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    B, K = x.shape
    NB, page, Hkv, D = k_cache.shape
    Hq = num_heads
    n = (Hq + 2 * Hkv) * D
    assert weight.n == n and weight.k == K and not weight.swiglu
    assert qkv.dtype == torch.bfloat16 and qkv.shape == (B, n) and qkv.stride(1) == 1
    assert out.is_contiguous() and out.shape == (B, Hq, D) and out.dtype == torch.bfloat16
    assert ln_weight.dtype == torch.bfloat16 and ln_weight.numel() == K and ln_weight.is_contiguous()
    assert bias is None or (bias.dtype == torch.bfloat16 and bias.numel() == n and bias.is_contiguous())
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape
    assert cache_seqlens.dtype == torch.int32 and block_table.dtype == torch.int32
    assert block_table.stride(1) == 1 and cache_seqlens.is_contiguous() and block_table.shape[0] == B
    assert rotary_cos is None or (rotary_cos.dtype == torch.bfloat16 and rotary_cos.shape[1] == D // 2
                                  and rotary_cos.is_contiguous() and rotary_sin.is_contiguous())
    assert status_word.dtype == torch.int32 and (status_word.is_pinned() or status_word.is_cuda)
    L = nv.lib()
    ldx = x.stride(0) if B > 1 else max(K, x.stride(0))
    ldq = qkv.stride(0) if B > 1 else max(n, qkv.stride(0))
    scale = float(softmax_scale) if softmax_scale is not None else float(D) ** -0.5
    rc = L.mv_qkv_attn_decode(
        x.data_ptr(), ldx, ln_weight.data_ptr(), float(eps), weight.tiles.data_ptr(), nv.ptr(bias),
        qkv.data_ptr(), ldq, B, n, K, k_cache.data_ptr(), v_cache.data_ptr(), nv.ptr(rotary_cos),
        nv.ptr(rotary_sin), rotary_cos.shape[0] if rotary_cos is not None else 0, cache_seqlens.data_ptr(),
        block_table.data_ptr(), block_table.stride(0), out.data_ptr(),
        workspace.data_ptr() if workspace is not None else None,
        workspace.numel() * workspace.element_size() if workspace is not None else 0,
        sync_workspace.data_ptr(), sync_workspace.numel() * sync_workspace.element_size(),
        status_word.data_ptr(), Hq, Hkv, D, page, NB, scale, int(num_splits), nv.stream_ptr())
    if rc == nv.MV_ERR_UNSUPPORTED:
        return False
    nv.check("mv_qkv_attn_decode", rc)
    return True


def paged_attn_decode_slabs(qkv_slabs, ksplit, qkv_bias, num_heads, k_cache, v_cache, cache_seqlens,
                            block_table, rotary_cos=None, rotary_sin=None, softmax_scale=None,
                            num_splits=0, out=None, workspace=None):
    """`paged_attn_decode` with q | k | v read from the k-slabs the QKV projection left
    (`wgemm_slabs`: fp32 [ksplit][B][(Hq + 2 Hkv) D] at the start of `qkv_slabs`), bias added on the
    way in: the projection's slab-reduction launch is saved and the result is the same bit for bit."""
    assert qkv_slabs.is_cuda and qkv_slabs.is_contiguous()
    B = cache_seqlens.shape[0]
    NB, page, Hkv, D = k_cache.shape
    Hq = num_heads
    n = (Hq + 2 * Hkv) * D
    assert qkv_slabs.numel() * qkv_slabs.element_size() >= ksplit * B * n * 4
    assert qkv_bias is None or (qkv_bias.dtype == torch.bfloat16 and qkv_bias.numel() == n and qkv_bias.is_contiguous())
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape
    assert cache_seqlens.dtype == torch.int32 and block_table.dtype == torch.int32
    assert block_table.stride(1) == 1 and cache_seqlens.is_contiguous() and block_table.shape[0] == B
    L = nv.lib()
    if num_splits <= 0:
        num_splits = L.mv_paged_attn_decode_pick_splits(B, Hkv, D, block_table.shape[1] * page)
    if out is None:
        out = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=k_cache.device)
    assert out.is_contiguous() and out.shape == (B, Hq, D)
    if workspace is None:
        workspace = _attn_workspace(k_cache.device, L.mv_paged_attn_decode_workspace_bytes(B, Hq, D, num_splits))
    ws_ptr = workspace.data_ptr() if workspace is not None else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    scale = float(softmax_scale) if softmax_scale is not None else float(D) ** -0.5
    rot_max = rotary_cos.shape[0] if rotary_cos is not None else 0
    if rotary_cos is not None:
        assert rotary_cos.dtype == torch.bfloat16 and rotary_cos.shape[1] == D // 2
        assert rotary_cos.is_contiguous() and rotary_sin.is_contiguous()
    nv.check("mv_paged_attn_decode_slabs", L.mv_paged_attn_decode_slabs(
        qkv_slabs.data_ptr(), ksplit, nv.ptr(qkv_bias), k_cache.data_ptr(), v_cache.data_ptr(),
        nv.ptr(rotary_cos), nv.ptr(rotary_sin), rot_max, cache_seqlens.data_ptr(), block_table.data_ptr(),
        block_table.stride(0), out.data_ptr(), ws_ptr, ws_bytes, B, Hq, Hkv, D, page, NB, scale,
        num_splits, nv.stream_ptr()))
    return out


def flash_attn_with_kvcache(q, k_cache, v_cache, k=None, v=None, rotary_cos=None, rotary_sin=None,
                            cache_seqlens=None, cache_batch_idx=None, block_table=None,
                            softmax_scale=None, causal=False, window_size=(-1, -1),
                            rotary_interleaved=True, alibi_slopes=None, num_splits=0, out=None):
    """Keyword-compatible stand-in for the third-party call the reference makes.

    Supports what the reference passes: paged caches with a block table, new k/v appended
    at `cache_seqlens`, NeoX (non-interleaved) rotary, no sliding window, no alibi.
    q: [B,Sq,Hq,D].  Sq == 1 runs the decode kernel; Sq > 1 the prefill kernel.
    """
    assert block_table is not None and cache_seqlens is not None, "paged cache only"
    assert cache_batch_idx is None and alibi_slopes is None
    assert tuple(window_size) == (-1, -1), "sliding window is not part of this path"
    assert k is not None and v is not None, "the new token's k/v are appended in-kernel"
    if rotary_cos is not None:
        assert not rotary_interleaved, "NeoX (rotate-half) layout only"
    B, Sq, Hq, D = q.shape
    if Sq == 1:
        o = paged_attn_decode(q[:, 0], k[:, 0], v[:, 0], k_cache, v_cache, cache_seqlens, block_table,
                              rotary_cos, rotary_sin, softmax_scale, num_splits,
                              out=None if out is None else out[:, 0])
        return o.unsqueeze(1)
    from .prefill_attention import paged_attn_prefill

    assert rotary_cos is None, "prefill applies RoPE outside (core/layer/rope.py)"
    return paged_attn_prefill(q, k, v, k_cache, v_cache, cache_seqlens, block_table,
                              softmax_scale=softmax_scale, causal=causal, out=out)
