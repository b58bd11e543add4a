"""Rotary embedding as the prefill runner applies it (reference: core/layer/rope.py:4-44)."""
import torch

from ... import _native as nv


class RoPE:
    """`forward(q, k, cos, sin)` with q, k [B,S,H,D] and half-width tables [>=S, D/2]: NeoX halves
    at positions 0..S-1, evaluated like the reference's bf16 tensor expression (each product and
    the sum round to bf16, core/layer/rope.py:29-33)."""

    def forward(self, q, k, cos, sin):
        return self._rotate(q, cos, sin), self._rotate(k, cos, sin)

    @staticmethod
    def _rotate(x, cos, sin):
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4
        B, S, H, D = x.shape
        assert cos.shape[-1] == D // 2, "half-width rotary tables only"
        assert cos.dtype == torch.bfloat16 and cos.is_contiguous() and sin.is_contiguous()
        assert cos.shape[0] >= S
        rows = x.reshape(B * S, H * D)
        if rows.stride(1) != 1 or rows.stride(0) % 8:
            rows = rows.contiguous()
        out = torch.empty(B * S, H * D, dtype=torch.bfloat16, device=x.device)
        nv.check("mv_rope_prefill", nv.lib().mv_rope_prefill(
            rows.data_ptr(), rows.stride(0), out.data_ptr(), out.stride(0), cos.data_ptr(),
            sin.data_ptr(), cos.shape[0], B * S, S, H, D, nv.stream_ptr()))
        return out.view(B, S, H, D)
