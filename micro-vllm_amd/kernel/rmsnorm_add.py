"""RMSNorm ops, same names and argument meaning as the reference's kernel/rmsnorm_add.py:
`rmsnorm` (:37, allocating, prefill), `rmsnorm_` (:170, writes a caller buffer, decode),
`rmsnorm_residual_fused` (:89) and the `ProRMSNorm` module (:116).  The arithmetic runs in
mv_rmsnorm / mv_rmsnorm_residual (csrc/norm.hip)."""
from typing import Tuple

import torch

from .. import _native as nv
from ._rows import as_rows, require_bf16_cuda


def rmsnorm_(x: torch.Tensor, weight: torch.Tensor, out_buffer: torch.Tensor,
             eps: float = 1e-6) -> torch.Tensor:
    """y = (x * rsqrt(mean(x^2) + eps)) * w into `out_buffer`; returns `out_buffer`."""
    assert x.is_cuda and weight.is_cuda and out_buffer.is_cuda, "All must be on CUDA"
    assert x.shape[-1] == weight.shape[0], "Hidden dim mismatch"
    assert x.shape == out_buffer.shape, "Output buffer shape mismatch"
    require_bf16_cuda(x, weight, out_buffer)
    hidden = x.shape[-1]
    x2, rows, sx = as_rows(x, hidden)
    y2, _, sy = as_rows(out_buffer, hidden)
    nv.check("mv_rmsnorm", nv.lib().mv_rmsnorm(
        x2.data_ptr(), weight.data_ptr(), y2.data_ptr(), rows, hidden, sx, sy, float(eps),
        nv.stream_ptr()))
    return out_buffer


def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    assert x.is_cuda, "Input must be on CUDA device"
    assert weight.is_cuda, "Weight must be on CUDA device"
    assert x.shape[-1] == weight.shape[0], f"Hidden dim mismatch: {x.shape[-1]} vs {weight.shape[0]}"
    return rmsnorm_(x, weight, torch.empty_like(x), eps)


def rmsnorm_residual_into(x, residual, weight, out_normed, out_residual, eps):
    """Shared body of every add+RMSNorm entry point (buffers supplied by the caller)."""
    assert x.shape == residual.shape, f"Shape mismatch: x={x.shape}, residual={residual.shape}"
    assert x.shape == out_normed.shape, "Output buffer shape mismatch"
    assert x.shape == out_residual.shape, "Output buffer shape mismatch"
    assert x.shape[-1] == weight.shape[0], "Hidden dim mismatch"
    require_bf16_cuda(x, residual, weight, out_normed, out_residual)
    hidden = x.shape[-1]
    x2, rows, sx = as_rows(x, hidden)
    r2, _, sr = as_rows(residual, hidden)
    y2, _, sy = as_rows(out_normed, hidden)
    o2, _, so = as_rows(out_residual, hidden)
    nv.check("mv_rmsnorm_residual", nv.lib().mv_rmsnorm_residual(
        x2.data_ptr(), r2.data_ptr(), weight.data_ptr(), y2.data_ptr(), o2.data_ptr(), rows, hidden,
        sx, sr, sy, so, float(eps), nv.stream_ptr()))
    return out_normed, out_residual


def rmsnorm_residual_fused(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                           eps: float = 1e-6) -> Tuple[torch.Tensor, torch.Tensor]:
    """(normed, x + residual), both freshly allocated."""
    assert x.is_cuda, "Input must be on CUDA device"
    assert residual.is_cuda, "Residual must be on CUDA device"
    assert weight.is_cuda, "Weight must be on CUDA device"
    return rmsnorm_residual_into(x, residual, weight, torch.empty_like(x), torch.empty_like(x), eps)


class ProRMSNorm(torch.nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6):
        super().__init__()
        self.hidden_size = hidden_size
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(hidden_size))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return rmsnorm(x, self.weight, self.eps)

    def extra_repr(self) -> str:
        return f"{self.hidden_size}, eps={self.eps}"
