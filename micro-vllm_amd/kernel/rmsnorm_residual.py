"""Add+RMSNorm ops named as in the reference's kernel/rmsnorm_residual.py:
`rmsnorm_residual_gemm` (:127, two caller buffers - the decode path),
`rmsnorm_residual_fused` (:44, allocating) and `rmsnorm_residual_hybrid` (:169)."""
from typing import Tuple

import torch

from .rmsnorm_add import rmsnorm_residual_fused, rmsnorm_residual_into  # noqa: F401


def rmsnorm_residual_gemm(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                          out_normed_buffer: torch.Tensor, out_residual_buffer: torch.Tensor,
                          eps: float = 1e-6) -> Tuple[torch.Tensor, torch.Tensor]:
    """out_residual = x + residual; out_normed = rmsnorm(out_residual) * weight."""
    assert x.is_cuda and residual.is_cuda and weight.is_cuda, "All must be on CUDA"
    assert out_normed_buffer.is_cuda and out_residual_buffer.is_cuda, "Buffers must be on CUDA"
    assert out_normed_buffer.dtype == x.dtype, f"Dtype mismatch: {out_normed_buffer.dtype} vs {x.dtype}"
    assert out_residual_buffer.dtype == x.dtype, f"Dtype mismatch: {out_residual_buffer.dtype} vs {x.dtype}"
    return rmsnorm_residual_into(x, residual, weight, out_normed_buffer, out_residual_buffer, eps)


def rmsnorm_residual_hybrid(x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor,
                            out_normed_buffer: torch.Tensor,
                            eps: float = 1e-6) -> Tuple[torch.Tensor, torch.Tensor]:
    """Caller supplies only the normed buffer; the residual output is allocated."""
    assert x.is_cuda and residual.is_cuda and weight.is_cuda, "All must be on CUDA"
    assert out_normed_buffer.is_cuda, "Buffer must be on CUDA"
    assert out_normed_buffer.dtype == x.dtype, "Dtype mismatch"
    return rmsnorm_residual_into(x, residual, weight, out_normed_buffer, torch.empty_like(x), eps)
