"""Shared argument checks for the row-wise ops."""
import torch


def as_rows(t: torch.Tensor, hidden: int):
    """View `t` as [rows, hidden] without copying; return (tensor2d, rows, row_stride)."""
    assert t.shape[-1] == hidden, f"Hidden dim mismatch: {t.shape[-1]} vs {hidden}"
    assert t.stride(-1) == 1, "last dimension must be contiguous"
    t2 = t.view(-1, hidden) if t.dim() != 2 else t
    rows = t2.shape[0]
    stride = t2.stride(0) if rows > 1 else hidden
    return t2, rows, stride


def require_bf16_cuda(*tensors):
    for t in tensors:
        assert t.is_cuda, "All must be on CUDA"
        assert t.dtype == torch.bfloat16, f"bf16 only, got {t.dtype}"
