"""SwiGLU ops named as in the reference's kernel/swiglu.py: `swiglu_gemm` (:100, caller
buffer) and `swiglu_fused` (:49, allocating).  Layout: gate_up[..., :d] = up,
gate_up[..., d:] = gate; out = silu(gate) * up.  Runs in mv_swiglu (csrc/elementwise.hip)."""
import torch

from .. import _native as nv


def swiglu_gemm(gate_up: torch.Tensor, out_buffer: torch.Tensor) -> torch.Tensor:
    assert gate_up.is_cuda and out_buffer.is_cuda, "All must be on CUDA"
    assert gate_up.dtype == out_buffer.dtype, f"Dtype mismatch: {gate_up.dtype} vs {out_buffer.dtype}"
    assert gate_up.dtype == torch.bfloat16, "bf16 only"
    last_dim = gate_up.shape[-1]
    assert last_dim % 2 == 0, "last dimension must hold up|gate halves"
    expected = gate_up.shape[:-1] + (last_dim // 2,)
    assert out_buffer.shape == expected, f"Shape mismatch: {out_buffer.shape} vs {expected}"
    assert out_buffer.is_contiguous(), "output buffer must be contiguous"
    gu = gate_up.contiguous()
    rows = gu.numel() // last_dim
    nv.check("mv_swiglu", nv.lib().mv_swiglu(gu.data_ptr(), out_buffer.data_ptr(), rows,
                                             last_dim // 2, nv.stream_ptr()))
    return out_buffer


def swiglu_fused(gate_up: torch.Tensor) -> torch.Tensor:
    assert gate_up.is_cuda, "Input must be on CUDA"
    out = torch.empty(gate_up.shape[:-1] + (gate_up.shape[-1] // 2,), dtype=gate_up.dtype,
                      device=gate_up.device)
    return swiglu_gemm(gate_up, out)
