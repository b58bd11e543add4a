"""Decode-time projections: weights kept in the wave-tile layout the HIP skinny GEMM streams.

Stands in for the pre-transposed `[in,out]` weights plus `torch.matmul(..., out=)` calls of
the reference's decode runner (core/layer/model_graph.py:87-118, :145, :171, :183-185, :223).
"""
from typing import Optional

import torch

from .. import _native as nv


class StreamedWeight:
    """A Linear weight ([N,K], y = x @ W^T) relaid out once into 16x32 wave tiles.

    `swiglu=True` means rows are up/gate interleaved in groups of 8 (built by
    `from_up_gate`), so `wgemm(..., swiglu=True)` can apply SiLU(gate)*up in the epilogue.
    """

    def __init__(self, tiles: torch.Tensor, n: int, k: int, swiglu: bool = False):
        self.tiles, self.n, self.k, self.swiglu = tiles, n, k, swiglu

    @property
    def out_features(self) -> int:
        return self.n // 2 if self.swiglu else self.n

    @property
    def shape(self):
        """(in, out) like the reference's transposed weights, e.g. `_qkv_w.shape[1]`."""
        return (self.k, self.n)

    @property
    def nbytes(self) -> int:
        return self.n * self.k * 2

    @classmethod
    def from_linear(cls, w_nk: torch.Tensor) -> "StreamedWeight":
        assert w_nk.is_cuda and w_nk.dtype == torch.bfloat16 and w_nk.dim() == 2
        w_nk = w_nk.contiguous()
        n, k = w_nk.shape
        assert n % 16 == 0 and k % 32 == 0, f"weight {n}x{k}: need N%16==0 and K%32==0"
        tiles = torch.empty(n * k, dtype=torch.bfloat16, device=w_nk.device)
        nv.check("mv_shuffle_weight", nv.lib().mv_shuffle_weight(
            w_nk.data_ptr(), tiles.data_ptr(), n, k, nv.stream_ptr()))
        return cls(tiles, n, k, False)

    @classmethod
    def from_up_gate(cls, up: torch.Tensor, gate: torch.Tensor) -> "StreamedWeight":
        """up, gate: [I,K] each (the reference's mlp.w1 / mlp.w2 shards)."""
        assert up.shape == gate.shape and up.is_cuda and up.dtype == torch.bfloat16
        up, gate = up.contiguous(), gate.contiguous()
        i, k = up.shape
        assert i % 8 == 0 and k % 32 == 0
        inter = torch.empty(2 * i, k, dtype=torch.bfloat16, device=up.device)
        nv.check("mv_interleave_rows8", nv.lib().mv_interleave_rows8(
            up.data_ptr(), gate.data_ptr(), inter.data_ptr(), i, k, nv.stream_ptr()))
        sw = cls.from_linear(inter)
        sw.swiglu = True
        return sw


_workspaces = {}


def gemm_workspace(device, nbytes: int) -> Optional[torch.Tensor]:
    """One grow-only fp32 slab buffer per device and stream (k-split partial sums), so that
    projections issued on different streams never add into each other's slabs."""
    return nv.stream_workspace("wgemm", device, nbytes)


def wgemm(x: torch.Tensor, w: StreamedWeight, out: torch.Tensor,
          bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
          swiglu: bool = False, workspace: Optional[torch.Tensor] = None,
          plan_slabs: bool = False) -> torch.Tensor:
    """out = x @ W^T (+ bias) (+ residual), or swiglu(x @ W^T) when `swiglu`.

    x: [M,K] bf16, out: [M,N] bf16 (or fp32 for the un-rounded accumulators), M <= 64.
    Every optional term rounds to bf16 where the reference's separate ops round.
    `plan_slabs`: cut k as `wgemm_slabs` / `wgemm_postnorm` cut it for this shape (same sums, bit
    for bit, as the fused step that hands slabs to its next kernel).
    """
    assert x.is_cuda and out.is_cuda and x.dtype == torch.bfloat16
    assert x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1
    m, k = x.shape
    assert k == w.k, f"K mismatch: {k} vs {w.k}"
    assert swiglu == w.swiglu, "weight layout and epilogue disagree"
    ncols = w.out_features
    assert out.shape == (m, ncols), f"Output shape mismatch: {tuple(out.shape)} vs {(m, ncols)}"
    flags = nv.MV_GEMM_SWIGLU if swiglu else nv.MV_GEMM_DEFAULT
    if out.dtype == torch.float32:
        flags |= nv.MV_GEMM_OUT_F32
    else:
        assert out.dtype == torch.bfloat16
    if plan_slabs:
        flags |= nv.MV_GEMM_PLAN_SLABS
    if bias is not None:
        assert bias.dtype == torch.bfloat16 and bias.numel() == w.n and bias.is_contiguous()
    ld_res = 0
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.shape == (m, w.n) and residual.stride(1) == 1
        ld_res = residual.stride(0) if m > 1 else w.n
    L = nv.lib()
    if workspace is None:
        workspace = gemm_workspace(x.device, L.mv_wgemm_workspace_bytes(m, w.n, k))
    ws_ptr = workspace.data_ptr() if workspace is not None else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    ldx = x.stride(0) if m > 1 else max(k, x.stride(0))
    ldy = out.stride(0) if m > 1 else max(ncols, out.stride(0))
    nv.check("mv_wgemm", L.mv_wgemm(
        x.data_ptr(), ldx, w.tiles.data_ptr(), nv.ptr(bias), nv.ptr(residual), ld_res,
        out.data_ptr(), ldy, m, w.n, k, ws_ptr, ws_bytes, flags, nv.stream_ptr()))
    return out


def wgemm_slabs(x: torch.Tensor, w: StreamedWeight, workspace: torch.Tensor) -> int:
    """x @ W^T left in `workspace` as fp32 slabs [slab][M][N] (their sum, in slab order, is the
    product); returns the slab count.  For row-parallel projections whose output goes straight into
    the tensor-parallel reduction (`IpcAllReduce.addnorm`), which adds the slabs itself."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    m, k = x.shape
    assert k == w.k and not w.swiglu
    L = nv.lib()
    need = L.mv_wgemm_postnorm_workspace_bytes(m, w.n, k)
    assert workspace is not None and workspace.numel() * workspace.element_size() >= need
    ldx = x.stride(0) if m > 1 else max(k, x.stride(0))
    nv.check("mv_wgemm", L.mv_wgemm(
        x.data_ptr(), ldx, w.tiles.data_ptr(), None, None, 0, None, 0, m, w.n, k,
        workspace.data_ptr(), workspace.numel() * workspace.element_size(), nv.MV_GEMM_SLABS,
        nv.stream_ptr()))
    return L.mv_wgemm_slab_count(m, w.n, k)


def wgemm_postnorm(x: torch.Tensor, w: StreamedWeight, out: torch.Tensor, ln_weight: torch.Tensor,
                   eps: float, normed: torch.Tensor, bias: Optional[torch.Tensor] = None,
                   residual: Optional[torch.Tensor] = None, var_from_rounded: bool = False,
                   workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = x @ W^T (+ bias) (+ residual) as `wgemm`, and normed = rmsnorm(out) * ln_weight in the
    same pair of launches (the norm rides on the slab reduction).  Without `var_from_rounded` the
    variance is taken from the unrounded bf16(x @ W^T + bias) + residual: bit-identical to
    `wgemm` (no residual) followed by `rmsnorm_residual_gemm`; with it, from `out` as stored:
    bit-identical to `wgemm(..., residual=)` followed by `rmsnorm_`."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    m, k = x.shape
    assert k == w.k and not w.swiglu
    n = w.n
    assert out.shape == (m, n) and out.dtype == torch.bfloat16 and out.stride(1) == 1
    assert normed.shape == (m, n) and normed.dtype == torch.bfloat16 and normed.stride(1) == 1
    assert ln_weight.dtype == torch.bfloat16 and ln_weight.numel() == n and ln_weight.is_contiguous()
    if bias is not None:
        assert bias.dtype == torch.bfloat16 and bias.numel() == n and bias.is_contiguous()
    ld_res = 0
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.shape == (m, n) and residual.stride(1) == 1
        ld_res = residual.stride(0) if m > 1 else n
    L = nv.lib()
    need = L.mv_wgemm_postnorm_workspace_bytes(m, n, k)
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = gemm_workspace(x.device, need)
    ldx = x.stride(0) if m > 1 else max(k, x.stride(0))
    ldy = out.stride(0) if m > 1 else max(n, out.stride(0))
    ldn = normed.stride(0) if m > 1 else max(n, normed.stride(0))
    nv.check("mv_wgemm_postnorm", L.mv_wgemm_postnorm(
        x.data_ptr(), ldx, w.tiles.data_ptr(), nv.ptr(bias), nv.ptr(residual), ld_res,
        out.data_ptr(), ldy, ln_weight.data_ptr(), eps, normed.data_ptr(), ldn,
        nv.MV_NORM_VAR_FROM_ROUNDED if var_from_rounded else 0, m, n, k,
        workspace.data_ptr(), workspace.numel() * workspace.element_size(), nv.stream_ptr()))
    return out


def gemm_prefill(x: torch.Tensor, w: StreamedWeight, out: torch.Tensor,
                 bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                 swiglu: bool = False, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Same contract as `wgemm` for any number of rows, on the matrix cores (the prefill
    runner's torch.matmul calls, core/layer/model_prefill.py:42,54,58,62).  `workspace`: fp32
    buffer for k-slices (callers that are stream-captured pass their own fixed one; otherwise the
    grow-only buffer of the device and stream is used; a caller's own must have been zero-filled when it
    was allocated and be used for nothing else)."""
    assert x.is_cuda and out.is_cuda and x.dtype == torch.bfloat16
    assert x.dim() == 2 and out.dim() == 2 and x.stride(1) == 1 and out.stride(1) == 1
    m, k = x.shape
    assert k == w.k, f"K mismatch: {k} vs {w.k}"
    assert swiglu == w.swiglu, "weight layout and epilogue disagree"
    ncols = w.out_features
    assert out.shape == (m, ncols), f"Output shape mismatch: {tuple(out.shape)} vs {(m, ncols)}"
    flags = nv.MV_GEMM_SWIGLU if swiglu else nv.MV_GEMM_DEFAULT
    if out.dtype == torch.float32:
        flags |= nv.MV_GEMM_OUT_F32
    else:
        assert out.dtype == torch.bfloat16
    if bias is not None:
        assert bias.dtype == torch.bfloat16 and bias.numel() == w.n and bias.is_contiguous()
    ld_res = 0
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.shape == (m, w.n) and residual.stride(1) == 1
        ld_res = residual.stride(0) if m > 1 else w.n
    ldx = x.stride(0) if m > 1 else max(k, x.stride(0))
    ldy = out.stride(0) if m > 1 else max(ncols, out.stride(0))
    L = nv.lib()
    # k-slices for short prompts, accumulator slots for cut tiles, else 0.  The buffer starts with counters
    # that stay zero between launches, so it is one of its own (zero-filled when allocated), not the
    # single-token projections' slab buffer.
    need = L.mv_gemm_prefill_workspace_bytes(m, w.n, k)
    ws = (workspace if workspace is not None and workspace.numel() * 4 >= need
          else nv.stream_workspace("gemm_prefill", x.device, need))
    nv.check("mv_gemm_prefill_ws", L.mv_gemm_prefill_ws(
        x.data_ptr(), ldx, w.tiles.data_ptr(), nv.ptr(bias), nv.ptr(residual), ld_res,
        out.data_ptr(), ldy, m, w.n, k, flags, ws.data_ptr() if ws is not None else None,
        ws.numel() * 4 if ws is not None else 0, nv.stream_ptr()))
    return out


def wgemm_norm(x: torch.Tensor, ln_weight: torch.Tensor, eps: float, w: StreamedWeight,
               out: torch.Tensor, add_r: Optional[torch.Tensor] = None,
               sum_out: Optional[torch.Tensor] = None, var_from_rounded: bool = False,
               bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
               swiglu: bool = False, workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """wgemm on rmsnorm(x [+ add_r]) * ln_weight in ONE launch (M <= 4); `sum_out` receives
    bf16(x + add_r).  Bit-identical to rmsnorm_ / rmsnorm_residual_gemm followed by wgemm."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    m, k = x.shape
    assert m <= 4, "fused norm prologue is built for M <= 4"
    assert k == w.k and ln_weight.numel() == k and ln_weight.dtype == torch.bfloat16
    assert swiglu == w.swiglu
    ncols = w.out_features
    assert out.shape == (m, ncols) and out.stride(1) == 1
    flags = nv.MV_GEMM_SWIGLU if swiglu else nv.MV_GEMM_DEFAULT
    if out.dtype == torch.float32:
        flags |= nv.MV_GEMM_OUT_F32
    assert (add_r is None) == (sum_out is None)
    ld_r = ld_s = 0
    if add_r is not None:
        assert add_r.shape == x.shape and sum_out.shape == x.shape
        assert add_r.dtype == torch.bfloat16 and sum_out.dtype == torch.bfloat16
        ld_r = add_r.stride(0) if m > 1 else k
        ld_s = sum_out.stride(0) if m > 1 else k
    ld_res = 0
    if residual is not None:
        assert residual.dtype == torch.bfloat16 and residual.shape == (m, w.n)
        ld_res = residual.stride(0) if m > 1 else w.n
    L = nv.lib()
    if workspace is None:
        workspace = gemm_workspace(x.device, L.mv_wgemm_workspace_bytes(m, w.n, k))
    ws_ptr = workspace.data_ptr() if workspace is not None else None
    ws_bytes = workspace.numel() * workspace.element_size() if workspace is not None else 0
    ldx = x.stride(0) if m > 1 else max(k, x.stride(0))
    ldy = out.stride(0) if m > 1 else max(ncols, out.stride(0))
    nv.check("mv_wgemm_norm", L.mv_wgemm_norm(
        x.data_ptr(), ldx, nv.ptr(add_r), ld_r, ln_weight.data_ptr(), float(eps), nv.ptr(sum_out),
        ld_s, nv.MV_NORM_VAR_FROM_ROUNDED if var_from_rounded else 0, w.tiles.data_ptr(),
        nv.ptr(bias), nv.ptr(residual), ld_res, out.data_ptr(), ldy, m, w.n, k, ws_ptr, ws_bytes,
        flags, nv.stream_ptr()))
    return out


def embedding(table: torch.Tensor, ids: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out[r] = table[ids[r]] (model.transformer.wte(input_ids), model_graph.py:196)."""
    assert table.is_cuda and ids.is_cuda and out.is_cuda
    assert table.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and ids.dtype == torch.long
    assert table.is_contiguous() and out.is_contiguous() and ids.is_contiguous()
    rows = ids.numel()
    vocab, n = table.shape
    assert out.shape[-1] == n and out.numel() == rows * n
    nv.check("mv_embedding", nv.lib().mv_embedding(
        table.data_ptr(), ids.data_ptr(), out.data_ptr(), rows, n, vocab, nv.stream_ptr()))
    return out


def add_(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a += b in bf16 (b may be a bias row broadcast over a's rows)."""
    assert a.is_cuda and b.is_cuda and a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    assert a.is_contiguous() and b.is_contiguous()
    nv.check("mv_add", nv.lib().mv_add(a.data_ptr(), b.data_ptr(), a.data_ptr(), a.numel(),
                                       b.numel(), nv.stream_ptr()))
    return a
