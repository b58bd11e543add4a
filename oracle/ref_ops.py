"""CPU restatement of the reference's element-wise ops (test infrastructure; see oracle/__init__.py).

All functions take and return CPU torch tensors; bf16 in, fp32 math, bf16 out with the rounding
points of the reference kernels.
"""
import torch

BF16 = torch.bfloat16


def rmsnorm_ref(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    """kernel/rmsnorm_add.py:133-167 (_rmsnorm_gemm_kernel) and :9-34 (_rmsnorm_kernel):
    fp32 sum of squares, rrms = rsqrt(sum/N + eps), y = (x * rrms) * w, one bf16 rounding."""
    xf = x.float()
    n = x.shape[-1]
    rrms = torch.rsqrt((xf * xf).sum(dim=-1, keepdim=True) / n + eps)
    return ((xf * rrms) * w.float()).to(BF16)


def rmsnorm_residual_ref(x: torch.Tensor, r: torch.Tensor, w: torch.Tensor, eps: float):
    """kernel/rmsnorm_residual.py:74-124 (and kernel/rmsnorm_add.py:56-86): s = x + r in fp32;
    RES_OUT = bf16(s) (:108); the variance uses the unrounded s (:109); the second pass reloads
    the ROUNDED RES_OUT (:119) and y = (s~ * rrms) * w rounded to bf16."""
    s = x.float() + r.float()
    n = x.shape[-1]
    res = s.to(BF16)
    rrms = torch.rsqrt((s * s).sum(dim=-1, keepdim=True) / n + eps)
    y = ((res.float() * rrms) * w.float()).to(BF16)
    return y, res


def swiglu_ref(gate_up: torch.Tensor) -> torch.Tensor:
    """kernel/swiglu.py:6-46: up = first half, gate = second half (:35-37);
    out = bf16((gate * sigmoid(gate)) * up) in fp32 (:40-45)."""
    d = gate_up.shape[-1] // 2
    up, gate = gate_up[..., :d].float(), gate_up[..., d:].float()
    return ((gate * torch.sigmoid(gate)) * up).to(BF16)


def rotary_tables_ref(dim: int, max_position: int, base: float = 10000.0):
    """core/paged_attention.py:50-62 + :187-190: inv_freq AND positions in bfloat16, cos/sin of
    the bf16 outer product; the pools keep the first dim/2 columns.  Returns (cos, sin) bf16
    [max_position, dim/2]."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).to(BF16) / dim))
    t = torch.arange(max_position, dtype=inv_freq.dtype)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    cos, sin = emb.cos(), emb.sin()
    return cos[:, :dim // 2].contiguous(), sin[:, :dim // 2].contiguous()


def rope_prefill_ref(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor):
    """core/layer/rope.py:10-39 (half-width table branch :28-33): positions 0..S-1, NeoX halves,
    evaluated with bf16 tensor ops (each multiply and the add/sub round to bf16)."""
    seq_len = q.shape[1]
    c = cos[:seq_len][None, :, None, :]
    s = sin[:seq_len][None, :, None, :]
    q1, q2 = q.chunk(2, dim=-1)
    k1, k2 = k.chunk(2, dim=-1)
    q_rot = torch.cat([q1 * c - q2 * s, q2 * c + q1 * s], dim=-1)
    k_rot = torch.cat([k1 * c - k2 * s, k2 * c + k1 * s], dim=-1)
    return q_rot, k_rot


def linear_ref(x: torch.Tensor, w_out_in: torch.Tensor, bias=None) -> torch.Tensor:
    """torch.matmul(x, W_in_out, out=bf16 buffer) of core/layer/model_graph.py:145,171,183,185:
    exact bf16 products, fp32 accumulation, one rounding; `.add_(bias)` (:147) rounds again.
    `w_out_in` is the [out,in] weight (the reference multiplies by its transpose); it may be
    fp32 (pre-converted once) or bf16."""
    y = (x.float() @ w_out_in.float().t()).to(BF16)
    if bias is not None:
        y = (y.float() + bias.float()).to(BF16)
    return y


def mlp_ref(x: torch.Tensor, w_gu_out_in: torch.Tensor, w_d_out_in: torch.Tensor) -> torch.Tensor:
    """core/layer/model_graph.py:183-185: gate_up = x @ Wgu (bf16) -> swiglu -> @ Wd (bf16)."""
    gu = linear_ref(x, w_gu_out_in)
    return linear_ref(swiglu_ref(gu), w_d_out_in)
