"""Token selection on the device (csrc/sampler.hip).

`greedy` = mv_argmax, the temperature -> 0 limit of the reference's Sampler
(core/layer/sampler.py:4-43).  `sample` = mv_sample_topk_topp: the reference's whole recipe
(temperature, top-k, softmax, top-p, draw) in one launch, the draw being the inverse CDF of a
uniform number per row.  `Sampler` keeps the reference's call signature and uses `sample` for
bf16 logits on the GPU; `_sample_impl` is the reference's recipe in torch ops, kept as the
statistical comparison point."""
import torch

from ... import _native as nv


def greedy(logits: torch.Tensor, out: torch.Tensor = None, workspace: torch.Tensor = None) -> torch.Tensor:
    """logits [bs,V] bf16 -> int64 [bs] (first index on ties), no host synchronisation.
    `workspace`: zero-initialised float32 buffer of mv_argmax_workspace_bytes() where the parts of a
    long row meet (default: one per device and stream)."""
    assert logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2
    assert logits.stride(1) == 1
    bs, v = logits.shape
    if out is None:
        out = torch.empty(bs, dtype=torch.long, device=logits.device)
    assert out.dtype == torch.long and out.is_contiguous() and out.numel() >= bs
    ld = logits.stride(0) if bs > 1 else max(v, logits.stride(0))
    if workspace is None:
        workspace = nv.stream_workspace("argmax", logits.device, nv.lib().mv_argmax_workspace_bytes())
    nv.check("mv_argmax", nv.lib().mv_argmax(logits.data_ptr(), ld, bs, v, out.data_ptr(),
                                             workspace.data_ptr(), workspace.numel() * 4,
                                             nv.stream_ptr()))
    return out[:bs]


def sample(logits: torch.Tensor, temperatures: torch.Tensor, top_ps: torch.Tensor, top_k: int,
           uniforms: torch.Tensor = None, out: torch.Tensor = None, generator=None,
           greedy_ids: torch.Tensor = None) -> torch.Tensor:
    """logits [bs,V] bf16 -> int64 [bs]; `uniforms` [bs] in [0,1) (drawn here when not given).
    `greedy_ids` [bs] int64: rows with temperature <= 0 take these without reading the logits.
    No host synchronisation."""
    assert logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2
    assert logits.stride(1) == 1
    bs, v = logits.shape
    dev = logits.device
    temperatures = temperatures.to(device=dev, dtype=torch.float32).contiguous()
    top_ps = top_ps.to(device=dev, dtype=torch.float32).contiguous()
    if uniforms is None:
        uniforms = torch.rand(bs, device=dev, dtype=torch.float32, generator=generator)
    uniforms = uniforms.to(device=dev, dtype=torch.float32).contiguous()
    assert temperatures.numel() == bs and top_ps.numel() == bs and uniforms.numel() == bs
    if out is None:
        out = torch.empty(bs, dtype=torch.long, device=dev)
    assert out.dtype == torch.long and out.is_contiguous() and out.numel() >= bs
    ld = logits.stride(0) if bs > 1 else max(v, logits.stride(0))
    if greedy_ids is not None:
        assert greedy_ids.dtype == torch.long and greedy_ids.is_contiguous() and greedy_ids.numel() >= bs
    nv.check("mv_sample_or_greedy", nv.lib().mv_sample_or_greedy(
        logits.data_ptr(), ld, bs, v, temperatures.data_ptr(), top_ps.data_ptr(), int(top_k),
        uniforms.data_ptr(), nv.ptr(greedy_ids), out.data_ptr(), nv.stream_ptr()))
    return out[:bs]


class Sampler:
    def __call__(self, logits, temperatures, top_ps, top_k):
        if logits.is_cuda and logits.dtype == torch.bfloat16 and logits.shape[-1] % 8 == 0:
            if logits.stride(-1) != 1 or (logits.dim() == 2 and logits.stride(0) % 8):
                logits = logits.contiguous()
            return sample(logits, temperatures, top_ps, top_k)
        return self._sample_impl(logits, temperatures, top_ps, top_k)

    @staticmethod
    def _sample_impl(logits, temp, top_p, top_k):
        """temperature scale -> top-k -> softmax -> sort -> keep while cumsum < top_p (at least
        one) -> renormalise -> multinomial -> map back to vocabulary ids."""
        logits = logits.float() / temp[:, None]
        vals, idxs = torch.topk(logits, top_k, dim=-1)
        probs = torch.softmax(vals, dim=-1)
        sorted_p, sorted_i = torch.sort(probs, descending=True, dim=-1)
        keep = torch.cumsum(sorted_p, dim=-1) < top_p[:, None]
        keep[..., 0] = True
        masked = sorted_p * keep
        masked = masked / masked.sum(dim=-1, keepdim=True)
        pick = torch.multinomial(masked, 1)
        return idxs.gather(-1, sorted_i.gather(-1, pick)).squeeze(-1)
