"""Token selection.  `greedy` is the device-side argmax (mv_argmax, csrc/sampler.hip): the
temperature -> 0 limit of the reference's Sampler (core/layer/sampler.py:4-43), which is the
deterministic - hence parity-checkable - case.  `Sampler` keeps the reference's call signature;
its stochastic branch follows the same top-k / top-p / multinomial recipe with torch ops."""
import torch

from ... import _native as nv


def greedy(logits: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """logits [bs,V] bf16 -> int64 [bs] (first index on ties), no host synchronisation."""
    assert logits.is_cuda and logits.dtype == torch.bfloat16 and logits.dim() == 2
    assert logits.stride(1) == 1
    bs, v = logits.shape
    if out is None:
        out = torch.empty(bs, dtype=torch.long, device=logits.device)
    assert out.dtype == torch.long and out.is_contiguous() and out.numel() >= bs
    ld = logits.stride(0) if bs > 1 else max(v, logits.stride(0))
    nv.check("mv_argmax", nv.lib().mv_argmax(logits.data_ptr(), ld, bs, v, out.data_ptr(),
                                             nv.stream_ptr()))
    return out[:bs]


class Sampler:
    def __call__(self, logits, temperatures, top_ps, top_k):
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
