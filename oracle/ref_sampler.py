"""CPU restatement of the reference's sampling recipe (test infrastructure).

Follows core/layer/sampler.py:21-43 (`Sampler._sample_impl`): logits / temperature -> top-k ->
softmax -> sort descending -> keep while the inclusive cumulative sum is < top_p (at least one)
-> renormalise -> draw.  The reference draws with torch.multinomial, whose result depends on
torch's RNG stream and algorithm; here the draw is the inverse CDF of an explicit uniform number,
which has the same distribution and makes the function deterministic (PARITY of individual draws
with the reference is therefore UNPINNED; the distribution is compared statistically in
tests/test_sampler_cpu.py).  Equal logits are ordered by ascending index.
"""
import torch


def sample_ref(logits: torch.Tensor, temperatures: torch.Tensor, top_ps: torch.Tensor, top_k: int,
               uniforms: torch.Tensor) -> torch.Tensor:
    """logits [bs,V] (bf16 or fp32), temperatures/top_ps/uniforms [bs] -> int64 [bs]."""
    bs, v = logits.shape
    k = min(top_k, v, 1024)
    out = torch.empty(bs, dtype=torch.long)
    for r in range(bs):
        x = logits[r].float()
        order = torch.sort(x, descending=True, stable=True).indices[:k]   # ties: ascending index
        vals = x[order]
        t = float(temperatures[r])
        if not t > 0:
            out[r] = order[0]
            continue
        s = (vals / t).to(torch.float32)
        e = torch.exp(s - s[0])
        cum = torch.cumsum(e, 0) / e.sum()
        valid = cum < float(top_ps[r])
        valid[0] = True
        m = int(valid.sum())
        target = float(uniforms[r]) * float(cum[m - 1])
        nb = int((cum[:m] <= target).sum())
        out[r] = order[min(nb, m - 1)]
    return out


def distribution_ref(logits_row: torch.Tensor, temperature: float, top_p: float, top_k: int):
    """The categorical distribution the recipe draws from: (indices, probabilities)."""
    x = logits_row.float()
    k = min(top_k, x.numel(), 1024)
    order = torch.sort(x, descending=True, stable=True).indices[:k]
    s = x[order] / temperature
    p = torch.softmax(s, 0)
    cum = torch.cumsum(p, 0)
    valid = cum < top_p
    valid[0] = True
    p = p * valid
    return order, p / p.sum()
