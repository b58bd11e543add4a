"""The sampling oracle against the reference's recipe (torch restatement of
core/layer/sampler.py:21-43 kept in core/layer/sampler.py::Sampler._sample_impl): same categorical
distribution, compared statistically; plus structural properties of the deterministic draw."""
import torch

from oracle.ref_sampler import distribution_ref, sample_ref


def test_inverse_cdf_draw_has_the_recipes_distribution():
    import micro_vllm_amd  # noqa: F401
    from micro_vllm_amd.core.layer.sampler import Sampler

    g = torch.Generator().manual_seed(0)
    logits = (torch.randn(1, 64, generator=g) * 2).to(torch.bfloat16)
    temp, top_p, top_k, n = 0.8, 0.9, 20, 40000
    idx, p = distribution_ref(logits[0], temp, top_p, top_k)
    want = torch.zeros(64)
    want[idx] = p
    u = torch.rand(n, generator=g)
    ours = sample_ref(logits.expand(n, -1), torch.full((n,), temp), torch.full((n,), top_p), top_k, u)
    torch.manual_seed(1)
    theirs = Sampler._sample_impl(logits.float().expand(n, -1).contiguous(), torch.full((n,), temp),
                                  torch.full((n,), top_p), top_k)
    f_ours = torch.bincount(ours, minlength=64).float() / n
    f_theirs = torch.bincount(theirs, minlength=64).float() / n
    # 5-sigma band of a binomial proportion
    band = 5 * torch.sqrt(want * (1 - want) / n) + 1e-4
    assert bool(((f_ours - want).abs() <= band).all())
    assert bool(((f_theirs - want).abs() <= band).all())
    assert set(ours.tolist()) <= set(idx[p > 0].tolist())      # never outside the kept prefix


def test_draw_edges():
    logits = torch.tensor([[0.0, 3.0, 3.0, -1.0, 2.0, 0.5, 0.25, -4.0]]).to(torch.bfloat16)
    one = torch.ones(1)
    # u = 0 -> the most probable entry, ties by lowest index; temperature 0 -> argmax
    assert sample_ref(logits, one, one, 8, torch.zeros(1)).item() == 1
    assert sample_ref(logits, torch.zeros(1), one, 8, torch.tensor([0.99])).item() == 1
    # top_p tiny -> only the first survives whatever u is
    assert sample_ref(logits, one, torch.tensor([1e-6]), 8, torch.tensor([0.999])).item() == 1
    # top_k 2 with u ~ 1 -> the second of the tied pair; note that `cum < top_p` drops the entry
    # that completes the mass, so top_p = 1 keeps only the first of two equal candidates
    assert sample_ref(logits, one, 2 * one, 2, torch.tensor([0.999])).item() == 2
    assert sample_ref(logits, one, one, 2, torch.tensor([0.999])).item() == 1


def test_oracle_distribution_matches_counts_recorded_from_the_reference(golden_dir):
    """tests/golden/sampler.npz holds, for three rows of logits, how often each index came out of
    40 000 draws of the reference's own Sampler._sample_impl (core/layer/sampler.py:21-43, run by
    tests/golden/generate_golden.py::gen_sampler).  The oracle's categorical distribution must
    explain those counts within a 5-sigma binomial band.  Indices with equal logits are compared
    as a group: which of several equal candidates falls off the top-k / top-p boundary is an
    implementation detail of torch.topk / torch.sort (the oracle and the device kernel take them in
    index order), the probability mass of the group is not."""
    import os

    import numpy as np

    d = np.load(os.path.join(golden_dir, "sampler.npz"))
    n = int(d["draws"])
    for ci in range(int(d["n_cases"])):
        logits = torch.from_numpy(d[f"c{ci}_logits"].view(np.int16)).view(torch.bfloat16)
        temp, top_p, top_k = float(d[f"c{ci}_temperature"]), float(d[f"c{ci}_top_p"]), int(d[f"c{ci}_top_k"])
        counts = torch.from_numpy(d[f"c{ci}_counts"]).float()
        idx, p = distribution_ref(logits, temp, top_p, top_k)
        want = torch.zeros(logits.numel())
        want[idx] = p
        assert abs(float(want.sum()) - 1.0) < 1e-5
        for v in torch.unique(logits.float()):
            grp = logits.float() == v
            w, f = float(want[grp].sum()), float(counts[grp].sum()) / n
            band = 5 * (w * (1 - w) / n) ** 0.5 + 1e-4
            assert abs(f - w) <= band, (ci, float(v), f, w)
            # the same number of group members survive in both (e.g. 5 of 6 equal candidates)
            assert int((counts[grp] > 0).sum()) == int((want[grp] > 0).sum()) or w * n / max(1, int(grp.sum())) < 30, \
                (ci, float(v))
        # nothing is ever drawn from a logit value the oracle gives no mass to
        support_vals = set(logits.float()[torch.from_numpy(d[f"c{ci}_support"])].tolist())
        assert support_vals <= set(logits.float()[want > 0].tolist()), ci
