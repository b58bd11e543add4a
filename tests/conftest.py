import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def bf16_from_u16(arr):
    """uint16 numpy array (bf16 bit patterns) -> torch.bfloat16 tensor."""
    return torch.from_numpy(np.ascontiguousarray(arr).view(np.int16)).view(torch.bfloat16)


def u16_of(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def ulp_diff(a, b):
    """Distance in bf16 units-in-the-last-place between two bf16 tensors (monotone integer map)."""
    def key(t):
        i = t.detach().cpu().contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a) - key(b)).abs()


def assert_close_bf16(actual, expected, atol=1e-3, rtol=2 ** -7, what=""):
    """|a - e| <= atol + rtol * |e|.  atol is the path's stated 1e-3; rtol = one bf16 unit in the
    last place (2^-7 relative), needed because both sides are ROUNDED to bf16, whose spacing
    exceeds 1e-3 for every |value| >= 0.25."""
    a, e = actual.detach().float().cpu(), expected.detach().float().cpu()
    assert a.shape == e.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(e.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    err = (a - e).abs()
    tol = atol + rtol * e.abs()
    bad = err > tol
    if bad.any():
        idx = torch.nonzero(bad)[0].tolist()
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.numel()} elements out of tolerance; first at {idx}: "
            f"got {a[tuple(idx)].item()} want {e[tuple(idx)].item()} (max err {err.max().item():.3e})")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def mv():
    """The product package (imported under its importable alias)."""
    import micro_vllm_amd  # noqa: F401
    return sys.modules["micro_vllm_amd"]
