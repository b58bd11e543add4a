"""HIP row-wise ops through the C ABI against (a) the outputs of the reference's own Triton kernels
(tests/golden) and (b) the CPU oracle on fresh seeded inputs."""
import os

import numpy as np
import pytest
import torch

from conftest import bf16_from_u16, u16_of, ulp_diff
from oracle.ref_ops import rmsnorm_ref, rmsnorm_residual_ref, swiglu_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _exact_frac(a, b):
    return float((ulp_diff(a, b) == 0).float().mean())


def test_rmsnorm_vs_reference_triton_outputs(mv, golden_dir):
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm, rmsnorm_

    z = np.load(os.path.join(golden_dir, "norms.npz"))
    for c in range(int(z["n_cases"])):
        x, w = bf16_from_u16(z[f"c{c}_x"]).to(DEV), bf16_from_u16(z[f"c{c}_w"]).to(DEV)
        eps = float(z[f"c{c}_eps"])
        buf = torch.empty_like(x)
        assert rmsnorm_(x, w, buf, eps) is buf
        y2 = rmsnorm(x, w, eps)
        for got, name in ((buf, "rmsnorm_decode"), (y2, "rmsnorm_prefill")):
            want = bf16_from_u16(z[f"c{c}_{name}"])
            assert ulp_diff(got, want).max() <= 1, (c, name)   # fp32 sum order only
            assert _exact_frac(got, want) > 0.995, (c, name)


def test_add_rmsnorm_vs_reference_triton_outputs(mv, golden_dir):
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm_residual_fused
    from micro_vllm_amd.kernel.rmsnorm_residual import rmsnorm_residual_gemm

    z = np.load(os.path.join(golden_dir, "norms.npz"))
    for c in range(int(z["n_cases"])):
        x, r, w = (bf16_from_u16(z[f"c{c}_{k}"]).to(DEV) for k in ("x", "r", "w"))
        eps = float(z[f"c{c}_eps"])
        yn, res = torch.empty_like(x), torch.empty_like(x)
        a, b = rmsnorm_residual_gemm(x, r, w, yn, res, eps)
        assert a is yn and b is res
        assert np.array_equal(u16_of(res), z[f"c{c}_addnorm_res"]), c      # bit exact
        want = bf16_from_u16(z[f"c{c}_addnorm_y"])
        assert ulp_diff(yn, want).max() <= 1, c
        assert _exact_frac(yn, want) > 0.995, c
        y2, r2 = rmsnorm_residual_fused(x, r, w, eps)
        assert torch.equal(y2, yn) and torch.equal(r2, res)


def test_swiglu_vs_reference_triton_outputs(mv, golden_dir):
    from micro_vllm_amd.kernel.swiglu import swiglu_fused, swiglu_gemm

    z = np.load(os.path.join(golden_dir, "swiglu.npz"))
    for c in range(int(z["n_cases"])):
        gu = bf16_from_u16(z[f"c{c}_gate_up"]).to(DEV)
        want = bf16_from_u16(z[f"c{c}_out"])
        out = torch.empty(gu.shape[0], gu.shape[1] // 2, dtype=torch.bfloat16, device=DEV)
        assert swiglu_gemm(gu, out) is out
        assert ulp_diff(out, want).max() <= 1, c
        assert _exact_frac(out, want) > 0.99, c
        assert torch.equal(swiglu_fused(gu), out)


@pytest.mark.parametrize("rows,n", [(1, 4096), (32, 4096), (7, 8192), (5, 2048), (40, 128)])
def test_norms_vs_oracle_seeded(mv, rows, n):
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm_
    from micro_vllm_amd.kernel.rmsnorm_residual import rmsnorm_residual_gemm

    g = torch.Generator().manual_seed(rows * 1000 + n)
    x = (torch.randn(rows, n, generator=g) * 3).to(torch.bfloat16)
    r = torch.randn(rows, n, generator=g).to(torch.bfloat16)
    w = (1 + 0.2 * torch.randn(n, generator=g)).to(torch.bfloat16)
    y = rmsnorm_(x.to(DEV), w.to(DEV), torch.empty(rows, n, dtype=torch.bfloat16, device=DEV), 1e-5)
    want = rmsnorm_ref(x, w, 1e-5)
    assert ulp_diff(y, want).max() <= 1 and _exact_frac(y, want) > 0.995
    yn, res = torch.empty_like(x, device=DEV), torch.empty_like(x, device=DEV)
    rmsnorm_residual_gemm(x.to(DEV), r.to(DEV), w.to(DEV), yn, res, 1e-5)
    wy, wres = rmsnorm_residual_ref(x, r, w, 1e-5)
    assert torch.equal(res.cpu(), wres)
    assert ulp_diff(yn, wy).max() <= 1 and _exact_frac(yn, wy) > 0.995


def test_norm_on_strided_rows_and_empty(mv):
    """Rows taken out of a wider buffer (the decode runner passes `_buf[:bs]` views)."""
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm_

    g = torch.Generator().manual_seed(5)
    big = torch.randn(6, 1024, generator=g).to(torch.bfloat16)
    w = torch.ones(512, dtype=torch.bfloat16)
    x = big.to(DEV)[:, :512]
    out = torch.zeros(6, 1024, dtype=torch.bfloat16, device=DEV)
    rmsnorm_(x, w.to(DEV), out[:, 512:], 1e-6)
    assert ulp_diff(out[:, 512:], rmsnorm_ref(big[:, :512], w, 1e-6)).max() <= 1
    assert torch.count_nonzero(out[:, :512]) == 0
    empty = torch.empty(0, 512, dtype=torch.bfloat16, device=DEV)
    assert rmsnorm_(empty, w.to(DEV), torch.empty_like(empty)).shape == (0, 512)


def test_swiglu_large_and_odd(mv):
    from micro_vllm_amd.kernel.swiglu import swiglu_fused

    g = torch.Generator().manual_seed(9)
    for rows, d in [(32, 14336), (3, 1001), (1, 8)]:
        gu = (torch.randn(rows, 2 * d, generator=g) * 4).to(torch.bfloat16)
        got = swiglu_fused(gu.to(DEV))
        want = swiglu_ref(gu)
        assert ulp_diff(got, want).max() <= 1, (rows, d)
        assert _exact_frac(got, want) > 0.99, (rows, d)


def test_embedding_add_and_errors(mv):
    from micro_vllm_amd.kernel.linear import add_, embedding

    g = torch.Generator().manual_seed(3)
    table = torch.randn(1000, 256, generator=g).to(torch.bfloat16)
    ids = torch.tensor([0, 999, 5, 5, 17], dtype=torch.long)
    out = embedding(table.to(DEV), ids.to(DEV), torch.empty(5, 256, dtype=torch.bfloat16, device=DEV))
    assert torch.equal(out.cpu(), table[ids])
    a = torch.randn(4, 256, generator=g).to(torch.bfloat16)
    b = torch.randn(4, 256, generator=g).to(torch.bfloat16)
    bias = torch.randn(256, generator=g).to(torch.bfloat16)
    assert torch.equal(add_(a.to(DEV), b.to(DEV)).cpu(), (a.float() + b.float()).to(torch.bfloat16))
    assert torch.equal(add_(a.to(DEV), bias.to(DEV)).cpu(), (a.float() + bias.float()).to(torch.bfloat16))
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm_
    with pytest.raises(AssertionError):   # CPU tensors are rejected like the reference does
        rmsnorm_(a, bias, torch.empty_like(a))
    with pytest.raises(AssertionError):
        rmsnorm_(a.to(DEV), bias.to(DEV)[:100], torch.empty_like(a, device=DEV))


def test_store_kv_matches_reference_semantics(mv):
    """slot -> (slot // page, slot % page); -1 skipped (core/cache_manager.py:113-115,171-178)."""
    from micro_vllm_amd.core.cache_manager import KVCacheManager

    mgr = KVCacheManager(6, 16, 2, 4, 64, dtype=torch.bfloat16, device=DEV, max_tokens=64,
                         max_batch_size=2)
    g = torch.Generator().manual_seed(11)
    T = 21
    key = torch.randn(T, 4, 64, generator=g).to(torch.bfloat16)
    val = torch.randn(T, 4, 64, generator=g).to(torch.bfloat16)
    ok, slots = mgr.alloc(1, T)
    slots = slots.clone()
    slots[3] = -1
    mgr.put(1, key.to(DEV), val.to(DEV), layer=1, slot_map=slots)
    kc = torch.zeros(6, 16, 4, 64, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    for i, s in enumerate(slots.tolist()):
        if s != -1:
            kc[s // 16, s % 16] = key[i]
            vc[s // 16, s % 16] = val[i]
    assert torch.equal(mgr.k_caches[1].cpu(), kc) and torch.equal(mgr.v_caches[1].cpu(), vc)
    assert torch.count_nonzero(mgr.k_caches[0]) == 0
