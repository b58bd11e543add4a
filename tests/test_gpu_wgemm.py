"""Weight-streaming skinny GEMM (mv_wgemm) against the CPU oracle's linear_ref / swiglu_ref."""
import pytest
import torch

from conftest import assert_close_bf16, ulp_diff
from oracle.ref_ops import linear_ref, swiglu_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _mk(m, n, k, seed, scale=0.05):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) * scale).to(torch.bfloat16)
    return x, w, g


def test_tile_layout_is_what_the_header_says(mv):
    """tile (nt,kt), lane l, element j  <-  W[16nt + (l&15)][32kt + 8(l>>4) + j]."""
    from micro_vllm_amd.kernel.linear import StreamedWeight

    n, k = 48, 96
    w = torch.arange(n * k, dtype=torch.float32).reshape(n, k)
    w = (w % 251).to(torch.bfloat16)  # exactly representable, asymmetric
    tiles = StreamedWeight.from_linear(w.to(DEV)).tiles.cpu().reshape(n // 16, k // 32, 64, 8)
    want = w.reshape(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(n // 16, k // 32, 64, 8)
    assert torch.equal(tiles, want)


@pytest.mark.parametrize("m", [1, 2, 3, 4, 5, 8, 16, 17, 32, 33, 64])
@pytest.mark.parametrize("n,k", [(64, 128), (272, 2048), (4096, 4096)])
def test_wgemm_vs_oracle(mv, m, n, k):
    from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm

    x, w, _ = _mk(m, n, k, seed=m * 7 + n + k)
    sw = StreamedWeight.from_linear(w.to(DEV))
    # un-rounded accumulators: fp32 sums of exact bf16 products, only the order differs
    acc = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.float32, device=DEV))
    want32 = x.float() @ w.float().t()
    assert (acc.cpu() - want32).abs().max().item() <= 1e-4 * max(1.0, want32.abs().max().item())
    y = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV))
    want = linear_ref(x, w)
    # fp32 order noise (1e-4 of the output scale, felt only by sums that cancel to ~0) plus one
    # bf16 unit in the last place for values that sit on a rounding boundary
    assert_close_bf16(y, want, atol=1e-4 * max(1.0, float(want32.abs().max())), rtol=2 ** -7,
                      what=f"wgemm m={m} n={n} k={k}")
    assert float((ulp_diff(y, want) == 0).float().mean()) > 0.99


@pytest.mark.parametrize("m", [1, 4, 6, 32])
def test_wgemm_bias_and_residual_rounding_points(mv, m):
    """bf16(bf16(bf16(acc) + bias) + residual): the reference's matmul(out=), .add_(bias), .add_(res)."""
    from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm

    n, k = 512, 1024
    x, w, g = _mk(m, n, k, seed=40 + m)
    bias = torch.randn(n, generator=g).to(torch.bfloat16)
    res = torch.randn(m, n, generator=g).to(torch.bfloat16)
    sw = StreamedWeight.from_linear(w.to(DEV))
    y = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV),
              bias=bias.to(DEV), residual=res.to(DEV))
    base = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV)).cpu()
    # given the kernel's own bf16(acc), the two adds must be bit exact
    want = ((base.float() + bias.float()).to(torch.bfloat16).float() + res.float()).to(torch.bfloat16)
    assert torch.equal(y.cpu(), want)
    assert_close_bf16(y, (linear_ref(x, w, bias).float() + res.float()).to(torch.bfloat16))


@pytest.mark.parametrize("m", [1, 3, 8, 32])
def test_wgemm_swiglu_epilogue_equals_separate_ops(mv, m):
    """Fused gate/up GEMM + SwiGLU == matmul -> swiglu_gemm (model_graph.py:183-184), bit for bit."""
    from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm
    from micro_vllm_amd.kernel.swiglu import swiglu_gemm

    inter, k = 704, 512
    g = torch.Generator().manual_seed(60 + m)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16)
    up = (torch.randn(inter, k, generator=g) * 0.05).to(torch.bfloat16)
    gate = (torch.randn(inter, k, generator=g) * 0.05).to(torch.bfloat16)
    fused_w = StreamedWeight.from_up_gate(up.to(DEV), gate.to(DEV))
    plain_w = StreamedWeight.from_linear(torch.cat([up, gate], 0).to(DEV))
    fused = wgemm(x.to(DEV), fused_w, torch.empty(m, inter, dtype=torch.bfloat16, device=DEV), swiglu=True)
    gu = wgemm(x.to(DEV), plain_w, torch.empty(m, 2 * inter, dtype=torch.bfloat16, device=DEV))
    sep = swiglu_gemm(gu, torch.empty(m, inter, dtype=torch.bfloat16, device=DEV))
    assert torch.equal(fused, sep)
    want = swiglu_ref(linear_ref(x, torch.cat([up, gate], 0)))
    assert_close_bf16(fused, want, what="fused swiglu")


def test_wgemm_ksplit_variants_agree_and_are_deterministic(mv):
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm

    m, n, k = 2, 6144, 4096
    x, w, _ = _mk(m, n, k, seed=77)
    sw = StreamedWeight.from_linear(w.to(DEV))
    want = linear_ref(x, w)
    outs = []
    try:
        for ks in (1, 2, 4, 8):
            assert nv.lib().mv_wgemm_tune(1, ks) == 0
            y1 = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV))
            y2 = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV))
            assert torch.equal(y1, y2), f"ksplit {ks}: run-to-run difference"
            assert_close_bf16(y1, want, atol=1e-3, rtol=2 ** -7)
            outs.append(y1)
        for bpc in (1, 2, 4):
            assert nv.lib().mv_wgemm_tune(0, bpc) == 0
            y = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV))
            assert_close_bf16(y, want, atol=1e-3, rtol=2 ** -7)
        for waves in (8, 16):
            assert nv.lib().mv_wgemm_tune(2, waves) == 0
            y = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV))
            assert_close_bf16(y, want, atol=1e-3, rtol=2 ** -7)
    finally:
        nv.lib().mv_wgemm_tune(1, 0)
        nv.lib().mv_wgemm_tune(0, 0)
        nv.lib().mv_wgemm_tune(2, 0)


def test_wgemm_strided_rows_and_big_k(mv):
    """x and y rows taken from wider static buffers; K = 14336 (the down projection)."""
    from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm

    m, n, k = 3, 256, 14336
    x, w, _ = _mk(m, n, k, seed=91, scale=0.02)
    xb = torch.zeros(8, k + 64, dtype=torch.bfloat16, device=DEV)
    xb[:m, :k] = x.to(DEV)
    yb = torch.full((8, n + 32), 7.0, dtype=torch.bfloat16, device=DEV)
    sw = StreamedWeight.from_linear(w.to(DEV))
    wgemm(xb[:m, :k], sw, yb[:m, :n])
    assert_close_bf16(yb[:m, :n], linear_ref(x, w))
    assert (yb[m:] == 7).all() and (yb[:, n:] == 7).all()


def test_wgemm_rejects_bad_shapes(mv):
    from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm

    x, w, _ = _mk(2, 64, 128, seed=1)
    sw = StreamedWeight.from_linear(w.to(DEV))
    with pytest.raises(AssertionError):
        wgemm(x.to(DEV)[:, :96], sw, torch.empty(2, 64, dtype=torch.bfloat16, device=DEV))
    with pytest.raises(AssertionError):
        wgemm(x.to(DEV), sw, torch.empty(2, 48, dtype=torch.bfloat16, device=DEV))
    with pytest.raises(AssertionError):
        StreamedWeight.from_linear(torch.zeros(60, 128, dtype=torch.bfloat16, device=DEV))
    big = torch.zeros(65, 128, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(AssertionError):
        wgemm(big, sw, torch.empty(65, 64, dtype=torch.bfloat16, device=DEV))


@pytest.mark.parametrize("m", [1, 2, 3, 4])
@pytest.mark.parametrize("n,k,swiglu", [(512, 1024, False), (6144, 4096, False), (1408, 512, True),
                                        (256, 14336, False)])
def test_wgemm_norm_prologue_is_bit_identical_to_two_kernels(mv, m, n, k, swiglu):
    """rmsnorm_ -> wgemm and rmsnorm_residual_gemm -> wgemm against their one-launch forms."""
    from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, wgemm_norm
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm_
    from micro_vllm_amd.kernel.rmsnorm_residual import rmsnorm_residual_gemm

    g = torch.Generator().manual_seed(m * 31 + n + k)
    x = (torch.randn(m, k, generator=g) * 2).to(torch.bfloat16).to(DEV)
    r = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    lw = (1 + 0.2 * torch.randn(k, generator=g)).to(torch.bfloat16).to(DEV)
    w = (torch.randn(n, k, generator=g) * 0.03).to(torch.bfloat16)
    if swiglu:
        sw = StreamedWeight.from_up_gate(w[:n // 2].to(DEV), w[n // 2:].to(DEV))
    else:
        sw = StreamedWeight.from_linear(w.to(DEV))
    ncols = sw.out_features
    bias = None if swiglu else torch.randn(n, generator=g).to(torch.bfloat16).to(DEV)
    eps = 1e-5

    def out():
        return torch.empty(m, ncols, dtype=torch.bfloat16, device=DEV)

    # plain norm
    normed = rmsnorm_(x, lw, torch.empty_like(x), eps)
    two = wgemm(normed, sw, out(), bias=bias, swiglu=swiglu)
    one = wgemm_norm(x, lw, eps, sw, out(), bias=bias, swiglu=swiglu)
    assert torch.equal(one, two)
    # add + norm (variance from the un-rounded sum) with the sum written out
    n2, s2 = torch.empty_like(x), torch.empty_like(x)
    rmsnorm_residual_gemm(x, r, lw, n2, s2, eps)
    two = wgemm(n2, sw, out(), bias=bias, swiglu=swiglu)
    s1 = torch.empty_like(x)
    one = wgemm_norm(x, lw, eps, sw, out(), add_r=r, sum_out=s1, bias=bias, swiglu=swiglu)
    assert torch.equal(s1, s2) and torch.equal(one, two)
    # add (rounded) then norm of the rounded sum: h = a.add_(r); rmsnorm_(h)
    from micro_vllm_amd.kernel.linear import add_
    h = add_(x.clone(), r)
    two = wgemm(rmsnorm_(h, lw, torch.empty_like(x), eps), sw, out(), bias=bias, swiglu=swiglu)
    s3 = torch.empty_like(x)
    one = wgemm_norm(x, lw, eps, sw, out(), add_r=r, sum_out=s3, var_from_rounded=True, bias=bias,
                     swiglu=swiglu)
    assert torch.equal(s3, h) and torch.equal(one, two)


@pytest.mark.parametrize("m", [1, 4, 5, 16, 32, 33, 64])
@pytest.mark.parametrize("n,k", [(4096, 4096), (4096, 14336), (512, 1024), (8192, 512)])
def test_wgemm_postnorm_is_bit_identical_to_separate_kernels(mv, m, n, k):
    """Projection followed by add+RMSNorm (variance from the unrounded sum) or by residual add and
    the next block's RMSNorm (variance from the stored sum) against the form where the norm rides
    on the slab reduction: same bits in the projection output and in the normalised row."""
    from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, wgemm_postnorm
    from micro_vllm_amd.kernel.rmsnorm_add import rmsnorm_
    from micro_vllm_amd.kernel.rmsnorm_residual import rmsnorm_residual_gemm

    g = torch.Generator().manual_seed(m * 131 + n + k)
    x = torch.randn(m, k, generator=g).to(torch.bfloat16).to(DEV)
    r = torch.randn(m, n, generator=g).to(torch.bfloat16).to(DEV)
    lw = (1 + 0.2 * torch.randn(n, generator=g)).to(torch.bfloat16).to(DEV)
    bias = torch.randn(n, generator=g).to(torch.bfloat16).to(DEV)
    sw = StreamedWeight.from_linear((torch.randn(n, k, generator=g) * 0.03).to(torch.bfloat16).to(DEV))
    eps = 1e-5
    new = lambda: torch.empty(m, n, dtype=torch.bfloat16, device=DEV)

    # matmul -> rmsnorm_residual_gemm(y, r): sum = bf16(y + r), variance from the unrounded y + r
    for b in (None, bias):
        y = wgemm(x, sw, new(), bias=b, plan_slabs=True)    # k cut as the slab forms cut it
        n2, s2 = new(), new()
        rmsnorm_residual_gemm(y, r, lw, n2, s2, eps)
        s1, n1 = new(), new()
        wgemm_postnorm(x, sw, s1, lw, eps, n1, bias=b, residual=r)
        assert torch.equal(s1, s2) and torch.equal(n1, n2), f"add+norm m={m} bias={b is not None}"
    # matmul with residual epilogue -> rmsnorm_ of the stored hidden state
    h2 = wgemm(x, sw, new(), residual=r, plan_slabs=True)
    n2 = rmsnorm_(h2, lw, new(), eps)
    h1, n1 = new(), new()
    wgemm_postnorm(x, sw, h1, lw, eps, n1, residual=r, var_from_rounded=True)
    assert torch.equal(h1, h2) and torch.equal(n1, n2), f"residual then norm m={m}"
    # no residual at all: norm of the projection
    y2 = wgemm(x, sw, new(), plan_slabs=True)
    n2 = rmsnorm_(y2, lw, new(), eps)
    y1, n1 = new(), new()
    wgemm_postnorm(x, sw, y1, lw, eps, n1)
    assert torch.equal(y1, y2) and torch.equal(n1, n2), f"norm of projection m={m}"
