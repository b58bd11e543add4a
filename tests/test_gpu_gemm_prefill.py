"""Many-row MFMA projection (mv_gemm_prefill) against the CPU oracle's linear_ref / swiglu_ref and
against the single-token kernel on the same tile image."""
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


# ragged M (partial m-blocks), N that is not a multiple of the 256-column block, one k stage
@pytest.mark.parametrize("m", [1, 16, 127, 128, 129, 300, 1024])
@pytest.mark.parametrize("n,k", [(64, 64), (272, 2048), (4096, 1024), (2560, 2048)])
def test_gemm_prefill_vs_oracle(mv, m, n, k):
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    x, w, _ = _mk(m, n, k, seed=m + 3 * n + k)
    sw = StreamedWeight.from_linear(w.to(DEV))
    acc = gemm_prefill(x.to(DEV), sw, torch.empty(m, n, dtype=torch.float32, device=DEV))
    want32 = x.float() @ w.float().t()
    scale = max(1.0, want32.abs().max().item())
    assert (acc.cpu() - want32).abs().max().item() <= 1e-4 * scale
    y = gemm_prefill(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV))
    want = linear_ref(x, w)
    assert_close_bf16(y, want, atol=1e-4 * scale, rtol=2 ** -7, what=f"gemm_prefill m={m} n={n} k={k}")
    assert float((ulp_diff(y, want) == 0).float().mean()) > 0.99


def test_gemm_prefill_epilogues(mv):
    """bias and residual rounding points, strided rows, fused SwiGLU (rows interleaved by 8)."""
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    m, n, k = 200, 512, 1024
    x, w, g = _mk(m, n, k, seed=5)
    bias = torch.randn(n, generator=g).to(torch.bfloat16)
    res = torch.randn(m, n, generator=g).to(torch.bfloat16)
    sw = StreamedWeight.from_linear(w.to(DEV))
    xb = torch.zeros(m, k + 64, dtype=torch.bfloat16, device=DEV)
    xb[:, :k] = x.to(DEV)
    yb = torch.full((m, n + 32), 7.0, dtype=torch.bfloat16, device=DEV)
    gemm_prefill(xb[:, :k], sw, yb[:, :n], bias=bias.to(DEV), residual=res.to(DEV))
    t = linear_ref(x, w, bias)
    want = (t.float() + res.float()).to(torch.bfloat16)
    # a one-ulp flip of an intermediate survives the cancellation in t + res: allow one ulp
    # of the larger operand
    err = (yb[:, :n].cpu().float() - want.float()).abs()
    tol = 1e-3 + 2 ** -7 * torch.maximum(t.float().abs(), want.float().abs())
    assert bool((err <= tol).all()), f"bias+residual: max err {err.max().item():.3e}"
    assert float((ulp_diff(yb[:, :n], want) == 0).float().mean()) > 0.99
    assert torch.all(yb[:, n:] == 7.0), "wrote past the row"

    inter = 768
    up = (torch.randn(inter, k, generator=g) * 0.05).to(torch.bfloat16)
    gate = (torch.randn(inter, k, generator=g) * 0.05).to(torch.bfloat16)
    gu = StreamedWeight.from_up_gate(up.to(DEV), gate.to(DEV))
    act = gemm_prefill(x.to(DEV), gu, torch.empty(m, inter, dtype=torch.bfloat16, device=DEV), swiglu=True)
    want = swiglu_ref(linear_ref(x, torch.cat([up, gate], 0)))
    # gate and up may each sit on a rounding boundary (one bf16 ulp apart between two fp32
    # summation orders); the product then moves by up to two ulps
    assert_close_bf16(act, want, atol=1e-3, rtol=2 ** -6, what="fused swiglu")
    assert float((ulp_diff(act, want) == 0).float().mean()) > 0.98


def _cus():
    return torch.cuda.get_device_properties(0).multi_processor_count


# tile width the planner gives each case on a 256-CU device: form 1 = 256 columns, 2 = 192, 3 = 224
WIDE_FORM = {(1000, 8208): 2, (2300, 4096): 2, (1300, 6144): 2, (1025, 8192): 2, (2048, 4096): 2,
             (700, 22016): 2, (2048, 28672): 3, (2048, 8192): 1, (2000, 7120): 3}


@pytest.mark.parametrize("m,n,k,mode", [(1000, 8208, 256, "plain"), (2300, 4096, 512, "residual"),
                                        (1300, 6144, 512, "bias"), (2048, 28672, 256, "swiglu"),
                                        (2048, 8192, 256, "plain"),
                                        # 224-column tiles, the last one 11 strips wide, the last m-block ragged
                                        (2000, 7120, 256, "residual"),
                                        (1025, 8192, 320, "bias"), (2048, 4096, 1024, "f32"),
                                        (700, 22016, 256, "swiglu"), (4096, 2048, 4096, "plain"),
                                        # tiles for half the CUs and a long k: two k-slices of the wide form
                                        (2000, 4096, 8192, "residual"), (1500, 4096, 8192, "bias")])
def test_gemm_prefill_wide_form(mv, m, n, k, mode):
    """Shapes whose 256 x 256 tiles fill at least half the CUs take the wide form (both operands
    through LDS, eight waves on a tile): ragged last m- and n-blocks, every epilogue, k from the
    shortest the form accepts to the model's."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    form = nv.lib().mv_gemm_prefill_form(m, n, k)
    assert form in (1, 2, 3), "this shape is meant to take the wide form"
    # all three tile widths are among the cases (the narrower ones where they fill the CUs in fuller rounds)
    if _cus() == 256:
        assert form == WIDE_FORM.get((m, n), 1), f"tile width of {(m, n, k)}: form {form}"
    if k >= 8192:
        assert nv.lib().mv_gemm_prefill_workspace_bytes(m, n, k) == 4096 + 2 * m * n * 4, "meant to be cut in two k-slices"
    x, w, g = _mk(m, n, k, seed=m + n + k, scale=0.03)
    bias = torch.randn(n, generator=g).to(torch.bfloat16) if mode == "bias" else None
    res = torch.randn(m, n, generator=g).to(torch.bfloat16) if mode == "residual" else None
    swiglu = mode == "swiglu"
    sw = (StreamedWeight.from_up_gate(w[:n // 2].to(DEV), w[n // 2:].to(DEV)) if swiglu
          else StreamedWeight.from_linear(w.to(DEV)))
    ncols = sw.out_features
    dt = torch.float32 if mode == "f32" else torch.bfloat16
    xb = torch.zeros(m, k + 64, dtype=torch.bfloat16, device=DEV)       # strided rows in
    xb[:, :k] = x.to(DEV)
    yb = torch.full((m, ncols + 32), 7.0, dtype=dt, device=DEV)         # and out
    gemm_prefill(xb[:, :k], sw, yb[:, :ncols], bias=None if bias is None else bias.to(DEV),
                 residual=None if res is None else res.to(DEV), swiglu=swiglu)
    assert torch.all(yb[:, ncols:] == 7.0), "wrote past the row"
    got = yb[:, :ncols].cpu()
    want32 = x.float() @ w.float().t()
    scale = max(1.0, want32.abs().max().item())
    if mode == "f32":
        assert (got - want32).abs().max().item() <= 1e-4 * scale
        return
    if swiglu:
        want = swiglu_ref(linear_ref(x, w))
        assert_close_bf16(got, want, atol=1e-3, rtol=2 ** -6, what="wide form, fused swiglu")
    else:
        t0 = linear_ref(x, w).float()
        t1 = linear_ref(x, w, bias).float()
        want = t1 if res is None else (t1 + res.float()).to(torch.bfloat16).float()
        mag = torch.maximum(torch.maximum(t0.abs(), t1.abs()), want.abs())
        # as for the k-sliced product below: each rounding point may flip by one unit at its own
        # magnitude between two fp32 summation orders, so two units of the largest intermediate
        err = (got.float() - want).abs()
        assert bool((err <= 1e-3 + 2 ** -6 * mag).all()), f"{mode}: max err {err.max().item():.3e}"
        want = want.to(torch.bfloat16)
    assert float((ulp_diff(got, want) == 0).float().mean()) > 0.98


@pytest.mark.parametrize("m,n,k,mode,strips", [
    (3072, 6144, 1024, "bias", 12),        # 48 tiles per XCD on 32 CUs: the 16 left over in two parts each
    (2560, 6144, 3072, "residual", 12),    # 40 tiles per XCD: the 8 left over in three parts each
    (4096, 6144, 2048, "swiglu", 16),      # the fused activation behind a cut tile
    (2300, 7168, 1024, "f32", 14),         # a ragged last m-block, fp32 out
    (3000, 7168, 1536, "residual", 16),    # 28 n-blocks: the eighth XCD owns none
])
def test_gemm_prefill_wide_form_cuts_left_over_tiles(mv, m, n, k, mode, strips):
    """With more tiles than CUs and a remainder of at most half a round, the left-over tiles are cut
    along k between workgroups inside the launch (the last part adds the others' sums in its
    epilogue).  Checked against the oracle, against the same launch with the cut switched off, and
    twice in a row (the parts count themselves in through counters that must be back at zero)."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    if _cus() != 256:
        pytest.skip("the cases are laid out for 32 CUs per XCD")
    L = nv.lib()
    x, w, g = _mk(m, n, k, seed=m + n + k + strips, scale=0.03)
    bias = torch.randn(n, generator=g).to(torch.bfloat16) if mode == "bias" else None
    res = torch.randn(m, n, generator=g).to(torch.bfloat16) if mode == "residual" else None
    swiglu = mode == "swiglu"
    sw = (StreamedWeight.from_up_gate(w[:n // 2].to(DEV), w[n // 2:].to(DEV)) if swiglu
          else StreamedWeight.from_linear(w.to(DEV)))
    ncols = sw.out_features
    dt = torch.float32 if mode == "f32" else torch.bfloat16

    def run():
        yb = torch.full((m, ncols + 32), 7.0, dtype=dt, device=DEV)
        gemm_prefill(x.to(DEV), sw, yb[:, :ncols], bias=None if bias is None else bias.to(DEV),
                     residual=None if res is None else res.to(DEV), swiglu=swiglu)
        assert torch.all(yb[:, ncols:] == 7.0), "wrote past the row"
        return yb[:, :ncols].cpu()

    try:
        L.mv_debug_prefill_options(1, strips)
        assert L.mv_gemm_prefill_workspace_bytes(m, n, k) == 4096 + 256 * 256 * 1024, "meant to cut its left-over tiles"
        cut, again = run(), run()
        L.mv_debug_prefill_options(0, strips)
        assert L.mv_gemm_prefill_workspace_bytes(m, n, k) == 0
        whole = run()
    finally:
        L.mv_debug_prefill_options(-1, -1)
    assert torch.equal(cut, again), "the second launch differs from the first"
    want32 = x.float() @ w.float().t()
    scale = max(1.0, want32.abs().max().item())
    if mode == "f32":
        assert (cut - want32).abs().max().item() <= 1e-4 * scale
        assert (cut - whole).abs().max().item() <= 1e-4 * scale
        return
    if swiglu:
        want = swiglu_ref(linear_ref(x, w))
        assert_close_bf16(cut, want, atol=1e-3, rtol=2 ** -6, what="cut tiles, fused swiglu")
        assert_close_bf16(cut, whole, atol=1e-3, rtol=2 ** -6, what="cut vs whole tiles, fused swiglu")
    else:
        t0 = linear_ref(x, w).float()
        t1 = linear_ref(x, w, bias).float()
        want = t1 if res is None else (t1 + res.float()).to(torch.bfloat16).float()
        mag = torch.maximum(torch.maximum(t0.abs(), t1.abs()), want.abs())
        for name, ref in (("oracle", want), ("whole tiles", whole.float())):
            err = (cut.float() - ref).abs()
            assert bool((err <= 1e-3 + 2 ** -6 * mag).all()), f"{mode} vs {name}: max err {err.max().item():.3e}"
        want = want.to(torch.bfloat16)
    assert float((ulp_diff(cut, want) == 0).float().mean()) > 0.98
    assert float((ulp_diff(cut, whole) == 0).float().mean()) > 0.98


def test_gemm_prefill_cut_tiles_replayed_from_a_captured_graph(mv):
    """The cut of left-over tiles keeps its counters inside the library and returns them to zero in
    the launch itself: a captured launch replays any number of times with the bits of the eager one."""
    import ctypes
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    if _cus() != 256:
        pytest.skip("the case is laid out for 32 CUs per XCD")
    L = nv.lib()
    m, n, k = 3072, 6144, 1024
    x, w, _ = _mk(m, n, k, seed=11, scale=0.03)
    xd, sw = x.to(DEV), StreamedWeight.from_linear(w.to(DEV))
    try:
        L.mv_debug_prefill_options(1, 12)
        need = L.mv_gemm_prefill_workspace_bytes(m, n, k)
        assert need == 4096 + 256 * 256 * 1024, "meant to cut its left-over tiles"
        ws = torch.zeros(need // 4, dtype=torch.float32, device=DEV)   # starts with the parts' counters
        eager = gemm_prefill(xd, sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV), workspace=ws).clone()
        out = torch.zeros(m, n, dtype=torch.bfloat16, device=DEV)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            side.synchronize()
            nv.check("begin", L.mv_graph_begin_capture(side.cuda_stream))
            gemm_prefill(xd, sw, out, workspace=ws)
            h = ctypes.c_void_p()
            nv.check("end", L.mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
            for replay in range(3):
                out.zero_()
                nv.check("launch", L.mv_graph_launch(h, side.cuda_stream))
                side.synchronize()
                assert torch.equal(out, eager), f"replay {replay} differs from the eager launch"
            L.mv_graph_destroy(h)
    finally:
        L.mv_debug_prefill_options(-1, -1)


def test_gemm_prefill_cut_tiles_on_two_streams_at_once(mv):
    """Two streams issue launches that cut their left-over tiles, back to back and unsynchronised: each
    stream has its own workspace (counters and accumulator slots), so neither sees the other's parts."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    if _cus() != 256:
        pytest.skip("the case is laid out for 32 CUs per XCD")
    L = nv.lib()
    m, n, k = 3072, 6144, 1024
    cases = []
    for seed in (21, 22):
        x, w, _ = _mk(m, n, k, seed=seed, scale=0.03)
        cases.append((x.to(DEV), StreamedWeight.from_linear(w.to(DEV))))
    try:
        L.mv_debug_prefill_options(1, 12)
        serial = [gemm_prefill(xd, sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV)).clone() for xd, sw in cases]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        outs = [torch.zeros(m, n, dtype=torch.bfloat16, device=DEV) for _ in cases]
        for _ in range(4):
            for st, (xd, sw), out in zip(streams, cases, outs):
                with torch.cuda.stream(st):
                    gemm_prefill(xd, sw, out)
        torch.cuda.synchronize()
    finally:
        L.mv_debug_prefill_options(-1, -1)
    for i, (out, want) in enumerate(zip(outs, serial)):
        assert torch.equal(out, want), f"stream {i}: differs from the launch on its own"


_ONE_WORKGROUP_PER_TILE_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
import micro_vllm_amd
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
g = torch.Generator().manual_seed(77)
m, n, k = 2500, 9216, 1024
x = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16)
w = (torch.randn(n, k, generator=g) * 0.03).to(torch.bfloat16)
y = gemm_prefill(x.cuda(), StreamedWeight.from_linear(w.cuda()), torch.empty(m, n, dtype=torch.bfloat16, device="cuda"))
torch.save(y.cpu(), sys.argv[2])
"""


def test_gemm_prefill_wide_form_one_workgroup_per_tile_gives_the_same_bits(mv, tmp_path):
    """MVLLM_PREFILL_PERSIST=0 (read once per process: a child process) launches one workgroup per tile
    where the default has one per CU walk several tiles and cut the left-over ones along k: a tile
    that is not cut is summed in the same order either way, so the outputs must be the same bits
    wherever the default did not cut, and within the usual bound where it did."""
    import os
    import subprocess
    import sys
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "per_tile.pt")
    r = subprocess.run([sys.executable, "-c", _ONE_WORKGROUP_PER_TILE_CHILD, root, out], capture_output=True, text=True,
                       timeout=280, env=dict(os.environ, MVLLM_PREFILL_PERSIST="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    per_tile = torch.load(out, weights_only=True)
    g = torch.Generator().manual_seed(77)
    m, n, k = 2500, 9216, 1024
    x = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn(n, k, generator=g) * 0.03).to(torch.bfloat16)
    assert nv.lib().mv_gemm_prefill_form(m, n, k) in (1, 2, 3), "meant to take the wide form"
    walked = gemm_prefill(x.to(DEV), StreamedWeight.from_linear(w.to(DEV)),
                          torch.empty(m, n, dtype=torch.bfloat16, device=DEV)).cpu()
    same = float((ulp_diff(walked, per_tile) == 0).float().mean())
    assert same > 0.98, f"only {same:.4f} of the outputs agree bit for bit"
    err = (walked.float() - per_tile.float()).abs()
    assert bool((err <= 1e-3 + 2 ** -6 * per_tile.float().abs()).all())
    want = linear_ref(x, w)
    assert float((ulp_diff(walked, want) == 0).float().mean()) > 0.98


def test_gemm_prefill_agrees_with_single_token_kernel(mv):
    """Same weights, same rows: the two kernels differ only in fp32 summation order."""
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill, wgemm

    m, n, k = 48, 1024, 4096
    x, w, _ = _mk(m, n, k, seed=9, scale=0.02)
    sw = StreamedWeight.from_linear(w.to(DEV))
    a = gemm_prefill(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV))
    b = wgemm(x.to(DEV), sw, torch.empty(m, n, dtype=torch.bfloat16, device=DEV))
    assert_close_bf16(a, b.cpu(), atol=1e-4 * max(1.0, float(b.float().abs().max())), rtol=2 ** -7,
                      what="prefill vs single-token kernel")
    assert float((ulp_diff(a, b) == 0).float().mean()) > 0.99


def test_gemm_prefill_rejects_bad_shapes(mv):
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    w = StreamedWeight.from_linear(torch.zeros(64, 96, dtype=torch.bfloat16, device=DEV))
    x = torch.zeros(4, 96, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(AssertionError):  # K = 96 is not a multiple of 64
        gemm_prefill(x, w, torch.empty(4, 64, dtype=torch.bfloat16, device=DEV))
    assert gemm_prefill(x[:0], w, torch.empty(0, 64, dtype=torch.bfloat16, device=DEV)).shape == (0, 64)


@pytest.mark.parametrize("m", [1, 32, 100, 128, 300, 512])
@pytest.mark.parametrize("n,k,mode", [(512, 4096, "plain"), (4096, 14336, "residual"), (1536, 4096, "swiglu"),
                                      (6144, 4096, "bias"), (256, 8192, "f32")])
def test_gemm_prefill_k_slices(mv, m, n, k, mode):
    """Few output tiles and a long k: the product is cut into k-slices that run side by side and are
    added by a second launch (with the epilogue).  Against the oracle as the unsplit product is,
    and against the unsplit product itself (the C entry without a workspace): the two differ only
    in fp32 summation order."""
    import ctypes

    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill

    L = nv.lib()
    assert L.mv_gemm_prefill_workspace_bytes(m, n, k) > 0, "this shape is meant to be split"
    x, w, g = _mk(m, n, k, seed=m + n + k, scale=0.03)
    bias = torch.randn(n, generator=g).to(torch.bfloat16) if mode == "bias" else None
    res = torch.randn(m, n, generator=g).to(torch.bfloat16) if mode == "residual" else None
    swiglu = mode == "swiglu"
    if swiglu:
        sw = StreamedWeight.from_up_gate(w[:n // 2].to(DEV), w[n // 2:].to(DEV))
    else:
        sw = StreamedWeight.from_linear(w.to(DEV))
    ncols = sw.out_features
    dt = torch.float32 if mode == "f32" else torch.bfloat16
    xd = x.to(DEV)
    split = gemm_prefill(xd, sw, torch.empty(m, ncols, dtype=dt, device=DEV),
                         bias=None if bias is None else bias.to(DEV),
                         residual=None if res is None else res.to(DEV), swiglu=swiglu)
    whole = torch.empty(m, ncols, dtype=dt, device=DEV)
    flags = (nv.MV_GEMM_SWIGLU if swiglu else 0) | (nv.MV_GEMM_OUT_F32 if mode == "f32" else 0)
    rd = None if res is None else res.to(DEV)
    bd = None if bias is None else bias.to(DEV)
    nv.check("mv_gemm_prefill", L.mv_gemm_prefill(
        xd.data_ptr(), k, sw.tiles.data_ptr(), nv.ptr(bd), nv.ptr(rd), n if rd is not None else 0,
        whole.data_ptr(), ncols, m, n, k, flags, nv.stream_ptr()))
    want32 = x.float() @ w.float().t()
    scale = max(1.0, want32.abs().max().item())
    if mode == "f32":
        assert (split.cpu() - want32).abs().max().item() <= 1e-4 * scale
        assert (split - whole).abs().max().item() <= 1e-4 * scale
        return
    if swiglu:
        want = swiglu_ref(linear_ref(x, w))
        assert_close_bf16(split, want, atol=1e-3, rtol=2 ** -6, what="k-sliced swiglu")
        assert_close_bf16(split, whole.cpu(), atol=1e-3, rtol=2 ** -6, what="k-sliced vs whole swiglu")
    else:
        # Every rounding point may flip by one unit in the last place AT ITS OWN MAGNITUDE between
        # two fp32 summation orders, and the flip survives where a later addend cancels most of
        # the value: the bound is taken at the largest intermediate, not at the result.
        t0 = linear_ref(x, w).float()
        t1 = linear_ref(x, w, bias).float()
        want = t1 if res is None else (t1 + res.float()).to(torch.bfloat16).float()
        mag = torch.maximum(torch.maximum(t0.abs(), t1.abs()), want.abs())
        tol = 1e-3 + 2 ** -6 * mag
        for other, name in ((want, "oracle"), (whole.cpu().float(), "whole")):
            err = (split.cpu().float() - other).abs()
            assert bool((err <= tol).all()), f"{mode} vs {name}: max err {err.max().item():.3e}"
        want = want.to(torch.bfloat16)
    assert float((ulp_diff(split, want) == 0).float().mean()) > 0.98
    assert float((ulp_diff(split, whole) == 0).float().mean()) > 0.98
