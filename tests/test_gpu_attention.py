"""Paged decode attention (mv_paged_attn_decode) against the CPU restatement of
flash_attn_with_kvcache.  PARITY UNPINNED at the reference level (third-party kernel, no golden
vector upstream): the oracle restates the documented contract.

Checked per case:
  * block-table indexing and the KV append (rotated k, raw v, every untouched slot): bit exact;
  * output vs the exact-softmax oracle: 1e-5 absolute + one bf16 unit in the last place (both
    sides round the output to bf16) - this is the tight check of the kernel's arithmetic, made
    whenever the dot-product form of the kernel runs (mv_paged_attn_decode_form == 0);
  * output vs the oracle that also rounds P to bf16 (what a tensor-core kernel does, and what the
    matrix-core form of the kernel does): the path's stated 1e-3 + one bf16 unit + that
    rounding's own bound 2^-9 * sum_t p_t |v_t|."""
import pytest
import torch

from conftest import assert_close_bf16
from oracle.ref_attention import flash_attn_with_kvcache_ref
from oracle.ref_ops import rotary_tables_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_case(B, Hq, Hkv, D, page, ctxs, seed, n_pages=None, rotary=True, pad_cols=2):
    """Random cache pre-filled for `ctxs` tokens per row through a shuffled block table."""
    g = torch.Generator().manual_seed(seed)
    need = [(c + 1 + page - 1) // page for c in ctxs]
    n_pages = n_pages or (sum(need) + 3)
    perm = torch.randperm(n_pages, generator=g).tolist()
    width = max(need) + pad_cols
    bt = torch.full((B, width), -1, dtype=torch.int32)
    it = iter(perm)
    for b in range(B):
        for i in range(need[b]):
            bt[b, i] = next(it)
    k_cache = torch.randn(n_pages, page, Hkv, D, generator=g).to(torch.bfloat16)
    v_cache = torch.randn(n_pages, page, Hkv, D, generator=g).to(torch.bfloat16)
    q = torch.randn(B, 1, Hq, D, generator=g).to(torch.bfloat16)
    k = torch.randn(B, 1, Hkv, D, generator=g).to(torch.bfloat16)
    v = torch.randn(B, 1, Hkv, D, generator=g).to(torch.bfloat16)
    seqlens = torch.tensor(ctxs, dtype=torch.int32)
    cos = sin = None
    if rotary:
        cos, sin = rotary_tables_ref(D, 32 * 256)
    return dict(q=q, k=k, v=v, k_cache=k_cache, v_cache=v_cache, bt=bt, seqlens=seqlens, cos=cos, sin=sin)


def dot_form(case):
    """True when the library runs its fp32-P kernel for this case's shape."""
    from micro_vllm_amd import _native as nv
    _, page, hkv, d = case["k_cache"].shape
    form = nv.lib().mv_paged_attn_decode_form(case["q"].shape[2], hkv, d, page)
    assert form in (0, 1)
    return form == 0


def check_output(got, case, what="", scale=None):
    """Both oracle forms (see the module docstring)."""
    got = got.reshape(case["q"].shape)
    if dot_form(case):
        exact = flash_attn_with_kvcache_ref(case["q"], case["k_cache"].clone(), case["v_cache"].clone(),
                                            case["k"], case["v"], case["cos"], case["sin"], case["seqlens"],
                                            case["bt"], softmax_scale=scale, p_bf16=False)
        assert_close_bf16(got, exact, atol=1e-5, rtol=2 ** -7, what=what + " (exact softmax)")
    faithful, bound = flash_attn_with_kvcache_ref(
        case["q"], case["k_cache"].clone(), case["v_cache"].clone(), case["k"], case["v"], case["cos"],
        case["sin"], case["seqlens"], case["bt"], softmax_scale=scale, p_bf16=True, return_p_bound=True)
    err = (got.float().cpu() - faithful.float()).abs()
    assert bool((err <= 1e-3 + 2 ** -7 * faithful.float().abs() + bound).all()), \
        f"{what} (bf16-P oracle): max err {err.max().item():.3e}"


def run_both(case, num_splits=0):
    from micro_vllm_amd.core.paged_attention import flash_attn_with_kvcache

    kc_ref, vc_ref = case["k_cache"].clone(), case["v_cache"].clone()
    want = flash_attn_with_kvcache_ref(case["q"], kc_ref, vc_ref, case["k"], case["v"], case["cos"],
                                       case["sin"], case["seqlens"], case["bt"], causal=True,
                                       p_bf16=False)
    kc, vc = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
    to = lambda t: None if t is None else t.to(DEV)
    got = flash_attn_with_kvcache(
        q=case["q"].to(DEV), k_cache=kc, v_cache=vc, k=case["k"].to(DEV), v=case["v"].to(DEV),
        rotary_cos=to(case["cos"]), rotary_sin=to(case["sin"]), cache_seqlens=case["seqlens"].to(DEV),
        block_table=case["bt"].to(DEV), causal=True, window_size=(-1, -1), rotary_interleaved=False,
        alibi_slopes=None, num_splits=num_splits)
    return got, want, (kc, vc), (kc_ref, vc_ref)


@pytest.mark.parametrize("D,Hq,Hkv", [(128, 8, 2), (128, 4, 4), (128, 8, 1), (128, 4, 2),
                                      (64, 8, 2), (64, 8, 1), (64, 4, 4), (64, 2, 1),
                                      # 7, 6, 5 and 3 query heads per kv head (Qwen2-style ratios)
                                      (128, 14, 2), (128, 12, 2), (128, 5, 1), (64, 6, 2), (64, 7, 1)])
@pytest.mark.parametrize("splits", [1, 3, 0])
def test_decode_attention_vs_oracle(mv, D, Hq, Hkv, splits):
    ctxs = [0, 1, 15, 16, 17, 255, 256, 257, 700]
    case = make_case(len(ctxs), Hq, Hkv, D, 256, ctxs, seed=D + Hq * 3 + Hkv)
    got, want, (kc, vc), (kc_ref, vc_ref) = run_both(case, splits)
    assert got.shape == (len(ctxs), 1, Hq, D)
    check_output(got, case, what=f"attn D={D} Hq={Hq} Hkv={Hkv} splits={splits}")
    # the appended row (rotated k, raw v) and every untouched slot: bit exact
    assert torch.equal(vc.cpu(), vc_ref)
    assert torch.equal(kc.cpu(), kc_ref)


def test_small_pages_many_blocks_and_long_context(mv):
    """page=16 exercises block_table gathering on nearly every row load; ctx 4095 / 4096."""
    for page, ctxs in [(16, [5, 100, 333, 47]), (256, [4095, 4096])]:
        case = make_case(len(ctxs), 8, 2, 128, page, ctxs, seed=page)
        for splits in (1, 8, 0):
            got, want, (kc, vc), (kc_ref, vc_ref) = run_both(case, splits)
            check_output(got, case, what=f"page={page} splits={splits}")
            assert torch.equal(kc.cpu(), kc_ref) and torch.equal(vc.cpu(), vc_ref)


def test_duplicate_rows_write_the_same_slot(mv):
    """The scheduler pads decode batches by repeating sequences (core/scheduler.py:88-92): rows
    with one block table and position write identical KV into one slot and agree on the output."""
    case = make_case(1, 8, 2, 128, 256, [300], seed=5)
    for key in ("q", "k", "v", "bt", "seqlens"):
        case[key] = case[key].repeat(4, *([1] * (case[key].dim() - 1)))
    got, want, (kc, vc), (kc_ref, vc_ref) = run_both(case, 0)
    check_output(got, case, what="duplicates")
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[3])
    assert torch.equal(kc.cpu(), kc_ref) and torch.equal(vc.cpu(), vc_ref)


def test_no_rotary_and_custom_scale(mv):
    from micro_vllm_amd.core.paged_attention import paged_attn_decode

    case = make_case(3, 4, 2, 64, 256, [10, 200, 513], seed=8, rotary=False)
    kc_ref, vc_ref = case["k_cache"].clone(), case["v_cache"].clone()
    want = flash_attn_with_kvcache_ref(case["q"], kc_ref, vc_ref, case["k"], case["v"], None, None,
                                       case["seqlens"], case["bt"], softmax_scale=0.2, p_bf16=False)
    kc, vc = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
    got = paged_attn_decode(case["q"][:, 0].to(DEV), case["k"][:, 0].to(DEV), case["v"][:, 0].to(DEV),
                            kc, vc, case["seqlens"].to(DEV), case["bt"].to(DEV), softmax_scale=0.2)
    assert_close_bf16(got, want[:, 0], atol=1e-5)
    check_output(got, case, what="custom scale", scale=0.2)
    assert torch.equal(kc.cpu(), kc_ref)


def test_unallocated_page_is_skipped_not_dereferenced(mv):
    """A -1 entry where the new token should land (the reference's own allocator produces this for
    prompts of exactly k pages) must not fault: the write is dropped and the key ignored."""
    case = make_case(2, 4, 2, 128, 256, [256, 100], seed=13)
    case["bt"][0, 1] = -1
    kc, vc = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
    from micro_vllm_amd.core.paged_attention import paged_attn_decode
    got = paged_attn_decode(case["q"][:, 0].to(DEV), case["k"][:, 0].to(DEV), case["v"][:, 0].to(DEV),
                            kc, vc, case["seqlens"].to(DEV), case["bt"].to(DEV),
                            case["cos"].to(DEV), case["sin"].to(DEV))
    torch.cuda.synchronize()
    assert torch.isfinite(got.float()).all()
    # row 1 is unaffected and still matches the oracle
    kc_ref, vc_ref = case["k_cache"].clone(), case["v_cache"].clone()
    want = flash_attn_with_kvcache_ref(case["q"][1:], kc_ref, vc_ref, case["k"][1:], case["v"][1:],
                                       case["cos"], case["sin"], case["seqlens"][1:], case["bt"][1:],
                                       p_bf16=False)
    assert_close_bf16(got[1], want[0, 0], atol=1e-5)


def test_strided_qkv_views(mv):
    """q/k/v as views of one fused [B, (Hq+2Hkv)D] projection row, as the runner passes them."""
    from micro_vllm_amd.core.paged_attention import paged_attn_decode

    B, Hq, Hkv, D = 5, 8, 2, 128
    case = make_case(B, Hq, Hkv, D, 256, [3, 64, 65, 1000, 511], seed=21)
    qkv = torch.cat([case["q"][:, 0].reshape(B, -1), case["k"][:, 0].reshape(B, -1),
                     case["v"][:, 0].reshape(B, -1)], dim=1).to(DEV)
    q = qkv[:, :Hq * D].unflatten(1, (Hq, D))
    k = qkv[:, Hq * D:(Hq + Hkv) * D].unflatten(1, (Hkv, D))
    v = qkv[:, (Hq + Hkv) * D:].unflatten(1, (Hkv, D))
    kc, vc = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
    got = paged_attn_decode(q, k, v, kc, vc, case["seqlens"].to(DEV), case["bt"].to(DEV),
                            case["cos"].to(DEV), case["sin"].to(DEV))
    kc_ref, vc_ref = case["k_cache"].clone(), case["v_cache"].clone()
    want = flash_attn_with_kvcache_ref(case["q"], kc_ref, vc_ref, case["k"], case["v"], case["cos"],
                                       case["sin"], case["seqlens"], case["bt"], p_bf16=False)
    assert_close_bf16(got, want[:, 0], atol=1e-5)
    check_output(got, case, what="strided views")
    assert torch.equal(kc.cpu(), kc_ref)


def test_softmax_spike_forces_rescale(mv):
    """One key far above the rest, late in the sequence: the running maximum must jump and every
    earlier accumulator be rescaled."""
    case = make_case(1, 4, 1, 128, 256, [900], seed=33, rotary=False)
    pg = int(case["bt"][0, 850 // 256])
    case["k_cache"][pg, 850 % 256, 0] = case["q"][0, 0, 1] * 3.0
    got, want, _, _ = run_both(case, 1)
    check_output(got, case, what="spike, 1 split")
    got, want, _, _ = run_both(case, 5)
    check_output(got, case, what="spike, 5 splits")


def test_in_kernel_merge_equals_second_kernel_merge(mv):
    """Split merge by the last-arriving workgroup vs the separate merge kernel: same bits, and the
    arrival counters return to zero (the second call would otherwise never merge)."""
    from micro_vllm_amd import _native as nv

    case = make_case(6, 8, 2, 128, 256, [0, 63, 64, 1000, 2049, 4096], seed=55)
    outs = {}
    try:
        for mode in (1, 0, 1):
            assert nv.lib().mv_paged_attn_tune(0, mode) == 0
            for splits in (2, 7, 64):
                got, _, _, _ = run_both(case, splits)
                outs.setdefault(splits, []).append(got.clone())
    finally:
        nv.lib().mv_paged_attn_tune(0, 1)
    for splits, lst in outs.items():
        assert torch.equal(lst[0], lst[1]) and torch.equal(lst[0], lst[2]), splits
        check_output(lst[0], case, what=f"merge splits={splits}")


@pytest.mark.parametrize("D,Hq,Hkv", [(128, 8, 2), (64, 8, 2), (128, 8, 1)])
def test_slots_that_do_not_count_may_hold_anything(mv, D, Hq, Hkv):
    """The kernel requests the first rows of every split before it knows the sequence length, so it
    reads slots at and past the length and pages no block table points to.  Whatever they hold
    (NaN, Inf: a fresh cache is uninitialised memory) must not reach the output: same bits as with a
    clean cache, for every split count."""
    ctxs = [0, 1, 63, 64, 65, 300, 1000]
    page = 256
    case = make_case(len(ctxs), Hq, Hkv, D, page, ctxs, seed=91, pad_cols=3)
    poison = torch.tensor([float("nan"), float("inf"), float("-inf")]).to(torch.bfloat16)
    dirty = {k: v.clone() if torch.is_tensor(v) else v for k, v in case.items()}
    used = torch.zeros(case["k_cache"].shape[0], page, dtype=torch.bool)
    for b, c in enumerate(ctxs):
        for t in range(c + 1):  # keys 0..c-1 are cached, slot c receives the new key
            used[int(case["bt"][b, t // page]), t % page] = True
    for name in ("k_cache", "v_cache"):
        t = dirty[name]
        fill = poison[torch.arange(t.numel()) % 3].reshape(t.shape)
        dirty[name] = torch.where(used[:, :, None, None], t, fill)
    for splits in (1, 3, 8, 0):
        clean, _, _, _ = run_both(case, splits)
        got, _, (kc, vc), _ = run_both(dirty, splits)
        assert bool(torch.isfinite(got.float()).all()), f"splits={splits}"
        assert torch.equal(got, clean), f"splits={splits}"
        # and nothing but the new key's slot was written
        keep = ~used[:, :, None, None].expand_as(dirty["k_cache"])
        assert torch.equal(kc.cpu().view(torch.int16)[keep], dirty["k_cache"].view(torch.int16)[keep])
        assert torch.equal(vc.cpu().view(torch.int16)[keep], dirty["v_cache"].view(torch.int16)[keep])


@pytest.fixture
def matrix_core_form(mv):
    """Force the matrix-core form of the kernel wherever it applies (D = 128, 4 or 8 query heads per
    kv head) for the duration of a test."""
    from micro_vllm_amd import _native as nv
    assert nv.lib().mv_paged_attn_tune(2, 2) == 0
    yield
    assert nv.lib().mv_paged_attn_tune(2, 0) == 0


@pytest.mark.parametrize("Hq,Hkv", [(8, 2), (8, 1), (16, 2), (4, 1)])
@pytest.mark.parametrize("splits", [1, 3, 0, 64])
def test_matrix_core_form_vs_oracle(matrix_core_form, Hq, Hkv, splits):
    """Same cases as the dot-product form, lengths around the 32-row stage and the 256-row page."""
    ctxs = [0, 1, 15, 16, 17, 31, 32, 33, 255, 256, 257, 700, 2049]
    case = make_case(len(ctxs), Hq, Hkv, 128, 256, ctxs, seed=7 + Hq * 3 + Hkv)
    assert not dot_form(case)
    got, want, (kc, vc), (kc_ref, vc_ref) = run_both(case, splits)
    check_output(got, case, what=f"matrix-core Hq={Hq} Hkv={Hkv} splits={splits}")
    assert torch.equal(vc.cpu(), vc_ref)
    assert torch.equal(kc.cpu(), kc_ref)


def test_matrix_core_form_edge_cases(matrix_core_form):
    """Unallocated pages, duplicate rows, poisoned slots, small pages (multiples of the 32-row
    stage; a 16-row page falls back to the dot-product form), no rotary + custom scale."""
    from micro_vllm_amd.core.paged_attention import paged_attn_decode

    # small pages
    for page, ctxs in [(32, [5, 100, 333, 47]), (64, [64, 63, 65, 1000]), (16, [5, 100])]:
        case = make_case(len(ctxs), 8, 2, 128, page, ctxs, seed=page)
        assert dot_form(case) == (page == 16)
        for splits in (1, 8, 0):
            got, _, (kc, vc), (kc_ref, vc_ref) = run_both(case, splits)
            check_output(got, case, what=f"page={page} splits={splits}")
            assert torch.equal(kc.cpu(), kc_ref) and torch.equal(vc.cpu(), vc_ref)
    # the block table entry of the new token is -1: write dropped, key ignored, no fault
    case = make_case(2, 8, 2, 128, 256, [256, 100], seed=13)
    case["bt"][0, 1] = -1
    kc, vc = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
    got = paged_attn_decode(case["q"][:, 0].to(DEV), case["k"][:, 0].to(DEV), case["v"][:, 0].to(DEV),
                            kc, vc, case["seqlens"].to(DEV), case["bt"].to(DEV),
                            *[t.to(DEV) for t in (case["cos"], case["sin"])])
    assert bool(torch.isfinite(got.float()).all())
    assert torch.equal(kc[case["bt"][0, 0].item()].cpu(), case["k_cache"][case["bt"][0, 0].item()])
    # duplicates
    case = make_case(1, 8, 1, 128, 256, [300], seed=5)
    for key in ("q", "k", "v", "bt", "seqlens"):
        case[key] = case[key].repeat(4, *([1] * (case[key].dim() - 1)))
    got, _, (kc, vc), (kc_ref, vc_ref) = run_both(case, 0)
    check_output(got, case, what="duplicates")
    assert torch.equal(got[0], got[1]) and torch.equal(got[0], got[3])
    assert torch.equal(kc.cpu(), kc_ref) and torch.equal(vc.cpu(), vc_ref)
    # no rotary, custom scale
    case = make_case(3, 8, 2, 128, 256, [10, 200, 513], seed=8, rotary=False)
    kc, vc = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
    got = paged_attn_decode(case["q"][:, 0].to(DEV), case["k"][:, 0].to(DEV), case["v"][:, 0].to(DEV),
                            kc, vc, case["seqlens"].to(DEV), case["bt"].to(DEV), softmax_scale=0.2)
    check_output(got, case, what="custom scale", scale=0.2)


def test_matrix_core_form_merge_kernels_agree_and_poison_is_ignored(matrix_core_form):
    from micro_vllm_amd import _native as nv

    case = make_case(6, 8, 2, 128, 256, [0, 63, 64, 1000, 2049, 4096], seed=55)
    outs = {}
    try:
        for mode in (1, 0, 1):
            assert nv.lib().mv_paged_attn_tune(0, mode) == 0
            for splits in (2, 7, 64):
                got, _, _, _ = run_both(case, splits)
                outs.setdefault(splits, []).append(got.clone())
    finally:
        nv.lib().mv_paged_attn_tune(0, 1)
    for splits, lst in outs.items():
        assert torch.equal(lst[0], lst[1]) and torch.equal(lst[0], lst[2]), splits
    test_slots_that_do_not_count_may_hold_anything(None, 128, 8, 2)


def test_calls_on_two_streams_do_not_share_state(mv):
    """Two split-KV launches in flight at once on two streams, each with its own workspace (the
    default workspace is per stream): every output equals the one a lone call produces.  The split
    merge keeps its arrival tickets in the workspace, so nothing on the device is shared."""
    from micro_vllm_amd.core.paged_attention import paged_attn_decode

    cases = [make_case(4, 8, 2, 128, 256, [4096, 3000, 4000, 2500], seed=71),
             make_case(3, 8, 2, 128, 256, [4095, 3500, 100], seed=72)]
    dev = []
    for c in cases:
        dev.append({k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in c.items()})

    def launch(d, kc, vc):
        return paged_attn_decode(d["q"][:, 0], d["k"][:, 0], d["v"][:, 0], kc, vc, d["seqlens"], d["bt"],
                                 d["cos"], d["sin"], num_splits=16)

    alone = [launch(d, d["k_cache"].clone(), d["v_cache"].clone()).clone() for d in dev]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    caches = [(d["k_cache"].clone(), d["v_cache"].clone()) for d in dev]
    torch.cuda.synchronize()
    outs = [[], []]
    for _ in range(25):
        for i in (0, 1):
            with torch.cuda.stream(streams[i]):
                outs[i].append(launch(dev[i], *caches[i]).clone())
    torch.cuda.synchronize()
    for i in (0, 1):
        for o in outs[i]:
            assert torch.equal(o, alone[i])
        check_output(outs[i][0], cases[i], what=f"stream {i}")


def _slab_case(B, Hq, Hkv, D, hidden, bias, seed):
    """A QKV projection of B rows and the cache state it feeds: returns what both routes need."""
    from micro_vllm_amd.kernel.linear import StreamedWeight

    g = torch.Generator().manual_seed(seed)
    ctxs = [int(c) for c in torch.randint(0, 900, (B,), generator=g)]
    ctxs[0], ctxs[-1] = 0, 256
    case = make_case(B, Hq, Hkv, D, 256, ctxs, seed=seed + 1)
    n = (Hq + 2 * Hkv) * D
    x = torch.randn(B, hidden, generator=g).to(torch.bfloat16).to(DEV)
    w = StreamedWeight.from_linear((torch.randn(n, hidden, generator=g) * 0.05).to(torch.bfloat16).to(DEV))
    b = (torch.randn(n, generator=g) * 0.5).to(torch.bfloat16).to(DEV) if bias else None
    return case, x, w, b


@pytest.mark.parametrize("B,Hq,Hkv,D,hidden,bias,force_mfma", [
    (32, 8, 2, 128, 1024, False, False), (32, 8, 2, 128, 1024, True, True), (8, 8, 1, 128, 2048, True, False),
    (17, 8, 2, 64, 512, True, False), (5, 7, 1, 64, 448, True, False), (32, 32, 8, 128, 4096, False, False)])
def test_attention_fed_from_projection_slabs_equals_the_two_step_route(mv, B, Hq, Hkv, D, hidden, bias, force_mfma):
    """q | k | v summed from the QKV projection's k-slabs inside the attention kernel, bias included:
    the output and the cache rows it appends are the bits the projection's own slab reduction
    followed by the plain attention call produce."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core.paged_attention import paged_attn_decode, paged_attn_decode_slabs
    from micro_vllm_amd.kernel.linear import wgemm, wgemm_slabs

    L = nv.lib()
    case, x, w, b = _slab_case(B, Hq, Hkv, D, hidden, bias, seed=B + Hq + D)
    if force_mfma:
        assert L.mv_paged_attn_tune(2, 2) == 0
    try:
        to = lambda t: t.to(DEV)
        n = (Hq + 2 * Hkv) * D
        ws = torch.zeros(L.mv_wgemm_postnorm_workspace_bytes(B, n, hidden) // 4 + 64, dtype=torch.float32, device=DEV)
        outs = []
        for route in ("two steps", "slabs"):
            kc, vc = to(case["k_cache"]), to(case["v_cache"])
            for splits in (1, 3):
                kc.copy_(case["k_cache"]); vc.copy_(case["v_cache"])
                if route == "slabs":
                    ks = wgemm_slabs(x, w, ws)
                    assert ks == L.mv_wgemm_slab_count(B, n, hidden) >= 1
                    out = paged_attn_decode_slabs(ws, ks, b, Hq, kc, vc, to(case["seqlens"]), to(case["bt"]),
                                                  to(case["cos"]), to(case["sin"]), num_splits=splits)
                else:
                    qkv = wgemm(x, w, torch.empty(B, n, dtype=torch.bfloat16, device=DEV), bias=b, workspace=ws,
                                plan_slabs=True)
                    out = paged_attn_decode(
                        qkv[:, :Hq * D].unflatten(1, (Hq, D)), qkv[:, Hq * D:(Hq + Hkv) * D].unflatten(1, (Hkv, D)),
                        qkv[:, (Hq + Hkv) * D:].unflatten(1, (Hkv, D)), kc, vc, to(case["seqlens"]), to(case["bt"]),
                        to(case["cos"]), to(case["sin"]), num_splits=splits)
                outs.append((route, splits, out.clone(), kc.clone(), vc.clone()))
        for splits in (1, 3):
            a = next(o for o in outs if o[0] == "two steps" and o[1] == splits)
            s = next(o for o in outs if o[0] == "slabs" and o[1] == splits)
            assert torch.equal(a[2], s[2]), f"output differs at {splits} split(s)"
            assert torch.equal(a[3], s[3]) and torch.equal(a[4], s[4]), "appended cache rows differ"
    finally:
        if force_mfma:
            assert L.mv_paged_attn_tune(2, 0) == 0


def test_slab_entry_point_rejects_bad_arguments(mv):
    from micro_vllm_amd import _native as nv

    L = nv.lib()
    t = torch.zeros(4096, dtype=torch.float32, device=DEV)
    i = torch.zeros(64, dtype=torch.int32, device=DEV)
    args = lambda slabs, ks: (slabs, ks, None, t.data_ptr(), t.data_ptr(), None, None, 0, i.data_ptr(), i.data_ptr(), 4,
                              t.data_ptr(), None, 0, 1, 2, 1, 64, 16, 1, 0.125, 1, None)
    assert L.mv_paged_attn_decode_slabs(*args(None, 1)) == nv.MV_ERR_INVALID_ARG
    assert L.mv_paged_attn_decode_slabs(*args(t.data_ptr(), 0)) == nv.MV_ERR_INVALID_ARG
    assert L.mv_paged_attn_decode_slabs(*args(t.data_ptr() + 4, 1)) == nv.MV_ERR_INVALID_ARG


@pytest.mark.parametrize("ks", [1, 3, 8, 11])
def test_slab_sums_of_any_count_follow_the_slab_order(mv, ks):
    """Slab counts up to and past the eight the kernel requests ahead: q | k | v must be the slabs
    added in order (fp32), one rounding, the bias, a second rounding - formed here with tensor
    operations and fed to the plain entry point."""
    from micro_vllm_amd.core.paged_attention import paged_attn_decode, paged_attn_decode_slabs

    B, Hq, Hkv, D = 6, 8, 2, 128
    n = (Hq + 2 * Hkv) * D
    g = torch.Generator().manual_seed(100 + ks)
    case = make_case(B, Hq, Hkv, D, 256, [0, 5, 255, 256, 300, 1000], seed=ks)
    # slabs of very different magnitudes, so that the order of the additions matters
    slabs = torch.randn(ks, B, n, generator=g) * torch.logspace(-3, 1, ks).reshape(ks, 1, 1)
    bias = (torch.randn(n, generator=g) * 0.5).to(torch.bfloat16)
    acc = torch.zeros(B, n)
    for s in range(ks):
        acc = acc + slabs[s]
    qkv = (acc.to(torch.bfloat16).float() + bias.float()).to(torch.bfloat16).to(DEV)
    to = lambda t: t.to(DEV)
    outs = []
    for route in (0, 1):
        kc, vc = to(case["k_cache"]), to(case["v_cache"])
        if route == 0:
            out = paged_attn_decode(
                qkv[:, :Hq * D].unflatten(1, (Hq, D)), qkv[:, Hq * D:(Hq + Hkv) * D].unflatten(1, (Hkv, D)),
                qkv[:, (Hq + Hkv) * D:].unflatten(1, (Hkv, D)), kc, vc, to(case["seqlens"]), to(case["bt"]),
                to(case["cos"]), to(case["sin"]), num_splits=2)
        else:
            out = paged_attn_decode_slabs(to(slabs).contiguous(), ks, to(bias), Hq, kc, vc, to(case["seqlens"]),
                                          to(case["bt"]), to(case["cos"]), to(case["sin"]), num_splits=2)
        outs.append((out.clone(), kc, vc))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
