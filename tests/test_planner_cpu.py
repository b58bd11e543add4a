"""Launch planning of the library, queried through the C ABI without a GPU (the planner assumes
256 CUs when it cannot ask a device - the MI355X count): invariants every caller relies on when it
sizes workspaces, and the choices the design notes quote for the Llama-3-8B shapes."""
import pytest

SHAPES_8B = [(6144, 4096), (4096, 4096), (28672, 4096), (4096, 14336), (128256, 4096)]
SHARDS = [(768, 4096), (4096, 512), (3584 * 2, 4096), (4096, 1792), (1280, 8192), (8192, 1024), (8192, 3584)]


def test_slab_counts_divide_k_and_workspaces_hold_them(mv):
    from micro_vllm_amd import _native as nv

    L = nv.lib()
    for n, k in SHAPES_8B + SHARDS:
        for m in range(1, 65):
            ks = L.mv_wgemm_slab_count(m, n, k)
            assert ks >= 1 and (k // 32) % ks == 0, (m, n, k, ks)
            # the slab forms (MV_GEMM_SLABS, mv_wgemm_postnorm) need room for every slab, one included
            assert L.mv_wgemm_postnorm_workspace_bytes(m, n, k) >= ks * m * n * 4
            # an ordinary call may be asked to cut k the same way (MV_GEMM_PLAN_SLABS)
            if ks > 1:
                assert L.mv_wgemm_workspace_bytes(m, n, k) >= ks * m * n * 4
    assert L.mv_wgemm_slab_count(0, 4096, 4096) == 0 and L.mv_wgemm_slab_count(8, 4095, 4096) == 0


def test_single_token_projections_stream_whole_strips(mv):
    """M <= 4 (the dot form): whole strips beat every k-split for the four decode projections."""
    from micro_vllm_amd import _native as nv

    L = nv.lib()
    for n, k in SHAPES_8B:
        for m in (1, 2, 3, 4):
            assert L.mv_wgemm_slab_count(m, n, k) == 1


def test_batched_projection_plans_quoted_in_the_design_notes(mv):
    from micro_vllm_amd import _native as nv

    L = nv.lib()
    # 32 rows: QKV in two k-slices (groups of three strips, one unit per CU), O and down in four
    assert L.mv_wgemm_slab_count(32, 6144, 4096) == 2
    assert L.mv_wgemm_slab_count(32, 4096, 4096) == 4
    assert L.mv_wgemm_slab_count(32, 4096, 14336) == 4
    # gate/up: 1792 strips = 7 x 256, groups of seven, never split
    for m in (5, 8, 16, 32):
        assert L.mv_wgemm_slab_count(m, 28672, 4096) == 1


def test_prefill_forms(mv):
    from micro_vllm_amd import _native as nv

    L = nv.lib()
    # projections: the 256 x 256 form once its tiles fill half the CUs and k is not cut
    assert L.mv_gemm_prefill_form(4096, 6144, 4096) == 2       # 192-column tiles: two full rounds
    assert L.mv_gemm_prefill_form(4096, 28672, 4096) == 1      # 256-column tiles: seven rounds against ten
    assert L.mv_gemm_prefill_form(2048, 28672, 4096) == 3      # 224-column tiles: four full rounds, not 3.5
    assert L.mv_gemm_prefill_form(2048, 4096, 4096) == 1       # 128 tiles
    assert L.mv_gemm_prefill_form(1024, 4096, 4096) == 0       # 64 tiles
    assert L.mv_gemm_prefill_form(128, 4096, 4096) == 0 and L.mv_gemm_prefill_workspace_bytes(128, 4096, 4096) > 0
    assert L.mv_gemm_prefill_form(4096, 4096, 192) == 0        # fewer than 8 k-tiles
    assert L.mv_gemm_prefill_form(4096, 4100, 4096) == 0       # not a shape the kernels take at all
    # left-over tiles cut along k inside the launch (a slot of 256 KiB per workgroup behind 4 KiB of counters)
    # where that beats a wider or narrower tile in whole rounds; nothing to cut where the rounds are whole
    cut = 4096 + 256 * 256 * 1024
    assert L.mv_gemm_prefill_form(3072, 6144, 4096) == 2 and L.mv_gemm_prefill_workspace_bytes(3072, 6144, 4096) == cut
    assert L.mv_gemm_prefill_workspace_bytes(4096, 6144, 4096) == 0      # 64 tiles of 192 columns per XCD: two rounds
    assert L.mv_gemm_prefill_workspace_bytes(4096, 28672, 4096) == 0     # seven rounds
    assert L.mv_gemm_prefill_workspace_bytes(2048, 28672, 4096) == 0     # four rounds of 224 columns, not 3.5 cut
    assert L.mv_gemm_prefill_form(3072, 28672, 4096) == 3 and L.mv_gemm_prefill_workspace_bytes(3072, 28672, 4096) == 0
    # two k-slices and a reduction launch where the tiles fill three eighths to half of the CUs and k is long
    assert L.mv_gemm_prefill_workspace_bytes(2048, 4096, 14336) == 4096 + 2 * 2048 * 4096 * 4
    # attention: 256-row workgroups for long chunks that fill their rounds
    assert L.mv_paged_attn_prefill_form(1, 4096, 32, 8) == 8
    assert L.mv_paged_attn_prefill_form(1, 3072, 32, 8) == 8
    assert L.mv_paged_attn_prefill_form(8, 1024, 32, 8) == 8
    assert L.mv_paged_attn_prefill_form(8, 511, 32, 8) == 4    # short chunks
    assert L.mv_paged_attn_prefill_form(1, 0, 32, 8) == 0


def test_decode_attention_split_choice(mv):
    from micro_vllm_amd import _native as nv

    L = nv.lib()
    assert L.mv_paged_attn_decode_pick_splits(1, 8, 128, 4352) == 32     # one workgroup per CU
    assert L.mv_paged_attn_decode_pick_splits(32, 8, 128, 4352) == 1
    for bs in (1, 2, 4, 8, 16, 32, 64):
        s = L.mv_paged_attn_decode_pick_splits(bs, 8, 128, 4352)
        ws = L.mv_paged_attn_decode_workspace_bytes(bs, 32, 128, s)
        # unsplit calls need none; split ones start with the 1 MiB of arrival tickets
        assert s >= 1 and (ws == 0 if s == 1 else ws >= (1 << 20) + bs * 32 * s * (128 + 1) * 4)
