"""QKV projection + paged attention in one launch (mv_qkv_attn_decode) against the two launches it
stands for (mv_wgemm_norm, mv_paged_attn_decode), through the C ABI.

The fused launch is built to form every sum in the order the stand-alone kernels form it, so the
bar is bit equality: the projection's output rows, the attention output, and the whole key/value
cache (appended row included) must be identical; the attention output is in addition held to the
oracle like the stand-alone kernel's (tests/test_gpu_attention.py).  Sizes are those whose grid
fills the GPU (the launch is only offered there): the Llama-3-8B head shape at 1, 2 and 4 rows, a
multi-head shape, a two-heads-per-kv-head shape and a 64-wide head."""
import ctypes

import pytest
import torch

from test_gpu_attention import check_output, make_case

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(B, Hq, Hkv, D, hidden, ctxs, seed, bias):
    from micro_vllm_amd.kernel.linear import StreamedWeight

    case = make_case(B, Hq, Hkv, D, 256, ctxs, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    n = (Hq + 2 * Hkv) * D
    x = (torch.randn(B, hidden, generator=g) * 1.5).to(torch.bfloat16).to(DEV)
    lnw = (1 + 0.1 * torch.randn(hidden, generator=g)).to(torch.bfloat16).to(DEV)
    w = StreamedWeight.from_linear((torch.randn(n, hidden, generator=g) * 0.03).to(torch.bfloat16).to(DEV))
    b = (torch.randn(n, generator=g) * 0.1).to(torch.bfloat16).to(DEV) if bias else None
    return case, x, lnw, w, b, n


def _separate(case, x, lnw, w, b, n, Hq, Hkv, D, splits, kc, vc, ws):
    from micro_vllm_amd.core.paged_attention import paged_attn_decode
    from micro_vllm_amd.kernel.linear import wgemm_norm

    B = x.shape[0]
    qkv = torch.full((B, n), 7.0, dtype=torch.bfloat16, device=DEV)
    wgemm_norm(x, lnw, 1e-5, w, qkv, bias=b)
    out = paged_attn_decode(
        qkv[:, :Hq * D].unflatten(1, (Hq, D)), qkv[:, Hq * D:(Hq + Hkv) * D].unflatten(1, (Hkv, D)),
        qkv[:, (Hq + Hkv) * D:].unflatten(1, (Hkv, D)), kc, vc, case["seqlens"].to(DEV), case["bt"].to(DEV),
        case["cos"].to(DEV), case["sin"].to(DEV), num_splits=splits, workspace=ws)
    return qkv, out


def _workspaces(B, Hq, D, splits):
    from micro_vllm_amd import _native as nv

    L = nv.lib()
    ws = torch.zeros((L.mv_paged_attn_decode_workspace_bytes(B, Hq, D, splits) + 3) // 4, dtype=torch.float32,
                     device=DEV)
    sync = torch.zeros((L.mv_qkv_attn_workspace_bytes() + 3) // 4, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32).pin_memory()
    return ws, sync, status


@pytest.mark.parametrize("B,Hq,Hkv,D,hidden,bias", [
    (1, 32, 8, 128, 4096, False),     # Llama-3-8B, the headline configuration
    (2, 32, 8, 128, 4096, False),
    (4, 32, 8, 128, 4096, True),      # with a QKV bias (Qwen)
    (1, 16, 16, 128, 2048, True),     # one query head per kv head
    (1, 16, 8, 128, 2048, False),     # two
    (1, 64, 16, 64, 4096, False),     # heads of 64
])
def test_one_launch_equals_the_two_launches_bit_for_bit(mv, B, Hq, Hkv, D, hidden, bias):
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core.paged_attention import qkv_attn_decode

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    splits = cus // (B * Hkv)
    ctx_sets = [[4095, 3000, 257, 4096][:B], [3, 700, 0, 255][:B], [2049, 1, 1024, 640][:B]]
    ws_f, sync, status = _workspaces(B, Hq, D, splits)
    for ci, ctxs in enumerate(ctx_sets):
        case, x, lnw, w, b, n = _setup(B, Hq, Hkv, D, hidden, ctxs, seed=10 * ci + B + Hq, bias=bias)
        ws_s, _, _ = _workspaces(B, Hq, D, splits)
        kc_s, vc_s = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
        qkv_s, out_s = _separate(case, x, lnw, w, b, n, Hq, Hkv, D, splits, kc_s, vc_s, ws_s)

        kc_f, vc_f = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
        qkv_f = torch.full((B, n), -3.0, dtype=torch.bfloat16, device=DEV)
        out_f = torch.full((B, Hq, D), -5.0, dtype=torch.bfloat16, device=DEV)
        ok = qkv_attn_decode(x, lnw, 1e-5, w, b, qkv_f, Hq, kc_f, vc_f, case["seqlens"].to(DEV),
                             case["bt"].to(DEV), case["cos"].to(DEV), case["sin"].to(DEV), splits, out_f,
                             ws_f, sync, status)
        assert ok, "the library refused a shape it is documented to fuse"
        torch.cuda.synchronize()
        assert int(status[0]) == 0, "a wait inside the launch timed out"
        assert torch.equal(qkv_f, qkv_s), f"projection rows differ (ctxs {ctxs})"
        assert torch.equal(out_f, out_s), f"attention output differs (ctxs {ctxs})"
        assert torch.equal(kc_f, kc_s) and torch.equal(vc_f, vc_s), "key/value cache differs"
        # and the attention output against the oracle, given the q | k | v the projection produced
        case["q"] = qkv_f[:, :Hq * D].reshape(B, 1, Hq, D).cpu()
        case["k"] = qkv_f[:, Hq * D:(Hq + Hkv) * D].reshape(B, 1, Hkv, D).cpu()
        case["v"] = qkv_f[:, (Hq + Hkv) * D:].reshape(B, 1, Hkv, D).cpu()
        check_output(out_f.cpu(), case, what=f"fused attention B={B} Hq={Hq} Hkv={Hkv} D={D} ctxs={ctxs}")
    # the same workspace served every launch above: the epoch in it advanced once per launch
    assert int(sync[0]) == len(ctx_sets) and int(sync[64]) == 0
    assert nv.lib().mv_qkv_attn_workspace_bytes() <= sync.numel() * 4


def test_replayed_from_a_graph_and_shapes_alternating_on_one_workspace(mv):
    """Launches of different row counts share one flag workspace (epochs, not resets), eagerly and
    as a captured graph replayed several times with changing inputs."""
    from micro_vllm_amd import _native as nv
    from micro_vllm_amd.core.paged_attention import qkv_attn_decode

    Hq, Hkv, D, hidden = 32, 8, 128, 4096
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    L = nv.lib()
    sync = torch.zeros((L.mv_qkv_attn_workspace_bytes() + 3) // 4, dtype=torch.int32, device=DEV)
    status = torch.zeros(1, dtype=torch.int32).pin_memory()
    launches = 0
    for rnd, B in enumerate([1, 4, 2, 1]):
        splits = cus // (B * Hkv)
        ctxs = [1000 + 37 * rnd, 5, 4096, 300][:B]
        case, x, lnw, w, b, n = _setup(B, Hq, Hkv, D, hidden, ctxs, seed=100 + rnd, bias=False)
        ws_s, _, _ = _workspaces(B, Hq, D, splits)
        ws_f, _, _ = _workspaces(B, Hq, D, splits)
        kc_s, vc_s = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
        qkv_s, out_s = _separate(case, x, lnw, w, b, n, Hq, Hkv, D, splits, kc_s, vc_s, ws_s)
        kc_f, vc_f = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
        qkv_f = torch.zeros((B, n), dtype=torch.bfloat16, device=DEV)
        out_f = torch.zeros((B, Hq, D), dtype=torch.bfloat16, device=DEV)
        args = (x, lnw, 1e-5, w, b, qkv_f, Hq, kc_f, vc_f, case["seqlens"].to(DEV), case["bt"].to(DEV),
                case["cos"].to(DEV), case["sin"].to(DEV), splits, out_f, ws_f, sync, status)
        assert qkv_attn_decode(*args)
        launches += 1
        torch.cuda.synchronize()
        assert torch.equal(qkv_f, qkv_s) and torch.equal(out_f, out_s)
        if B == 1:
            # captured, then replayed with a different x every time
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                nv.check("begin", L.mv_graph_begin_capture(side.cuda_stream))
                assert qkv_attn_decode(*args)
                h = ctypes.c_void_p()
                nv.check("end", L.mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
                for rep in range(3):
                    x.copy_((torch.randn(B, hidden, generator=torch.Generator().manual_seed(500 + rep)) * 1.5)
                            .to(torch.bfloat16))
                    kc_f.copy_(case["k_cache"])
                    vc_f.copy_(case["v_cache"])
                    nv.check("launch", L.mv_graph_launch(h, side.cuda_stream))
                    launches += 1
                    side.synchronize()
                    kc_s.copy_(case["k_cache"])
                    vc_s.copy_(case["v_cache"])
                    qkv_s, out_s = _separate(case, x, lnw, w, b, n, Hq, Hkv, D, splits, kc_s, vc_s, ws_s)
                    side.synchronize()
                    assert torch.equal(qkv_f, qkv_s) and torch.equal(out_f, out_s), rep
                    assert torch.equal(kc_f, kc_s) and torch.equal(vc_f, vc_s)
                L.mv_graph_destroy(h)
            torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert int(status[0]) == 0 and int(sync[0]) == launches and int(sync[64]) == 0


def test_shapes_that_are_not_fused_are_refused_before_anything_runs(mv):
    from micro_vllm_amd.core.paged_attention import qkv_attn_decode

    cus = torch.cuda.get_device_properties(0).multi_processor_count
    for B, Hq, Hkv, D, hidden, splits in [
        (1, 8, 2, 128, 512, 4),                 # grid far below the CU count
        (1, 64, 8, 128, 8192, cus // 8),        # eight query heads per kv head: the matrix-core attention form
        (3, 32, 8, 128, 4096, 8),               # three rows
        (1, 32, 8, 128, 4096, 1),               # unsplit
    ]:
        case, x, lnw, w, b, n = _setup(B, Hq, Hkv, D, hidden, [100] * B, seed=7, bias=False)
        ws, sync, status = _workspaces(B, Hq, D, max(splits, 2))
        kc, vc = case["k_cache"].to(DEV), case["v_cache"].to(DEV)
        qkv = torch.full((B, n), 9.0, dtype=torch.bfloat16, device=DEV)
        out = torch.full((B, Hq, D), 9.0, dtype=torch.bfloat16, device=DEV)
        assert not qkv_attn_decode(x, lnw, 1e-5, w, b, qkv, Hq, kc, vc, case["seqlens"].to(DEV),
                                   case["bt"].to(DEV), case["cos"].to(DEV), case["sin"].to(DEV), splits, out,
                                   ws, sync, status)
        torch.cuda.synchronize()
        assert bool((qkv == 9.0).all()) and bool((out == 9.0).all()) and int(sync[0]) == 0
        assert torch.equal(kc.cpu(), case["k_cache"])
