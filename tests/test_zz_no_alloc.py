"""mv_paged_attn_decode_noalloc - the entry point that allocates nothing - through the C ABI,
against mv_paged_attn_decode on the same inputs: same bits in the output and in both caches, on two
calls in a row, the second with a smaller input (the built-in scratch keeps what the first call
left).  Also: the oracle bar of the attention tests, free device memory unchanged across each
call, and a clear refusal of sizes beyond the scratch."""
import pytest
import torch

from test_gpu_attention import check_output, make_case

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _args(case, dev_tensors, out):
    q, k, v, kc, vc, cos, sin, sl, bt = dev_tensors
    B, _, Hq, D = case["q"].shape
    Hkv = case["k"].shape[2]
    NB, page = kc.shape[0], kc.shape[1]
    head = (q.data_ptr(), Hq * D, k.data_ptr(), v.data_ptr(), Hkv * D, kc.data_ptr(), vc.data_ptr(),
            cos.data_ptr(), sin.data_ptr(), cos.shape[0], sl.data_ptr(), bt.data_ptr(), bt.stride(0),
            out.data_ptr())
    tail = (B, Hq, Hkv, D, page, NB, float(D) ** -0.5)
    return head, tail


def _to_dev(case):
    return tuple(case[n].to(DEV).contiguous() for n in ("q", "k", "v", "k_cache", "v_cache", "cos", "sin",
                                                        "seqlens", "bt"))


def _run_pair(nv, case, splits, check_free=True):
    """(out, k_cache, v_cache) of the existing entry point and of the no-allocation one."""
    L = nv.lib()
    B, _, Hq, D = case["q"].shape
    stream = nv.stream_ptr()
    res = []
    for noalloc in (False, True):
        t = _to_dev(case)
        out = torch.zeros(B, Hq, D, dtype=torch.bfloat16, device=DEV)
        head, tail = _args(case, t, out)
        if noalloc:
            torch.cuda.synchronize()
            free_before = torch.cuda.mem_get_info()[0]
            rc = L.mv_paged_attn_decode_noalloc(*head, *tail, splits, stream)
            nv.check("mv_paged_attn_decode_noalloc", rc)
            torch.cuda.synchronize()
            if check_free:   # nothing was taken from (or returned to) the device around the call
                assert torch.cuda.mem_get_info()[0] == free_before
        else:
            need = L.mv_paged_attn_decode_workspace_bytes(B, Hq, D, splits)
            ws = torch.zeros(need // 4 + 4, dtype=torch.float32, device=DEV)
            rc = L.mv_paged_attn_decode(*head, ws.data_ptr(), ws.numel() * 4, *tail, splits, stream)
            nv.check("mv_paged_attn_decode", rc)
        torch.cuda.synchronize()
        res.append((out, t[3], t[4]))
    return res


def test_no_alloc_entry_point_equals_the_allocating_one(mv):
    from micro_vllm_amd import _native as nv

    big = make_case(6, 8, 2, 128, 256, [0, 63, 64, 1000, 2049, 4096], seed=55)
    small = make_case(2, 8, 2, 128, 256, [5, 300], seed=56)
    _run_pair(nv, small, 2, check_free=False)     # first call on the device: address look-up, code load
    for case, splits, what in ((big, 7, "first call"), (small, 3, "second, smaller call"),
                               (big, 32, "third call, more splits"), (small, 1, "one split")):
        (ref_out, ref_k, ref_v), (out, kc, vc) = _run_pair(nv, case, splits)
        assert torch.equal(out, ref_out), what
        assert torch.equal(kc, ref_k) and torch.equal(vc, ref_v), what
        check_output(out, case, what=what)


def test_no_alloc_entry_point_refuses_what_does_not_fit(mv):
    from micro_vllm_amd import _native as nv

    L = nv.lib()
    case = make_case(2, 8, 2, 128, 256, [5, 300], seed=57)
    t = _to_dev(case)
    out = torch.zeros(2, 8, 128, dtype=torch.bfloat16, device=DEV)
    head, tail = _args(case, t, out)
    # the same pointers with a claimed batch whose partials cannot fit: refused before any launch
    big_tail = (64, 64, 16, 128, 256, t[3].shape[0], 0.1)
    assert L.mv_paged_attn_decode_workspace_bytes(64, 64, 128, 32) > 8 << 20
    rc = L.mv_paged_attn_decode_noalloc(*head, *big_tail, 32, nv.stream_ptr())
    assert rc == nv.MV_ERR_WORKSPACE
    text = L.mv_last_hip_error().decode()
    assert "mv_paged_attn_decode_noalloc" in text and "8388608" in text
    torch.cuda.synchronize()
    # and the entry point still works afterwards
    (ref_out, _, _), (got, _, _) = _run_pair(nv, case, 4)
    assert torch.equal(got, ref_out)
