"""The C-ABI library loads and exports every entry point include/mvllm.h declares (CPU only:
symbols are resolved, nothing is launched)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mvllm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mv_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = declared_functions()
    for must in ("mv_rmsnorm", "mv_rmsnorm_residual", "mv_swiglu", "mv_wgemm",
                 "mv_paged_attn_decode", "mv_store_kv", "mv_graph_launch", "mv_embedding"):
        assert must in names


def test_library_exports_every_declared_symbol(mv):
    from micro_vllm_amd import _native as nv

    assert os.path.exists(nv.LIB_PATH), "build the library first: python micro-vllm_amd/build.py"
    handle = ctypes.CDLL(nv.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(handle, n)]
    assert not missing, f"declared in mvllm.h but not exported: {missing}"


def test_ctypes_prototypes_cover_the_header(mv):
    from micro_vllm_amd import _native as nv

    assert sorted(nv.PROTOTYPES) == declared_functions()
    lib = nv.lib()
    assert lib.mv_version() >= 100
    assert lib.mv_status_string(0) == b"ok"
    assert lib.mv_status_string(-2) == b"unsupported shape"


def test_argument_validation_without_a_gpu(mv):
    """Entry points reject bad arguments before touching the device."""
    from micro_vllm_amd import _native as nv

    lib = nv.lib()
    assert lib.mv_rmsnorm(None, None, None, 1, 128, 128, 128, 1e-6, None) == nv.MV_ERR_INVALID_ARG
    assert lib.mv_swiglu(None, None, 1, 8, None) == nv.MV_ERR_INVALID_ARG
    # misaligned / unsupported shapes
    assert lib.mv_shuffle_weight(16, 32, 15, 32, None) == nv.MV_ERR_UNSUPPORTED
    assert lib.mv_wgemm(16, 32, 16, None, None, 0, 16, 16, 65, 16, 32, None, 0, 0, None) == nv.MV_ERR_UNSUPPORTED
    assert lib.mv_paged_attn_decode(16, 0, 16, 16, 0, 16, 16, None, None, 0, 16, 16, 1, 16, None, 0,
                                    1, 4, 4, 96, 256, 1, 1.0, 1, None) == nv.MV_ERR_UNSUPPORTED
    with pytest.raises(AssertionError):
        nv.check("mv_rmsnorm", nv.MV_ERR_INVALID_ARG)
