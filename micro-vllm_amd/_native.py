"""ctypes binding of libmvllm.so (the C ABI declared in include/mvllm.h).

The library is the only compute path: if it is missing, `lib()` raises - there is
no PyTorch or CPU fallback behind these ops.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVLLM_LIB selects another build of the same ABI (kernel-parameter experiments)
LIB_PATH = os.environ.get("MVLLM_LIB") or os.path.join(_HERE, "lib", "libmvllm.so")

MV_OK = 0
MV_ERR_INVALID_ARG = -1
MV_ERR_UNSUPPORTED = -2
MV_ERR_LAUNCH = -3
MV_ERR_WORKSPACE = -4
MV_ERR_GRAPH = -5

MV_GEMM_DEFAULT = 0
MV_GEMM_SWIGLU = 1
MV_GEMM_OUT_F32 = 2
MV_GEMM_SLABS = 4
MV_GEMM_PLAN_SLABS = 8
MV_NORM_VAR_FROM_ROUNDED = 1



_P = c_void_p
_I = c_int
_L = c_int64
_F = c_float
_Z = c_size_t

# name -> (restype, argtypes); mirrors include/mvllm.h one to one
PROTOTYPES = {
    "mv_version": (_I, []),
    "mv_status_string": (c_char_p, [_I]),
    "mv_last_hip_error": (c_char_p, []),
    "mv_rmsnorm": (_I, [_P, _P, _P, _I, _I, _L, _L, _F, _P]),
    "mv_rmsnorm_residual": (_I, [_P, _P, _P, _P, _P, _I, _I, _L, _L, _L, _L, _F, _P]),
    "mv_swiglu": (_I, [_P, _P, _L, _I, _P]),
    "mv_add": (_I, [_P, _P, _P, _L, _L, _P]),
    "mv_embedding": (_I, [_P, _P, _P, _I, _I, _L, _P]),
    "mv_shuffle_weight": (_I, [_P, _P, _I, _I, _P]),
    "mv_interleave_rows8": (_I, [_P, _P, _P, _I, _I, _P]),
    "mv_wgemm_workspace_bytes": (_Z, [_I, _I, _I]),
    "mv_wgemm_tune": (_I, [_I, _I]),
    "mv_wgemm": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _P, _Z, _I, _P]),
    "mv_wgemm_norm": (_I, [_P, _L, _P, _L, _P, _F, _P, _L, _I, _P, _P, _P, _L, _P, _L, _I, _I, _I, _P,
                           _Z, _I, _P]),
    "mv_wgemm_slab_count": (_I, [_I, _I, _I]),
    "mv_wgemm_postnorm_workspace_bytes": (_Z, [_I, _I, _I]),
    "mv_wgemm_postnorm": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _P, _F, _P, _L, _I, _I, _I, _I, _P, _Z, _P]),
    "mv_paged_attn_decode_workspace_bytes": (_Z, [_I, _I, _I, _I]),
    "mv_paged_attn_decode_pick_splits": (_I, [_I, _I, _I, _I]),
    "mv_paged_attn_tune": (_I, [_I, _I]),
    "mv_paged_attn_decode_form": (_I, [_I, _I, _I, _I]),
    "mv_paged_attn_decode": (_I, [_P, _L, _P, _P, _L, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _Z,
                                  _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mv_paged_attn_decode_slabs": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _P, _P, _I, _P, _P, _Z,
                                        _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mv_qkv_attn_workspace_bytes": (_Z, []),
    "mv_qkv_attn_decode": (_I, [_P, _L, _P, _F, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _P, _I, _P, _P, _I, _P,
                                _P, _Z, _P, _Z, _P, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mv_paged_attn_decode_noalloc": (_I, [_P, _L, _P, _P, _L, _P, _P, _P, _P, _I, _P, _P, _I, _P,
                                          _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "mv_gemm_prefill": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P]),
    "mv_gemm_prefill_workspace_bytes": (_Z, [_I, _I, _I]),
    "mv_gemm_prefill_form": (_I, [_I, _I, _I]),
    "mv_debug_prefill_options": (None, [_I, _I]),
    "mv_paged_attn_prefill_form": (_I, [_I, _I, _I, _I]),
    "mv_gemm_prefill_ws": (_I, [_P, _L, _P, _P, _P, _L, _P, _L, _I, _I, _I, _I, _P, _Z, _P]),
    "mv_rope_prefill": (_I, [_P, _L, _P, _L, _P, _P, _I, _L, _I, _I, _I, _P]),
    "mv_rope_store_prefill": (_I, [_P, _L, _P, _P, _I, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "mv_paged_attn_prefill": (_I, [_P, _L, _P, _P, _P, _P, _I, _P, _L, _I, _I, _I, _I, _I, _I, _I, _F, _P]),
    "mv_store_kv": (_I, [_P, _P, _L, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "mv_argmax_workspace_bytes": (_Z, []),
    "mv_argmax": (_I, [_P, _L, _I, _I, _P, _P, _Z, _P]),
    "mv_sample_topk_topp": (_I, [_P, _L, _I, _I, _P, _P, _I, _P, _P, _P]),
    "mv_sample_or_greedy": (_I, [_P, _L, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "mv_allreduce_handle_bytes": (_Z, []),
    "mv_allreduce_init": (_I, [_I, _I, _I, _I, ctypes.POINTER(c_void_p)]),
    "mv_allreduce_get_handle": (_I, [_P, _P]),
    "mv_allreduce_connect": (_I, [_P, _P]),
    "mv_allreduce_status": (_I, [_P, ctypes.POINTER(c_int)]),
    "mv_allreduce_set_timeout_ms": (_I, [_P, ctypes.c_double]),
    "mv_allreduce_destroy": (_I, [_P]),
    "mv_allreduce_step": (_I, [_P, _P, _L, _P, _L, _I, _I, _P]),
    "mv_allgather_init": (_I, [_I, _I, _I, _I, ctypes.POINTER(c_void_p)]),
    "mv_allgather_rows": (_I, [_P, _P, _L, _P, _L, _I, _I, _P]),
    "mv_allreduce_addnorm": (_I, [_P, _P, _L, _P, _I, _P, _L, _P, _L, _P, _F, _P, _L, _I, _I, _I, _P]),
    "mv_graph_begin_capture": (_I, [_P]),
    "mv_graph_end_capture": (_I, [_P, ctypes.POINTER(c_void_p)]),
    "mv_graph_launch": (_I, [_P, _P]),
    "mv_graph_destroy": (_I, [_P]),
}

_lib = None


class NativeError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, fn, status, detail=""):
        self.fn, self.status = fn, status
        super().__init__(f"{fn} failed: status {status} ({detail})")


def lib():
    """Load libmvllm.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python micro-vllm_amd/build.py` "
                "(hipcc --offload-arch=gfx950). There is no fallback path.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(fn_name, status):
    """Map a negative status to the Python exception the reference's ops would raise."""
    if status == MV_OK:
        return
    L = lib()
    text = L.mv_status_string(status).decode()
    if status == MV_ERR_LAUNCH or status == MV_ERR_GRAPH:
        text += ": " + L.mv_last_hip_error().decode()
        raise NativeError(fn_name, status, text)
    if status in (MV_ERR_INVALID_ARG, MV_ERR_UNSUPPORTED):
        # the reference asserts on device/shape/dtype mismatches
        raise AssertionError(f"{fn_name}: {text}")
    raise NativeError(fn_name, status, text)


def stream_ptr(stream=None):
    """hipStream_t of `stream` (default: torch's current stream) as an int."""
    import torch

    s = stream if stream is not None else torch.cuda.current_stream()
    return s.cuda_stream


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


_stream_ws = {}


def stream_workspace(kind, device, nbytes):
    """A zero-initialised device buffer of at least `nbytes` for kernels that meet in a workspace
    (split merges, arrival tickets), one per (purpose, device, current stream): calls on different
    streams never share one.  The library's kernels leave the ticket area zero after every launch,
    so the buffer is zero-filled only when it is (re)allocated."""
    import torch

    if nbytes <= 0:
        return None
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (kind, idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _stream_ws.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=dev)
        _stream_ws[key] = ws
    return ws
