"""Time line of one workgroup of the wide prefill GEMM across its tiles (stamp build of gemm_prefill.hip):
python tools/stamp_gemm_wide.py [N] [K]   (M = 4096; SWIGLU=1 for the fused activation).  Prints, per tile, the time spent in its k-tiles,
in its output stores, and from the end of the stores to the top of the next tile."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib
build = importlib.import_module("micro-vllm_amd.build")
if not os.environ.get("MVLLM_LIB"):
    os.environ["MVLLM_LIB"] = build.build_variant("wstamp", ["MV_GEMM_WSTAMP"], only=("gemm_prefill.hip",))
    if not os.path.exists("/dev/kfd"):
        print("built", os.environ["MVLLM_LIB"]); sys.exit(0)
import numpy as np
import torch
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
from micro_vllm_amd import _native as nv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 28672
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
M = 4096
SW = os.environ.get("SWIGLU", "0") == "1"
w = StreamedWeight(torch.empty(N * K, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02), N, K, SW)
x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
y = torch.empty(M, N // 2 if SW else N, dtype=torch.bfloat16, device="cuda")
for _ in range(3):
    gemm_prefill(x, w, y, swiglu=SW)
torch.cuda.synchronize()
buf = np.zeros(4096, dtype=np.uint64)
lib = nv.lib()
lib.mv_debug_read_wide_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.mv_debug_read_wide_stamps(buf.ctypes.data, buf.nbytes) == 0
wg = int(os.environ.get("WG", "77"))
t0 = int(buf[0])
last = max(i for i in range(2048) if buf[i])
print(f"shader clock over the workgroup's run: {(int(buf[2048 + last]) - int(buf[2048])) / ((int(buf[last]) - t0) / 100.0) / 1e3:.3f} GHz")
prev_end = None
for piece in range(4):
    s = buf[piece * 512: (piece + 1) * 512].astype(np.int64)
    if s[0] == 0:
        break
    nz = [i for i in range(512) if s[i]]
    nkt = nz[-1] - 1                      # the last two stamps are nkt and nkt + 1
    tops = s[0:nkt:2]
    d = np.diff(tops) / 100.0
    gap = "" if prev_end is None else f"; {(int(s[0]) - prev_end) / 100.0:5.2f} us after the previous epilogue"
    print(f"piece {piece}: {nkt} k-tiles, starts {(int(s[0]) - t0) / 100.0:8.2f} us{gap}; k-tiles {(int(s[nkt]) - int(s[0])) / 100.0:7.2f} us "
          f"(pairs: first 4 {np.round(d[:4], 2)}, median {np.median(d):.2f}, last 4 {np.round(d[-4:], 2)}); "
          f"epilogue {(int(s[nkt + 1]) - int(s[nkt])) / 100.0:6.2f} us")
    prev_end = int(s[nkt + 1])
