"""Where one wave of the prefill GEMM spends a k stage (stamped build, see build.build_variant):
   python tools/stamp_gemm_prefill.py [variant]     (default variant "stamp")
Build the variants first, e.g.
   python -c "import sys; sys.path.insert(0, 'micro-vllm_amd'); import build; \
              build.build_variant('stamp', ['MV_GEMM_STAMP'], only=('gemm_prefill.hip',)); \
              build.build_variant('stamp_now', ['MV_GEMM_STAMP', 'MV_EXP_NO_WLOAD'], only=('gemm_prefill.hip',)); \
              build.build_variant('stamp_nox', ['MV_GEMM_STAMP', 'MV_EXP_NO_XREAD'], only=('gemm_prefill.hip',))"
Phases per stage: 0->1 fragment reads + MFMA issue (incl. waits for weights), 1->2 wait for the x
tile two stages ahead, 2->3 barrier.  Measured with both operand paths removed a stage takes 1430
ticks, with either one present about 2800-2900, with both 3800: the kernel is bound by operand
delivery, not by the matrix pipe."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MVLLM_LIB"] = os.path.join(ROOT, "micro-vllm_amd", "lib", "libmvllm_%s.so" % (sys.argv[1] if len(sys.argv) > 1 else "stamp"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
dev = "cuda"
m, n, k = 4096, 28672, 4096
w = StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, True)
x = torch.randn(m, k, device=dev).to(torch.bfloat16)
y = torch.empty(m, n // 2, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    gemm_prefill(x, w, y, swiglu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    gemm_prefill(x, w, y, swiglu=True)
e1.record(); torch.cuda.synchronize()
t_kernel = e0.elapsed_time(e1) / 5 * 1e-3
L = nv.lib()
L.mv_debug_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(4 * 512, dtype=np.uint64)
assert L.mv_debug_read_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(512, 4)[: k // 64].astype(np.int64)
d01, d12, d23 = st[:, 1] - st[:, 0], st[:, 2] - st[:, 1], st[:, 3] - st[:, 2]
nxt = st[1:, 0] - st[:-1, 3]
tot = st[-1, 3] - st[0, 0]
wgs = (m // 128) * (n // 256)
rounds = wgs / 512.0
print(f"kernel {t_kernel*1e6:.0f} us, {wgs} workgroups = {rounds:.1f} rounds of 2 per CU -> {t_kernel/rounds*1e6:.1f} us per workgroup "
      f"-> tick = {t_kernel/rounds/tot*1e9:.3f} ns ({tot/(t_kernel/rounds)/1e9:.2f} GHz)")
print(f"stages {len(st)}, total {tot} ticks, per stage {tot/len(st):.0f}")
print(f"  reads+MFMA issue  mean {d01[2:].mean():7.0f}  p90 {np.percentile(d01[2:], 90):7.0f}")
print(f"  x wait            mean {d12[2:].mean():7.0f}  p90 {np.percentile(d12[2:], 90):7.0f}")
print(f"  barrier            mean {d23[2:].mean():7.0f}  p90 {np.percentile(d23[2:], 90):7.0f}")
print(f"  stage-to-stage gap mean {nxt.mean():7.0f}")
print("first 12 stages:", [(int(a), int(b), int(c)) for a, b, c in zip(d01[:12], d12[:12], d23[:12])])
