"""Where the workgroups of the M=1 projection launches spend their time (stamped build):
   python -c "import sys; sys.path.insert(0, 'micro-vllm_amd'); import build; \
              build.build_variant('wstamp', ['MV_WGEMM_STAMP'], only=('wgemm.hip',))"
   python tools/stamp_wgemm.py
A hipGraph of the four decode projections of 8 layers (distinct weights, so every launch streams
from HBM) is replayed; the stamps of its 32 launches are the constant 100 MHz counter, comparable
across workgroups and launches.  Printed per launch, in microseconds after the launch's earliest
workgroup entry: gap to the previous launch's last exit, then min / median / max over workgroups of
each phase."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MVLLM_LIB"] = os.path.join(ROOT, "micro-vllm_amd", "lib", "libmvllm_wstamp.so")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from micro_vllm_amd import _native as nv
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, wgemm_norm
dev = "cuda"
H, I, QKV = 4096, 14336, 6144
LAYERS = 8
def mk(n, k, gu=False):
    return StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k, gu)
layers = [(mk(QKV, H), mk(H, H), mk(2 * I, H, True), mk(H, I)) for _ in range(LAYERS)]
x = torch.randn(1, H, device=dev).to(torch.bfloat16); lw = torch.ones(H, dtype=torch.bfloat16, device=dev)
r = torch.randn(1, H, device=dev).to(torch.bfloat16); so = torch.empty_like(r)
qkv = torch.empty(1, QKV, dtype=torch.bfloat16, device=dev)
att = torch.randn(1, H, device=dev).to(torch.bfloat16)
h2 = torch.empty(1, H, dtype=torch.bfloat16, device=dev)
act = torch.empty(1, I, dtype=torch.bfloat16, device=dev)
def step():
    for (wq, wo, wgu, wd) in layers:
        wgemm_norm(x, lw, 1e-5, wq, qkv)
        wgemm(att, wo, h2)
        wgemm_norm(h2, lw, 1e-5, wgu, act, add_r=r, sum_out=so, swiglu=True)
        wgemm(act, wd, h2)
L = nv.lib()
if "--waves" in sys.argv:
    assert L.mv_wgemm_tune(2, int(sys.argv[sys.argv.index("--waves") + 1])) == 0
if "--contig" in sys.argv:
    assert L.mv_wgemm_tune(4, 1) == 0
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step(); side.synchronize()
    nv.check("b", L.mv_graph_begin_capture(side.cuda_stream)); step()
    h = ctypes.c_void_p(); nv.check("e", L.mv_graph_end_capture(side.cuda_stream, ctypes.byref(h)))
    for _ in range(3): L.mv_graph_launch(h, side.cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(side)
    for _ in range(10): L.mv_graph_launch(h, side.cuda_stream)
    e1.record(side); side.synchronize()
print(f"graph of {4*LAYERS} launches: {e0.elapsed_time(e1)*1e3/10/LAYERS:.2f} us per layer")
L.mv_debug_read_wgemm_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
buf = np.zeros(32 * 256 * 8, dtype=np.uint64)
assert L.mv_debug_read_wgemm_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(32, 256, 8).astype(np.float64)
# slot of a launch = launch counter & 31; the eager warm-up call used the first 32, the capture the next
names = ["QKV+norm", "O", "gate/up+addnorm", "down"]
order = np.argsort([st[i][:, 0][st[i][:, 0] > 0].min() for i in range(32)])
prev_end = None
agg = {}
for idx in order:
    rec = st[idx]
    have = rec[:, 0] > 0
    t0 = rec[have, 0].min()
    end = rec[have, 4].max()
    kind = names[idx % 4]
    row = dict(gap=(t0 - prev_end) / 100 if prev_end else float("nan"), dur=(end - t0) / 100)
    for slot, nm in ((0, "entry"), (1, "x staged"), (2, "first tiles"), (3, "stream done"), (4, "exit")):
        col = rec[have, slot]
        col = col[col > 0]
        if len(col):
            us = (col - t0) / 100
            row[nm] = (us.min(), np.median(us), us.max())
    row["exit_by_xcd"] = [float(np.mean((rec[have, 4] - t0)[np.arange(256)[have] % 8 == x]) / 100) for x in range(8)]
    row["exit_all"] = (rec[:, 4] - t0) / 100
    agg.setdefault(kind, []).append(row)
    prev_end = end
for kind in names:
    rows = agg[kind][1:]
    print(f"{kind}: {len(rows)} launches, first entry -> last exit {np.mean([r['dur'] for r in rows]):.2f} us, "
          f"gap after previous launch {np.nanmean([r['gap'] for r in rows]):.2f} us")
    for nm in ("entry", "x staged", "first tiles", "stream done", "exit"):
        vals = np.array([r[nm] for r in rows if nm in r])
        if len(vals):
            print(f"    {nm:12s} min {vals[:,0].mean():6.2f}  median {vals[:,1].mean():6.2f}  max {vals[:,2].mean():6.2f}")
    print("    mean exit by blockIdx %% 8: " + " ".join("%.2f" % np.mean([r["exit_by_xcd"][x] for r in rows]) for x in range(8)))
    ex = np.array([r["exit_all"] for r in rows])          # [launch, block]
    per_block = ex.mean(axis=0)
    print("    exit averaged over launches, per block: min %.2f max %.2f; blocks 0-7 of each 64: %s" % (
        per_block.min(), per_block.max(), " ".join("%.1f" % per_block[i] for i in (0, 1, 2, 3, 64, 65, 128, 129))))
    print("    spread within one launch (max - mean): %.2f us" % np.mean(ex.max(axis=1) - ex.mean(axis=1)))
