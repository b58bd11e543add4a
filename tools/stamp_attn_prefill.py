"""Where the waves of the pipelined prefill attention spend a tile step (stamped build):
   python -c "import sys; sys.path.insert(0, 'micro-vllm_amd'); import build; \
              build.build_variant('pfstamp', ['MV_PFX_STAMP'], only=('attn_prefill.hip',))"
   python tools/stamp_attn_prefill.py [S]
8B head shape, one sequence.  Stamps are shader-clock ticks of the eight waves of workgroup 0 (the
heaviest query block): step start | after the staging requests and the rescale | end of S^T stretch
(first block) | end of O^T stretch | (max finish) start of second block's S^T | its end | end of
its O^T | after the LDS stores and the barrier."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["MVLLM_LIB"] = os.path.join(ROOT, "micro-vllm_amd", "lib", "libmvllm_pfstamp.so")
sys.path.insert(0, ROOT)
import numpy as np
import torch
import micro_vllm_amd
from micro_vllm_amd import _native as nv
from micro_vllm_amd.core.prefill_attention import paged_attn_prefill_rows

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
hq, hkv, d, B = 32, 8, 128, 1
dev = "cuda"
pages = (S + 255) // 256
kc = torch.randn(pages, 256, hkv, d, device=dev).to(torch.bfloat16)
vc = torch.randn(pages, 256, hkv, d, device=dev).to(torch.bfloat16)
q = torch.randn(B * S, hq * d, device=dev).to(torch.bfloat16)
out = torch.empty_like(q)
bt = torch.randperm(pages, device=dev).to(torch.int32).reshape(B, pages)
sl = torch.zeros(B, dtype=torch.int32, device=dev)
L = nv.lib()
L.mv_debug_read_pfx_stamps.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
for rep in range(3):
    paged_attn_prefill_rows(q, B, S, hq, d, kc, vc, sl, bt, out)
    torch.cuda.synchronize()
buf = np.zeros(8 * 16 * 8, dtype=np.uint64)
assert L.mv_debug_read_pfx_stamps(buf.ctypes.data, buf.nbytes) == 0
st = buf.reshape(8, 16, 8).astype(np.float64)
names = ["requests+rescale", "S^T a", "O^T a", "max+top b", "S^T b", "O^T b", "stores+barrier"]
ticks = st[:, 1:, 0] - st[:, :-1, 0]
print(f"a tile step takes {np.median(ticks):.0f} shader-clock ticks (64 MFMAs of 32 cycles per SIMD = 2048; "
      f"the clock ran at 2.03 GHz under this kernel when it was read against the 100 MHz counter)")
for w in range(8):
    seg = np.diff(st[w], axis=1)
    print(f"wave {w}: " + "  ".join(f"{n} {np.median(seg[:, i]):5.0f}" for i, n in enumerate(names)))
print("start of step relative to wave 0 (ticks, median):", [int(np.median(st[w, :, 0] - st[0, :, 0])) for w in range(8)])

# ---- the launch as a whole: when every workgroup ran, where, and for how long
L.mv_debug_read_pfx_workgroups.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
wg = np.zeros(1024 * 3, dtype=np.uint64)
assert L.mv_debug_read_pfx_workgroups(wg.ctypes.data, wg.nbytes) == 0
n = (S // 64) * hkv
wg = wg.reshape(1024, 3)[:n]
t0 = wg[:, 0].min()
start = (wg[:, 0] - t0).astype(np.float64) / 100.0      # us
end = (wg[:, 1] - t0).astype(np.float64) / 100.0
tiles = (S // 64) - np.arange(n) // hkv                  # key tiles of workgroup i (heaviest first)
dur = end - start
A = np.vstack([tiles, np.ones(n)]).T
slope, icpt = np.linalg.lstsq(A, dur, rcond=None)[0]
print(f"launch: {end.max():.1f} us from the first workgroup's start to the last one's end; per workgroup "
      f"{slope:.3f} us per key tile + {icpt:.2f} us fixed (least squares over {n} workgroups)")
hw = wg[:, 2]
cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)   # xcc, se, cu
busy = {}
for i in range(n):
    busy.setdefault(int(cu[i]), []).append((start[i], end[i], int(tiles[i])))
tot = np.array([sum(e - s for s, e, _ in v) for v in busy.values()])
last = np.array([max(e for _, e, _ in v) for v in busy.values()])
print(f"{len(busy)} CUs seen; busy time per CU min {tot.min():.1f} median {np.median(tot):.1f} max {tot.max():.1f} us; "
      f"last end per CU min {last.min():.1f} median {np.median(last):.1f} max {last.max():.1f} us")
print("workgroups per CU:", np.bincount([len(v) for v in busy.values()]))
print("first starts (us):", np.sort(start)[:4], " starts of the second round:", np.sort(start)[256:260])
xcc = ((hw >> 32) & 0xf).astype(int)
for x in range(8):
    m = xcc == x
    if m.any():
        per_tile = (dur[m] - icpt) / tiles[m]
        big = m & (tiles >= 16)
        print(f"XCC {x}: {m.sum():3d} workgroups, kv heads {sorted(set((np.arange(n)[m] % hkv).tolist()))}, "
              f"us per key tile (workgroups of 16+ tiles) median {np.median(((dur - icpt) / tiles)[big]):.3f}, last end {end[m].max():.1f} us")
