"""Does a projection stream faster when its weights were read shortly before (Infinity Cache, 256 MiB)?
   python tools/exp_mall_prefetch.py
Times `wgemm` at M = 1 for the O (34 MB), QKV (50 MB) and down (117 MB) shapes of the 8B model three ways,
each launch preceded by a 1 GB read of other memory (so nothing of its own survives from the last launch):
  cold      nothing else
  touched   a plain read of the weight tiles (torch sum: default cache policy) between the flush and the launch
  twice     the projection launched twice in a row; the second is timed (its own non-temporal loads before it)
HIP events around the timed launch only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import micro_vllm_amd
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm

dev = "cuda"
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
for name, n, k in (("O", 4096, 4096), ("QKV", 6144, 4096), ("down", 4096, 14336)):
    w = StreamedWeight.from_linear((torch.randn(n, k, device=dev) * 0.03).to(torch.bfloat16))
    x = torch.randn(1, k, device=dev).to(torch.bfloat16)
    y = torch.empty(1, n, dtype=torch.bfloat16, device=dev)
    tiles32 = w.tiles.view(torch.int32)
    res = {}
    for mode in ("cold", "touched", "twice"):
        ts = []
        for rep in range(8):
            flush.view(torch.int64).sum()
            if mode == "touched":
                tiles32.sum()
            if mode == "twice":
                wgemm(x, w, y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            wgemm(x, w, y)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        res[mode] = ts[len(ts) // 2]
    print(f"{name:5s} {n * k * 2 / 1e6:6.1f} MB   cold {res['cold']:6.2f} us   touched {res['touched']:6.2f} us   twice {res['twice']:6.2f} us")
