"""Split counts for the matrix-core decode attention (mode 2) next to the dot form (mode 1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
from micro_vllm_amd import _native as nv
L = nv.lib()
for mode in (1, 2):
    L.mv_paged_attn_tune(2, mode)
    print("dot form" if mode == 1 else "matrix-core form", flush=True)
    for sp in (16, 32, 64):
        run(8, 8, 1, 128, 8192, layers=16, splits=sp)
    for sp in (2, 4, 8):
        run(16, 32, 8, 128, 4096, layers=8, splits=sp)
    for sp in (1, 2, 4):
        run(32, 32, 8, 128, 4096, layers=8, splits=sp)
    for sp in (1, 2):
        run(64, 32, 8, 128, 2048, layers=8, splits=sp)
    for sp in (16, 32):
        run(16, 4, 1, 128, 2048, layers=16, splits=sp)
L.mv_paged_attn_tune(2, 0)
