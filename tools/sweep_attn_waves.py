"""Decode attention at small batch over split counts (narrow workgroups)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run, nv
L = nv.lib()
for ctx in (1024, 2048, 4096, 8192, 16384):
    for sp in (8, 16, 32, 64):
        if ctx // sp >= 64:
            run(1, 32, 8, 128, ctx, splits=sp, layers=16 if ctx > 4096 else 32)
for sp in (16, 32, 64):
    run(8, 8, 1, 128, 8192, layers=16, splits=sp)
for sp in (8, 16, 32):
    run(1, 32, 4, 64, 2048, splits=sp)
for sp in (4, 8, 16):
    run(4, 32, 8, 128, 4096, splits=sp, layers=16)
