"""Decode attention of one tensor-parallel rank of the 8B shape (bs=1, ctx 4096) over split counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
for tp in (2, 4, 8):
    for sp in (0, 8, 16, 32, 64):
        run(1, 32 // tp, 8 // tp, 128, 4096, splits=sp)
for sp in (0, 8, 16, 32):
    run(16, 4, 1, 128, 4096, splits=sp, layers=16)
