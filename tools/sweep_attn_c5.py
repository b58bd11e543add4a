"""Decode attention of one rank of the 70B shape under 8-way tensor parallelism (8 query heads on
one kv head, bs 8, ctx 8192) and of the 8B shape at bs 16 / ctx 2048, over split counts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
for sp in (0, 8, 16, 32, 64):
    run(8, 8, 1, 128, 8192, splits=sp, layers=16)
for sp in (0, 8, 16, 32):
    run(16, 4, 1, 128, 2048, splits=sp, layers=16)
