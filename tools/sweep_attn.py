"""Decode attention alone over batch sizes / split counts (uses tools/bench_attn.run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
for sp in (1, 2, 3, 4, 8):
    run(32, 32, 8, 128, 4096, layers=8, splits=sp)
for sp in (2, 4, 8):
    run(16, 32, 8, 128, 2048, layers=8, splits=sp)
for sp in (0, 8, 16, 32):
    run(8, 8, 1, 128, 8192, layers=16, splits=sp)
