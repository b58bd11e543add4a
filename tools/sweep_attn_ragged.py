"""Split counts when the rows of a batch have different lengths (uniform in [256, 4096])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
for bs in (8, 16, 32, 40, 64):
    for sp in (0, 1, 2, 4, 8):
        run(bs, 32, 8, 128, 4096, layers=6, splits=sp, ragged_seed=3)
