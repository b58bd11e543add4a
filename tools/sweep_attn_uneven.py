"""Split counts for batch sizes whose (sequence, kv head) pairs do not divide the CUs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
for bs in (24, 40, 48, 56):
    for sp in (0, 1, 2, 3, 4, 8):
        run(bs, 32, 8, 128, 4096, layers=6, splits=sp)
