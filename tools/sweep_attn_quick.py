"""Decode attention, the shapes the split rule was tuned on (automatic split count)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
run(1, 32, 8, 128, 4096)
run(1, 32, 8, 128, 4096, splits=16)
run(1, 32, 8, 128, 1024)
run(1, 32, 8, 128, 8192, layers=16)
run(8, 8, 1, 128, 8192, layers=16)
run(1, 32, 4, 64, 2048)
run(4, 32, 8, 128, 4096, layers=16)
run(32, 32, 8, 128, 4096, layers=8)
