"""Does the row stride matter?  bs=32 x 8 kv heads (a head's rows 2 KB apart) against
bs=256 x 1 kv head (same workgroups and bytes, a head's rows back to back)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
run(32, 32, 8, 128, 4096, layers=4, splits=2)
run(256, 4, 1, 128, 4096, layers=4, splits=2)
run(32, 32, 8, 128, 4096, layers=4, splits=4)
run(256, 4, 1, 128, 4096, layers=4, splits=4)
