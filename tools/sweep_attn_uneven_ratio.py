"""Where the finer split starts to pay: rows uniform in [lo, 4096] for several lo, whole rows against
the uneven-row split count (2 at 32 rows, 4 at 64), so that the longest/mean ratio at which the
decode runner switches graphs rests on a measurement."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
for bs, fine in ((32, 2), (64, 4)):
    for lo in (3584, 3072, 2560, 2048, 1024):
        for sp in (1, fine):
            run(bs, 32, 8, 128, 4096, layers=6, splits=sp, ragged_seed=3, ragged_lo=lo)
