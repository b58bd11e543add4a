"""Does the 2 KB row stride of 8 interleaved kv heads cost bandwidth?  Same bytes, same number of
(sequence, kv head) pairs: 32 sequences x 8 kv heads against 256 sequences x 1 kv head."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
from micro_vllm_amd import _native as nv
L = nv.lib()
for mode in (1, 2):
    L.mv_paged_attn_tune(2, mode)
    print("dot form" if mode == 1 else "matrix-core form", flush=True)
    run(32, 32, 8, 128, 4096, layers=4, splits=1)
    run(256, 4, 1, 128, 4096, layers=4, splits=1)
    run(64, 16, 4, 128, 4096, layers=4, splits=1)
L.mv_paged_attn_tune(2, 0)
