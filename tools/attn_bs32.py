"""bs=32 (and 16, 8) decode attention at ctx 4096, 8B shape: python tools/attn_bs32.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
for bs in (32, 16, 8):
    run(bs, 32, 8, 128, 4096, layers=8)
