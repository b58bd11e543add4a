"""Decode attention at batch 32 and SHORT contexts (what a serving batch of 512-token prompts sees):
split counts 1, 2, 4 per context length.  python tools/sweep_attn_short.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_attn import run
for bs in (32, 16, 8):
    for ctx in (300, 576, 1100, 2048):
        for sp in (1, 2, 4, 8):
            if bs * 8 * sp <= 2048:
                run(bs, 32, 8, 128, ctx, layers=16, splits=sp)
