"""Time the four decode projections (M=1, weights rotated over >600 MB so every launch streams
from HBM) for several builds of the library:  python tools/sweep_libs.py base p1 p2 ...
(`name` -> micro-vllm_amd/lib/libmvllm_<name>.so, see build.build_variant; `main` = libmvllm.so)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "tools"))
import torch
from bench_wgemm import timed_graph
from micro_vllm_amd.kernel.linear import StreamedWeight, wgemm, wgemm_norm
dev = "cuda"
tot = 0.0
for (n, k, mode) in [(6144, 4096, "norm"), (4096, 4096, "plain"), (28672, 4096, "addnorm"), (4096, 14336, "plain")]:
    copies = max(4, int(600e6 / (n * k * 2)) + 1)
    ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device=dev).normal_(0, 0.02), n, k) for _ in range(copies)]
    x = torch.randn(1, k, device=dev).to(torch.bfloat16); lw = torch.ones(k, dtype=torch.bfloat16, device=dev)
    r = torch.randn(1, k, device=dev).to(torch.bfloat16); so = torch.empty_like(r)
    y = torch.empty(1, n, dtype=torch.bfloat16, device=dev)
    def fn():
        for w in ws:
            if mode == "norm": wgemm_norm(x, lw, 1e-5, w, y)
            elif mode == "addnorm": wgemm_norm(x, lw, 1e-5, w, y, add_r=r, sum_out=so)
            else: wgemm(x, w, y)
    t = timed_graph(fn) / copies
    tot += t
    print("  N=%%5d K=%%5d %%-8s %%6.2f us %%5.2f TB/s" %% (n, k, mode, t * 1e6, n * k * 2 / t / 1e12), flush=True)
print("  sum %%.2f us" %% (tot * 1e6), flush=True)
''' % (ROOT, ROOT)
for name in sys.argv[1:]:
    lib = os.path.join(ROOT, "micro-vllm_amd", "lib", "libmvllm.so" if name == "main" else f"libmvllm_{name}.so")
    print(name, flush=True)
    subprocess.run([sys.executable, "-c", CODE], env=dict(os.environ, MVLLM_LIB=lib), check=False)
