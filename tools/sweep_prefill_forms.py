"""Narrow (128 x 256 tiles, two workgroups per CU) against wide (256 x 256, one per CU) prefill GEMM
per projection shape and token count; each form runs in a process of its own because the choice
is read from MVLLM_PREFILL_WIDE once:  python tools/sweep_prefill_forms.py [M ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ms = sys.argv[1:] or ["256", "512", "1024", "1536", "2048", "3072", "4096", "8192"]
res = {}
for form in ("0", "1"):
    env = dict(os.environ, MVLLM_PREFILL_WIDE=form)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_gemm_prefill.py")] + ms, env=env,
                         capture_output=True, text=True, timeout=600).stdout
    for line in out.splitlines():
        if "N=" in line:
            p = line.replace("=", " ").replace(":", " ").split()
            res[(int(p[1]), int(p[3]), int(p[5]), form)] = float(p[6])
print(f"{'M':>6} {'N':>6} {'K':>6} {'narrow us':>10} {'wide us':>10}  tiles(wide)")
for (m, n, k, form) in sorted(res):
    if form == "0":
        a, b = res[(m, n, k, "0")], res.get((m, n, k, "1"), float("nan"))
        print(f"{m:6d} {n:6d} {k:6d} {a:10.1f} {b:10.1f}  {((m + 255) // 256) * ((n + 255) // 256):5d} {'wide' if b < a else 'narrow'}", flush=True)
