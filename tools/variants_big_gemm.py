"""Timing variants of the four-wave prefill GEMM (wrong results, timing only): builds one library per
variant on the CPU (python tools/variants_big_gemm.py build), then on the GPU box times the gate/up
projection with each, or with the ones named (python tools/variants_big_gemm.py [name ...])."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VARIANTS = {"bigbase": [], "nocopy": ["MV_BIG_NO_COPY"], "noread": ["MV_BIG_NO_READ"],
            "nocopyread": ["MV_BIG_NO_COPY", "MV_BIG_NO_READ"], "nobarrier": ["MV_BIG_NO_BARRIER"],
            "nomfma": ["MV_BIG_NO_MFMA"]}
if len(sys.argv) > 1 and sys.argv[1] == "build":
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "micro-vllm_amd", "build.py"))
    b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
    for name, defs in VARIANTS.items():
        print(b.build_variant(name, defs, only=("gemm_prefill.hip",)))
    sys.exit(0)
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
from micro_vllm_amd.kernel.linear import StreamedWeight, gemm_prefill
m, n, k = 4096, 28672, 4096
ws = [StreamedWeight(torch.empty(n * k, dtype=torch.bfloat16, device="cuda").normal_(0, 0.02), n, k, True) for _ in range(4)]
x = torch.randn(m, k, device="cuda").to(torch.bfloat16); y = torch.empty(m, n // 2, dtype=torch.bfloat16, device="cuda")
for w in ws: gemm_prefill(x, w, y, swiglu=True)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    for w in ws: gemm_prefill(x, w, y, swiglu=True)
e1.record(); torch.cuda.synchronize()
print(f"{os.environ.get('VARIANT'):12s} mband {os.environ.get('MVLLM_PREFILL_MBAND', '-'):3s} gate/up {e0.elapsed_time(e1) / 12 * 1e3:8.1f} us", flush=True)
''' % ROOT
for name in (sys.argv[1:] or VARIANTS):
    env = dict(os.environ, MVLLM_LIB=os.path.join(ROOT, "micro-vllm_amd", "lib", f"libmvllm_{name}.so"),
               MVLLM_PREFILL_FOUR_WAVES="1", VARIANT=name)
    subprocess.run([sys.executable, "-c", code], env=env, check=True, timeout=120)
