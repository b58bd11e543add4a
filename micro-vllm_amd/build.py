"""Build the gfx950 shared library (libmvllm.so) from csrc/*.hip with hipcc.

Run as a script (`python micro-vllm_amd/build.py`) or call build(); hipcc
cross-compiles without a GPU, so this works on a CPU-only machine.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libmvllm.so")

SOURCES = [
    "runtime.hip",
    "norm.hip",
    "elementwise.hip",
    "wgemm.hip",
    "attn_decode.hip",
    "kv_store.hip",
    "attn_prefill.hip",
    "gemm_prefill.hip",
    "sampler.hip",
    "allreduce.hip",
    "qkv_attn.hip",
]

# -ffp-contract=off: rounding is part of the parity contract; FMAs are written explicitly
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-gpu-rdc",
    "-Wall",
    "-Wno-unused-function",
]


# per-source additions.  attn_prefill.hip: the pipelined attention deals its vector work out slot
# by slot beside the MFMAs; the SLP vectoriser would gather the row-sum additions of many slots
# into one chain of packed additions (slower beside MFMAs than the scalar form, and serial)
FILE_FLAGS = {
    "attn_prefill.hip": ["-fno-slp-vectorize"],
}


def _flags_for(source_name):
    return FLAGS + FILE_FLAGS.get(source_name, [])


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_variant(name, defines, only=("wgemm.hip",), verbose=False):
    """A second library `lib/libmvllm_<name>.so` with extra -D flags for the listed sources (the
    rest are reused from the main build); select it at run time with MVLLM_LIB=<path>."""
    build(verbose=verbose)
    hipcc = _hipcc()
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        if s in only:
            obj = os.path.join(OBJDIR, s.replace(".hip", f".{name}.o"))
            cmd = [hipcc] + _flags_for(s) + [f"-D{d}" for d in defines] + ["-c", src, "-o", obj]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stdout)
        objs.append(obj)
    out = os.path.join(LIBDIR, f"libmvllm_{name}.so")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stdout)
    return out


def build(force=False, verbose=True, jobs=None):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "mvllm.h"))
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    todo = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            todo.append((src, obj))

    def compile_one(pair):
        src, obj = pair
        cmd = [hipcc] + _flags_for(os.path.basename(src)) + ["-c", src, "-o", obj]
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}")
        if verbose and r.stdout.strip():
            print(r.stdout)

    jobs = jobs or min(4, max(1, (os.cpu_count() or 2) // 2))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(compile_one, todo))

    if force or todo or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print("[build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
