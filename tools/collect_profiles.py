"""Turn what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into the committed summaries
under profiles/:  python tools/collect_profiles.py r2
  <tag>_bench_kernel_stats.csv          rocprofv3 --stats of the default bench run
  <tag>_bench_trace_summary.txt         per kernel and launch shape: average duration, gap, share
  <tag>_bench_line_under_rocprof.json   the bench line that run printed
  <tag>_pmc_{fetch,write}_by_kernel_bs{1,32}.csv   FETCH_SIZE / WRITE_SIZE sums per kernel
  <tag>_pmc_mfma_{gemm,attn}_prefill.txt           matrix-pipe counters of the prefill kernels
  traffic.json                          HBM bytes per launch of the dominant kernels (bench.py reads it)"""
import csv, datetime, glob, json, os, shutil, subprocess, sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
py = sys.executable

shutil.copy(os.path.join(src, "trace", f"{tag}_kernel_stats.csv"), os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
out = subprocess.run([py, os.path.join(ROOT, "tools", "trace_summary.py"), os.path.join(src, "trace", f"{tag}_kernel_trace.csv"), "--tail", "1.0"],   # every leg of the bench line
                     capture_output=True, text=True, check=True).stdout
open(os.path.join(dst, f"{tag}_bench_trace_summary.txt"), "w").write(out)
line = [l for l in open(os.path.join(src, "trace_bench.json")) if l.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_bench_line_under_rocprof.json"), "w"), indent=1)

def by_kernel(path, counter, out_path):
    tot, n = OrderedDict(), {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"][:70]
        tot[k] = tot.get(k, 0.0) + float(r["Counter_Value"])
        n[k] = n.get(k, 0) + 1
    with open(out_path, "w") as f:
        f.write(f"kernel,launches,{counter}_KiB_total,{counter}_KiB_per_launch\n")
        for k in sorted(tot, key=lambda k: -tot[k]):
            f.write(f'"{k}",{n[k]},{tot[k]:.1f},{tot[k] / n[k]:.2f}\n')

today = datetime.date.today().isoformat()
for bs, kernel in ((1, "wgemm_dot_kernel"), (32, "wgemm_mfma_kernel")):
    files = {}
    for c, short in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        files[c] = glob.glob(os.path.join(src, f"pmc_{c}_bs{bs}", "*counter_collection.csv"))[0]
        by_kernel(files[c], c, os.path.join(dst, f"{tag}_pmc_{short}_by_kernel_bs{bs}.csv"))
    for k in (kernel, "paged_attn_decode_kernel"):
        subprocess.run([py, os.path.join(ROOT, "tools", "pmc_traffic.py"), files["FETCH_SIZE"], files["WRITE_SIZE"],
                        f"{k}:llama3-8b:bs{bs}:tp1",
                        f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) of bench.py, {today}, "
                        f"tools/profile_round.sh {tag}; per-kernel sums in profiles/{tag}_pmc_*_by_kernel.csv"], check=True)
for what in ("gemm", "attn"):
    p = os.path.join(src, f"pmc_mfma_{what}_summary.txt")
    if os.path.exists(p):
        head = ("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 "
                "SQ_WAIT_ANY SQ_WAIT_INST_ANY of tools/%s (4096 tokens), sums over the dispatches of the kernel;\n"
                "SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles (16 per v_mfma_f32_16x16x32_bf16), SQ_WAVE_CYCLES / "
                "SQ_WAIT_* quad-cycles of wave lifetime.\n" % ("bench_gemm_prefill.py" if what == "gemm" else "attn_prefill_only.py"))
        log = open(os.path.join(src, f"pmc_mfma_{what}.log")).read()
        open(os.path.join(dst, f"{tag}_pmc_mfma_{what}_prefill.txt"), "w").write(head + open(p).read() + "\ntimings of that (profiled) run:\n" + log)
print("profiles/ refreshed from", src)
