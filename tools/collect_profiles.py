"""Turn what tools/profile_round.sh left under gpurun_out/prof_<tag>/ into the committed summaries
under profiles/:  python tools/collect_profiles.py r3
  <tag>_<leg>_kernel_stats.csv      rocprofv3 --stats of the run of ONE leg (bs1, bs32-even, bs32-ragged,
                                    prefill, engine)
  <tag>_<leg>_trace_summary.txt     per kernel and launch shape: average duration, gap, share
  <tag>_<leg>_line_under_rocprof.json   the bench line that run printed
  <tag>_pmc_{fetch,write}_by_kernel_bs{1,32}.csv   FETCH_SIZE / WRITE_SIZE sums per kernel
  <tag>_pmc_fetch_calibration.json  FETCH_SIZE of a reader of a known byte count in the attention's pattern
  <tag>_pmc_mfma_{gemm,attn}_prefill.txt   matrix-pipe counters of the prefill kernels, busy share, and the
                                    un-profiled time of the same tool
  traffic.json                      HBM bytes per launch of the dominant kernels (bench.py reads it)"""
import csv, datetime, glob, json, os, re, shutil, subprocess, sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles")
py = sys.executable

for leg in ("bs1", "bs32-even", "bs32-ragged", "prefill", "engine"):
    d = os.path.join(src, f"trace_{leg}")
    stats = glob.glob(os.path.join(d, "**", f"{tag}_kernel_stats.csv"), recursive=True)
    trace = glob.glob(os.path.join(d, "**", f"{tag}_kernel_trace.csv"), recursive=True)
    if not stats or not trace:
        print("missing leg", leg)
        continue
    shutil.copy(stats[0], os.path.join(dst, f"{tag}_{leg}_kernel_stats.csv"))
    # the decode legs: the steady state (last half); prefill / engine: everything the run did
    tail = "0.5" if leg.startswith("bs") else "1.0"
    out = subprocess.run([py, os.path.join(ROOT, "tools", "trace_summary.py"), trace[0], "--tail", tail],
                         capture_output=True, text=True, check=True).stdout
    open(os.path.join(dst, f"{tag}_{leg}_trace_summary.txt"), "w").write(out)
    lines = [l for l in open(os.path.join(src, f"trace_{leg}.json")) if l.startswith("{")]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(dst, f"{tag}_{leg}_line_under_rocprof.json"), "w"), indent=1)


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


# ---- calibration: FETCH_SIZE of tools/micro/kv_stream, which reads a known number of bytes in the
# attention kernel's pattern (256-byte row segments at a 2 KiB stride through a shuffled page table)
calib = {}
for bs in (1, 32):
    files = glob.glob(os.path.join(src, f"pmc_calib_bs{bs}", "**", "*counter_collection.csv"), recursive=True)
    log = os.path.join(src, f"pmc_calib_bs{bs}.log")
    if not files or not os.path.exists(log):
        continue
    m = re.search(r"calib: .* (\d+) bytes per launch", open(log).read())
    if not m:
        continue
    known = float(m.group(1))
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(files[0]))
            if r["Counter_Name"] == "FETCH_SIZE" and "stream_kernel" in r["Kernel_Name"]]
    if vals:
        kib = sum(vals) / len(vals)
        calib[f"bs{bs}"] = {"known_bytes_per_launch": known, "FETCH_SIZE_KiB_per_launch": kib, "launches": len(vals),
                            "bytes_per_counted_KiB": known / kib,
                            "what": "tools/micro/kv_stream BS 4096 calib SPLITS under rocprofv3 --pmc FETCH_SIZE"}
if calib:
    json.dump(calib, open(os.path.join(dst, f"{tag}_pmc_fetch_calibration.json"), "w"), indent=1)

today = datetime.date.today().isoformat()
for bs, kernel in ((1, "wgemm_dot_kernel"), (32, "wgemm_mfma_kernel")):
    files = {}
    for c, short in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        got = glob.glob(os.path.join(src, f"pmc_{c}_bs{bs}", "**", "*counter_collection.csv"), recursive=True)
        if not got:
            break
        files[c] = got[0]
        by_kernel(files[c], c, os.path.join(dst, f"{tag}_pmc_{short}_by_kernel_bs{bs}.csv"))
    if len(files) < 2:
        continue
    for k in (kernel, "paged_attn_decode_kernel"):
        args = [py, os.path.join(ROOT, "tools", "pmc_traffic.py"), files["FETCH_SIZE"], files["WRITE_SIZE"],
                f"{k}:llama3-8b:bs{bs}:tp1",
                f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (one pass each) of bench.py, {today}, "
                f"tools/profile_round.sh {tag}; per-kernel sums in profiles/{tag}_pmc_*_by_kernel.csv"]
        if "paged_attn" in k and f"bs{bs}" in calib:
            args.append(str(calib[f"bs{bs}"]["bytes_per_counted_KiB"]))
        subprocess.run(args, check=True)
for what, tool in (("gemm", "bench_gemm_prefill.py 4096"), ("attn", "attn_prefill_only.py 4096 1")):
    p = os.path.join(src, f"pmc_mfma_{what}_summary.txt")
    if os.path.exists(p):
        head = (f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 "
                f"SQ_WAIT_ANY SQ_WAIT_INST_ANY of tools/{tool}, sums over the dispatches of the kernel;\n"
                "SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles (16 per v_mfma_f32_16x16x32_bf16), SQ_WAVE_CYCLES / "
                "SQ_WAIT_* quad-cycles of wave lifetime.\n")
        log = open(os.path.join(src, f"pmc_mfma_{what}.log")).read()
        plain = os.path.join(src, f"plain_{what}.log")
        plain_txt = open(plain).read() if os.path.exists(plain) else "(not run)\n"
        open(os.path.join(dst, f"{tag}_pmc_mfma_{what}_prefill.txt"), "w").write(
            head + open(p).read() + "\nthe same tool WITHOUT the profiler, same box, just before:\n" + plain_txt
            + "\nunder the counters:\n" + log)
print("profiles/ refreshed from", src)
