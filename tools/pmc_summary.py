"""Sum rocprofv3 --pmc counter_collection.csv files per counter for kernels whose name contains KEY:
   python tools/pmc_summary.py KEY file1.csv [file2.csv ...]
Also prints the dispatches' average duration (End_Timestamp - Start_Timestamp of the counter run) and,
when SQ_VALU_MFMA_BUSY_CYCLES was collected, the matrix pipes' busy share:
   SQ_VALU_MFMA_BUSY_CYCLES / (duration x 1024 SIMDs x 2.4 GHz)
(the counter sums the cycles each of the chip's 1024 SIMDs had a matrix instruction executing; 2.4 GHz
is the peak engine clock, so with the clock lower under matrix load the true share is higher)."""
import csv, sys
from collections import defaultdict
key = sys.argv[1]
tot, n = defaultdict(float), defaultdict(int)
dur = {}
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if key in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
            dur[(path, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for c in sorted(tot):
    print(f"{c:32s} {tot[c]:18.0f}  over {n[c]} dispatches  ({tot[c]/n[c]:.0f} per dispatch)")
if dur:
    avg = sum(dur.values()) / len(dur)
    print(f"dispatch duration under the counters: {avg * 1e6:.1f} us average over {len(dur)} dispatches")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in tot:
        per = tot["SQ_VALU_MFMA_BUSY_CYCLES"] / n["SQ_VALU_MFMA_BUSY_CYCLES"]
        print(f"matrix pipes busy: {per / (avg * 1024 * 2.4e9) * 100:.1f} % of SIMD-cycles at 2.4 GHz "
              f"({per:.0f} busy SIMD-cycles per dispatch / ({avg * 1e6:.1f} us x 1024 SIMDs x 2.4 GHz))")
