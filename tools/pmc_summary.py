"""Sum rocprofv3 --pmc counter_collection.csv files per counter for kernels whose name contains KEY:
   python tools/pmc_summary.py KEY file1.csv [file2.csv ...]"""
import csv, sys
from collections import defaultdict
key = sys.argv[1]
tot, n = defaultdict(float), defaultdict(int)
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        if key in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
            n[r["Counter_Name"]] += 1
for c in sorted(tot):
    print(f"{c:32s} {tot[c]:18.0f}  over {n[c]} dispatches  ({tot[c]/n[c]:.0f} per dispatch)")
