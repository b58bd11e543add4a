"""Summarise a rocprofv3 --kernel-trace CSV: per kernel (and per launch shape) the average
duration and the average idle gap in front of it, over the last `--tail` fraction of the run
(the steady-state decode steps).

  python tools/trace_summary.py gpurun_out/prof/.../*_kernel_trace.csv [--tail 0.5]
"""
import csv
import re
import sys
from collections import OrderedDict, defaultdict


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace("mv::", "")
    return name[:48]


def main():
    path = sys.argv[1]
    tail = 0.5
    if "--tail" in sys.argv:
        tail = float(sys.argv[sys.argv.index("--tail") + 1])
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[int(len(rows) * (1 - tail)):]
    stats = OrderedDict()
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        key = (short(r["Kernel_Name"]), r.get("Grid_Size_X", "?"), r.get("LDS_Block_Size", "?"))
        st = stats.setdefault(key, defaultdict(float))
        st["n"] += 1
        st["dur"] += e - s
        if prev_end is not None and s - prev_end < 50000:  # longer pauses are host-side, not step time
            st["gap"] += max(0, s - prev_end)
        prev_end = e
    total = sum(st["dur"] + st["gap"] for st in stats.values())
    print(f"{'kernel':50s} {'grid':>9s} {'lds':>7s} {'calls':>7s} {'avg_us':>9s} {'gap_us':>8s} {'share':>7s}")
    for (name, grid, lds), st in sorted(stats.items(), key=lambda kv: -(kv[1]["dur"] + kv[1]["gap"])):
        n = st["n"]
        print(f"{name:50s} {grid:>9s} {lds:>7s} {int(n):7d} {st['dur']/n/1e3:9.2f} {st['gap']/n/1e3:8.2f} "
              f"{(st['dur']+st['gap'])/total*100:6.1f}%")
    span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
    busy = sum(st["dur"] for st in stats.values())
    print(f"span {span/1e6:.3f} ms, kernels busy {busy/1e6:.3f} ms ({busy/span*100:.1f}%)")


if __name__ == "__main__":
    main()
