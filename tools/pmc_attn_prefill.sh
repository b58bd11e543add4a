#!/bin/bash
# L2 hit / miss and fetch size of the prefill attention launch alone (4096 tokens, 8B head shape).
# Each counter group in a run of its own.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/pfpmc && cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum"; do
  name=$(echo $grp | tr ' ' '_')
  timeout -s KILL 120 rocprofv3 --pmc $grp --kernel-trace -d $R/gpurun_out/pfpmc/$name -o out --output-format csv -- python3 $R/tools/attn_prefill_only.py 4096 1 > $R/gpurun_out/pfpmc/$name.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pfpmc/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if "paged_attn_prefill" in r["Kernel_Name"]:
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        print(f"{k}: {v / n:.4g} per launch over {n} launches")
PY
