"""HBM traffic per launch of the dominant decode kernel from two rocprofv3 --pmc passes
(FETCH_SIZE and WRITE_SIZE, each collected in a run of its own):

  python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> KEY [SOURCE [BYTES_PER_KIB]]

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 64 bytes per
128-byte request of a wide streaming read, so it is doubled.  Writes the per-launch average
(bytes) under KEY into profiles/traffic.json, which bench.py reads for roofline.traffic.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_launch(path, counter, kernel_substr):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or kernel_substr not in r["Kernel_Name"]:
            continue
        tot += float(r["Counter_Value"])
        n += 1
    return tot, n


def main():
    fetch_csv, write_csv, key = sys.argv[1:4]
    source = sys.argv[4] if len(sys.argv) > 4 else None
    # bytes really fetched per KiB the counter reports, from a calibration run in this kernel's own
    # access pattern (tools/micro/kv_stream calib); default: the wide-stream rule, 2048
    per_kib = float(sys.argv[5]) if len(sys.argv) > 5 else None
    kernel = key.split(":")[0]
    f, nf = per_launch(fetch_csv, "FETCH_SIZE", kernel)
    w, nw = per_launch(write_csv, "WRITE_SIZE", kernel)
    assert nf > 0 and nw > 0, "no dispatches of the kernel in the counter files"
    fetch_b = f * (per_kib if per_kib else 2048.0) / nf
    write_b = w * 1024 / nw
    out = os.path.join(ROOT, "profiles", "traffic.json")
    d = json.load(open(out)) if os.path.exists(out) else {}
    d[key] = int(fetch_b + write_b)
    d[key + ":detail"] = {"launches": nf, "fetch_bytes_per_launch": int(fetch_b),
                          "write_bytes_per_launch": int(write_b),
                          "note": "FETCH_SIZE KiB x 1024 x 2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE KiB x 1024"}
    if per_kib:
        d[key + ":detail"]["note"] = (f"FETCH_SIZE KiB x {per_kib:.1f} bytes (calibrated: a reader of a known byte count in "
                                      "this kernel's access pattern, 256-byte row segments at a 2 KiB stride through a "
                                      "shuffled page table - profiles/*_pmc_fetch_calibration.json), WRITE_SIZE KiB x 1024")
    elif "paged_attn" in kernel:
        # the counter is calibrated for wide coalesced streams only; this kernel reads 256-byte rows
        # (16 lanes x 16 B) at a 2 KiB stride, every row exactly once
        d[key + ":detail"]["note"] += ("; UNCALIBRATED for this kernel's access width (256-byte row segments): the "
                                       "figure is below the bytes the kernel must read (every cached row once: "
                                       "(ctx + 1) x 2 x D x 2 bytes per sequence and kv head), so it is kept as a "
                                       "ratio between variants only")
    if source:
        d[key + ":detail"]["source"] = source
    json.dump(d, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps({key: d[key], "detail": d[key + ":detail"]}))


if __name__ == "__main__":
    main()
