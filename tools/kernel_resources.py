"""Print VGPR/SGPR/LDS/scratch/occupancy per kernel of one csrc/*.hip file (compile only, no GPU)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    if not os.path.isabs(src):
        src = os.path.join(ROOT, "micro-vllm_amd", "csrc", src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
           "-ffp-contract=off", "-c", src, "-o", "/dev/null",
           "-Rpass-analysis=kernel-resource-usage"]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark: +([^:]+): +(.+?) \[-Rpass", line)
        if not m:
            if "error" in line:
                print(line)
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = subprocess.run(["c++filt", val], stdout=subprocess.PIPE, text=True).stdout.strip()
            cur = re.sub(r"\(.*", "", cur)
            rows[cur] = {}
        elif cur:
            rows[cur][key] = val
    for name, r in rows.items():
        print(f"{name:60s} vgpr={r.get('VGPRs','?'):>4} agpr={r.get('AGPRs','?'):>3} sgpr={r.get('TotalSGPRs','?'):>3} "
              f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4} occ={r.get('Occupancy [waves/SIMD]','?'):>2} "
              f"lds={r.get('LDS Size [bytes/block]','?')}")


if __name__ == "__main__":
    main()
