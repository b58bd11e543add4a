# A/B of the decode step at 32 rows with and without slab-fed attention (run through gpurun)
set -u
A="--bs 32 --steps 100 --warmup 10 --tail-steps 0 --no-cpu-baseline --no-prefill --no-bs32 --no-engine"
for i in 1 2 3; do
  for m in 1 0; do
    MVLLM_QKV_SLABS=$m timeout -k 10 200 python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slabs=$m', d['value'], d['ms_per_step'])" || exit 1
  done
done
