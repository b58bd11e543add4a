# One rank's shard of the 70B shape (TP 8) at 8 rows, ctx 8192: step time with and without slab-fed attention
set -u
A="--model llama3-70b-tp8shard --bs 8 --ctx 8192 --steps 60 --warmup 10 --tail-steps 0 --no-cpu-baseline --no-prefill --no-bs32 --no-engine"
for m in 1 0 1 0; do
  MVLLM_QKV_SLABS=$m timeout -k 10 300 python3 bench.py $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slabs=$m', d['value'], d['ms_per_step'], d['step_hbm_frac'])" || exit 1
done
