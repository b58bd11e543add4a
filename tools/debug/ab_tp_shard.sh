# One rank's shard of the 8B shape at 16 rows, ctx 2048: step time with and without slab-fed attention
set -u
A="--bs 16 --ctx 2048 --steps 100 --warmup 10 --tail-steps 0 --no-cpu-baseline --no-prefill --no-bs32 --no-engine"
for model in llama3-8b-tp8shard llama3-8b-tp4shard llama3-8b-tp2shard llama3-8b; do
  for m in 1 0 1 0; do
    MVLLM_QKV_SLABS=$m timeout -k 10 200 python3 bench.py --model $model $A 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$model slabs=$m', d['value'], d['ms_per_step'])" || exit 1
  done
done
