set -u
export TMPDIR=/tmp
ROOT=$(pwd)
A="--bs 32 --steps 20 --warmup 5 --tail-steps 0 --no-cpu-baseline --no-prefill --no-bs32 --no-engine"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_slab_on -o t -- python3 $ROOT/bench.py $A > $ROOT/gpurun_out/prof_slab_on.json 2> $ROOT/gpurun_out/prof_slab_on.err || exit 1
export MVLLM_QKV_SLABS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_slab_off -o t -- python3 $ROOT/bench.py $A > $ROOT/gpurun_out/prof_slab_off.json 2> $ROOT/gpurun_out/prof_slab_off.err || exit 1
