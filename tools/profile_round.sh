#!/bin/bash
# One round's profiles on the GPU box (run through gpurun from the repository root):
#   bash tools/profile_round.sh r3
# Writes under gpurun_out/prof_<tag>/:
#   trace_<leg>/   rocprofv3 --kernel-trace --stats of ONE leg of the bench line each (bs1, bs32-even,
#                  bs32-ragged, prefill, engine), so that every figure the line quotes can be recomputed
#                  from the kernel statistics of the run that produced it;
#   pmc_*          HBM traffic counters (FETCH_SIZE, WRITE_SIZE: one rocprofv3 --pmc pass each) of short
#                  bs=1 and bs=32 runs, and of tools/micro/kv_stream (a reader of a KNOWN byte count in
#                  the attention kernel's access pattern: the calibration of FETCH_SIZE for 256-byte rows);
#   pmc_mfma_*     matrix-pipe counters of the prefill kernels, next to the un-profiled time of the same
#                  binary (the counters slow a kernel down).
# Copy the summaries to profiles/ afterwards: python tools/collect_profiles.py <tag>.
set -u
TAG=${1:-r3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
QUIET="--no-cpu-baseline"
run() { echo "== $*"; "$@" || { echo "FAILED: $*"; exit 1; }; }
leg() {   # leg NAME bench-arguments...
  local name=$1; shift
  run timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -o "$TAG" -- python3 "$ROOT/bench.py" "$@" > "$OUT/trace_$name.json" 2> "$OUT/trace_$name.err"
}
leg bs1 --steps 50 --warmup 10 --no-bs32 --no-prefill --no-engine $QUIET
leg bs32-even --bs 32 --steps 30 --warmup 5 --no-prefill --no-engine $QUIET
leg bs32-ragged --bs 32 --ragged --steps 30 --warmup 5 --no-prefill --no-engine $QUIET
leg prefill --steps 1 --warmup 0 --tail-steps 0 --no-bs32 --no-engine $QUIET
leg engine --steps 1 --warmup 0 --tail-steps 0 --no-bs32 --no-prefill $QUIET
SHORT="--steps 3 --warmup 1 --tail-steps 0 --no-cpu-baseline --no-prefill --no-bs32 --no-engine"
for BS in 1 32; do
  for C in FETCH_SIZE WRITE_SIZE; do
    run timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C}_bs$BS" -o p -- python3 "$ROOT/bench.py" --bs $BS $SHORT > "$OUT/pmc_${C}_bs$BS.json" 2> "$OUT/pmc_${C}_bs$BS.err"
  done
done
# calibration of FETCH_SIZE for the attention's rows: a known byte count in the same pattern
if [ -x "$ROOT/tools/micro/kv_stream" ]; then
  run timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_calib_bs1" -o p -- "$ROOT/tools/micro/kv_stream" 1 4096 calib 32 > "$OUT/pmc_calib_bs1.log" 2> "$OUT/pmc_calib_bs1.err"
  run timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_calib_bs32" -o p -- "$ROOT/tools/micro/kv_stream" 32 4096 calib 1 > "$OUT/pmc_calib_bs32.log" 2> "$OUT/pmc_calib_bs32.err"
fi
# matrix-pipe counters of the prefill kernels: first the same tool without the profiler (the time the
# kernel takes when nothing is counted), then one counter pass
CNT="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY"
run timeout -k 10 200 python3 "$ROOT/tools/bench_gemm_prefill.py" 4096 > "$OUT/plain_gemm.log" 2> "$OUT/plain_gemm.err"
run timeout -s KILL 300 rocprofv3 --pmc $CNT --output-format csv -d "$OUT/pmc_mfma_gemm" -o p -- python3 "$ROOT/tools/bench_gemm_prefill.py" 4096 > "$OUT/pmc_mfma_gemm.log" 2> "$OUT/pmc_mfma_gemm.err"
run timeout -k 10 200 python3 "$ROOT/tools/attn_prefill_only.py" 4096 1 > "$OUT/plain_attn.log" 2> "$OUT/plain_attn.err"
run timeout -s KILL 300 rocprofv3 --pmc $CNT --output-format csv -d "$OUT/pmc_mfma_attn" -o p -- python3 "$ROOT/tools/attn_prefill_only.py" 4096 1 > "$OUT/pmc_mfma_attn.log" 2> "$OUT/pmc_mfma_attn.err"
python3 "$ROOT/tools/pmc_summary.py" gemm_prefill_wide $(find "$OUT/pmc_mfma_gemm" -name "*counter_collection.csv") > "$OUT/pmc_mfma_gemm_summary.txt"
python3 "$ROOT/tools/pmc_summary.py" paged_attn_prefill $(find "$OUT/pmc_mfma_attn" -name "*counter_collection.csv") > "$OUT/pmc_mfma_attn_summary.txt"
ls "$OUT"
