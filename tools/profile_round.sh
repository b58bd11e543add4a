#!/bin/bash
# One round's profiles on the GPU box (run through gpurun from the repository root):
#   bash tools/profile_round.sh r2
# Writes under gpurun_out/prof_<tag>/: a kernel trace with statistics of the default bench run, and
# HBM traffic counters (FETCH_SIZE, WRITE_SIZE: one rocprofv3 --pmc pass each) of short bs=1 and
# bs=32 runs.  Copy the summaries to profiles/ afterwards (tools/pmc_traffic.py, tools/trace_summary.py).
set -u
TAG=${1:-r2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SHORT="--steps 3 --warmup 1 --tail-steps 0 --no-cpu-baseline --no-prefill --no-bs32 --no-engine"
run() { echo "== $*"; "$@" || { echo "FAILED: $*"; exit 1; }; }
run timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o "$TAG" -- python3 "$ROOT/bench.py" --steps 50 --warmup 10 --no-cpu-baseline > "$OUT/trace_bench.json" 2> "$OUT/trace_bench.err"
for BS in 1 32; do
  for C in FETCH_SIZE WRITE_SIZE; do
    run timeout -s KILL 300 rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_${C}_bs$BS" -o p -- python3 "$ROOT/bench.py" --bs $BS $SHORT > "$OUT/pmc_${C}_bs$BS.json" 2> "$OUT/pmc_${C}_bs$BS.err"
  done
done
# matrix-pipe counters of the prefill kernels (one pass; SQ_VALU_MFMA_BUSY_CYCLES counts cycles,
# SQ_BUSY_CYCLES and SQ_WAVE_CYCLES per-SE / quad-cycle units: tools/pmc_summary.py prints the sums)
run timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_mfma_gemm" -o p -- python3 "$ROOT/tools/bench_gemm_prefill.py" 4096 > "$OUT/pmc_mfma_gemm.log" 2> "$OUT/pmc_mfma_gemm.err"
run timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_mfma_attn" -o p -- python3 "$ROOT/tools/attn_prefill_only.py" 4096 1 > "$OUT/pmc_mfma_attn.log" 2> "$OUT/pmc_mfma_attn.err"
python3 "$ROOT/tools/pmc_summary.py" gemm_prefill_wide $(find "$OUT/pmc_mfma_gemm" -name "*counter_collection.csv") > "$OUT/pmc_mfma_gemm_summary.txt"
python3 "$ROOT/tools/pmc_summary.py" paged_attn_prefill $(find "$OUT/pmc_mfma_attn" -name "*counter_collection.csv") > "$OUT/pmc_mfma_attn_summary.txt"
ls -R "$OUT" | head -60
