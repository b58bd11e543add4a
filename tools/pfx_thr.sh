#!/bin/bash
# Raise thresholds of the pipelined prefill attention: speed, and the margins of the full-size
# per-layer test, per build variant (lib/libmvllm_NAME.so; "base" = the library as built)
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=$PWD/micro-vllm_amd/lib/libmvllm_$v.so; [ "$v" = base ] && lib=$PWD/micro-vllm_amd/lib/libmvllm.so
  echo "== $v"
  MVLLM_LIB=$lib timeout -k 10 120 python tools/attn_prefill_only.py 4096 1 2>&1 | grep -v amdgpu.ids
  MVLLM_LIB=$lib timeout -k 10 300 python -u -m pytest tests/test_gpu_full_size.py -x -q --timeout 300 --timeout-method thread -k prefill 2>&1 | grep -E "passed|failed|worst ratio" | cut -c1-200
  MVLLM_LIB=$lib timeout -k 10 300 python tools/dbg_fullsize_prefill.py 2>&1 | grep "against fp32"
done
