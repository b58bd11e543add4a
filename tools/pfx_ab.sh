#!/bin/bash
# A/B of builds of the pipelined prefill attention in one process sequence (same box, same clocks):
# tools/pfx_ab.sh NAME...   (lib/libmvllm_NAME.so; "base" = lib/libmvllm.so), two rounds
cd "$(dirname "$0")/.."
for round in 1 2; do
  for v in "$@"; do
    lib=$PWD/micro-vllm_amd/lib/libmvllm_$v.so; [ "$v" = base ] && lib=$PWD/micro-vllm_amd/lib/libmvllm.so
    printf "%-12s " "$v"
    MVLLM_LIB=$lib timeout -k 10 120 python tools/attn_prefill_only.py 4096 1 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
