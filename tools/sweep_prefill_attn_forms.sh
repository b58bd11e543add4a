#!/bin/bash
# Prefill attention, 8B head shape: 128-row workgroups (plain kernel) against 256-row workgroups
# (pipelined kernel) per chunk length and batch; the form is read once per process.
cd "$(dirname "$0")/.."
for sb in "256 16" "512 8" "1024 4" "1024 8" "1536 2" "2048 1" "2048 2" "3072 1" "4096 1" "6144 1"; do
  set -- $sb
  for w in 4 8; do
    printf "waves=%s " $w
    MVLLM_PREFILL_ATTN_WAVES=$w timeout -k 10 120 python tools/attn_prefill_only.py $1 $2 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
