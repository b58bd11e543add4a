#!/bin/bash
# Timing experiments on the pipelined prefill attention: the same launch with parts compiled out
# (wrong results, timing only).  Build the variants first (build.build_variant, see DESIGN).
set -e
cd "$(dirname "$0")/.."
for v in "" _pfx_noexp _pfx_nostage _pfx_nobar _pfx_nokread _pfx_novread _pfx_noexp_noread _pfx_mfmaonly; do
  echo "variant: libmvllm$v"
  MVLLM_LIB=$PWD/micro-vllm_amd/lib/libmvllm$v.so timeout -k 10 120 python tools/attn_prefill_only.py 4096 1
done
