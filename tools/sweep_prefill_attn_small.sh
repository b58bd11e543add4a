cd /root/repo
for sb in "64 1" "128 1" "256 1" "512 1" "1024 1" "256 4" "512 2" "512 4" "128 16"; do
  set -- $sb
  for w in 4 8; do
    printf "waves=%s " $w
    MVLLM_PREFILL_ATTN_WAVES=$w timeout -k 10 120 python tools/attn_prefill_only.py $1 $2 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
