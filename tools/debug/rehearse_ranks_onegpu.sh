# The several-ranks bench line with 2, 4 and 8 ranks sharing ONE GPU (rendezvous over gloo): checks the
# path - IPC mapping, flags, captured step, all ranks identical - not link latency.
set -u
for n in 2 4 8; do
  MVLLM_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port $((29520 + n)) bench.py --gpus $n --steps 20 --warmup 5 \
    > gpurun_out/r2_bench_tp${n}_onegpu.json 2> gpurun_out/r2_bench_tp${n}_onegpu.err || { echo "N=$n FAILED"; tail -5 gpurun_out/r2_bench_tp${n}_onegpu.err; exit 1; }
  tail -c 400 gpurun_out/r2_bench_tp${n}_onegpu.json | head -c 300; echo
done
