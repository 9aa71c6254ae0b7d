set -e
export EDIGPU_DIST_BACKEND=gloo
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "transposed" > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -3 gpurun_out/gpu_tests.log
for ex in transpose allgather; do
  for n in 2 4; do
    EDIGPU_EXCHANGE=$ex timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $n --steps 20 --warmup 3 > gpurun_out/multi_${ex}_$n.json 2> gpurun_out/multi_${ex}_$n.err || { tail -20 gpurun_out/multi_${ex}_$n.err; exit 1; }
    cat gpurun_out/multi_${ex}_$n.json
  done
done
