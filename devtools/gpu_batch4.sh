#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -k "not config5" --durations=8 > gpurun_out/tests_batch4.log 2>&1
tail -25 gpurun_out/tests_batch4.log
for mode in weak strong; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --transport staged --scaling $mode > gpurun_out/bench_n2_$mode.json 2> gpurun_out/bench_n2_$mode.err
  tail -c 1500 gpurun_out/bench_n2_$mode.json; tail -3 gpurun_out/bench_n2_$mode.err | cut -c1-300
done
