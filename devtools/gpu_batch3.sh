#!/bin/bash
# round-2 batch 3: all GPU tests but the 4-minute config-#5 one, the default bench line, the N = 2 bench on one GPU (staged)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "not config5" --durations=8 > gpurun_out/tests_batch3.log 2>&1
tail -22 gpurun_out/tests_batch3.log
timeout 600 python bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err
tail -c 3000 gpurun_out/bench_n1.json; tail -3 gpurun_out/bench_n1.err
for mode in weak strong; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --transport staged --scaling $mode > gpurun_out/bench_n2_$mode.json 2> gpurun_out/bench_n2_$mode.err
  tail -c 2500 gpurun_out/bench_n2_$mode.json; tail -3 gpurun_out/bench_n2_$mode.err
done
