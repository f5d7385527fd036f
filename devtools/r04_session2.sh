#!/bin/bash
# r04 GPU session 2: the whole GPU suite + smoke + a bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/s2
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/s2/tests.log 2>&1
tail -25 gpurun_out/s2/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s2/smoke.log 2>&1; tail -3 gpurun_out/s2/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s2/bench_n1.json 2> gpurun_out/s2/bench_n1.err
echo "bench rc=$?"; head -c 1500 gpurun_out/s2/bench_n1.json; echo; tail -5 gpurun_out/s2/bench_n1.err
