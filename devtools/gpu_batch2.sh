#!/bin/bash
# round-2 batch: new GPU tests (configs, node dropout, views, sweep), then binned-kernel option PMC at d=64
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py tests/test_gpu_sharded.py tests/test_driver.py -m gpu -x -q --durations=12 > gpurun_out/tests_batch2.log 2>&1
tail -25 gpurun_out/tests_batch2.log
bash devtools/sweep_session.sh gowalla 64 none
