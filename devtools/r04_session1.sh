#!/bin/bash
# r04 GPU session 1: the new tests, then the launch-form probe
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sell_native.py -x -q -m gpu -k "not at_scale" 2>&1 | tail -25 > gpurun_out/r04_s1_tests_native.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sell or column_slab or spmm_vs_oracle or noise or bignn or ngcf or per_layer or simgcl" 2>&1 | tail -25 > gpurun_out/r04_s1_tests_parity.log
timeout 900 python devtools/r04_probe.py > gpurun_out/r04_s1_probe.log 2>&1
tail -5 gpurun_out/r04_s1_tests_native.log gpurun_out/r04_s1_tests_parity.log
tail -c 3000 gpurun_out/r04_s1_probe.log
