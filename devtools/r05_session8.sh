#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fused_steps.py tests/test_driver.py tests/test_gpu_sharded.py tests/test_gpu_configs.py -m gpu -x -q -k "not config5" 2>&1 | tail -40) > gpurun_out/r05_s8_tests.log 2>&1
tail -8 gpurun_out/r05_s8_tests.log
