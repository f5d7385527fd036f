#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_gpu_fused_steps.py tests/test_driver.py -m gpu -x -q 2>&1 | tail -40) > gpurun_out/r05_s9_tests.log 2>&1
tail -8 gpurun_out/r05_s9_tests.log
(timeout 600 python devtools/epoch_probe.py SimGCL XSimGCL NCL 2>&1 | tail -30) > gpurun_out/r05_s9_epochs.log 2>&1
tail -12 gpurun_out/r05_s9_epochs.log
