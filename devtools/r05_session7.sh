#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/r05_s7_gpu_tests.log 2>&1
tail -6 gpurun_out/r05_s7_gpu_tests.log
