#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 500 python devtools/r05_probe.py gowalla,yelp2018,amazon-book 64 2>&1 | tail -80) > gpurun_out/r05_s6_probe.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_sell_native.py -x -q 2>&1 | tail -8) > gpurun_out/r05_s6_tests.log 2>&1
tail -3 gpurun_out/r05_s6_tests.log
