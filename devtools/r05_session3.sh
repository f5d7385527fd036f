#!/bin/bash
# r05 session 3: the longest-first schedule of the resident-round launch
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 400 python -m pytest tests/test_gpu_sell_native.py tests/test_gpu_fused_steps.py -x -q -k "resident_round or borrowed_plan or dying" 2>&1 | tail -40) > gpurun_out/r05_s3_tests.log 2>&1
(timeout 500 python devtools/r05_probe.py gowalla,yelp2018,amazon-book 64 2>&1 | tail -80) > gpurun_out/r05_s3_probe.log 2>&1
(timeout 300 python devtools/r05_probe.py g-1.3m 64 quick 2>&1 | tail -40) > gpurun_out/r05_s3_probe_big.log 2>&1
tail -5 gpurun_out/r05_s3_tests.log
