#!/bin/bash
# r05 session 1: the resident-round launch + the W = 16 plan: tests, timings on every quoted shape, the per-wave clock
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 400 python -m pytest tests/test_gpu_sell_native.py -x -q -k "resident_round or native_plan_equals or launch_forms" 2>&1 | tail -15) > gpurun_out/r05_s1_tests.log 2>&1
(timeout 500 python devtools/r05_probe.py gowalla,yelp2018,amazon-book 64 2>&1 | tail -80) > gpurun_out/r05_s1_probe.log 2>&1
(timeout 200 python devtools/r05_trace.py gowalla 64 2>&1 | tail -40) > gpurun_out/r05_s1_trace.log 2>&1
(timeout 300 python devtools/r05_probe.py g-1.3m 64 quick 2>&1 | tail -40) > gpurun_out/r05_s1_probe_big.log 2>&1
(timeout 200 python devtools/r05_probe.py gowalla 128 quick 2>&1 | tail -40) > gpurun_out/r05_s1_probe_d128.log 2>&1
tail -5 gpurun_out/r05_s1_tests.log
grep -c bit_identical gpurun_out/r05_probe.jsonl
