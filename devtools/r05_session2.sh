#!/bin/bash
# r05 session 2: the gather ceiling of the vector memory path (microbenchmark) + the view-lifetime and capture tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 300 devtools/microbench/gather_rate > gpurun_out/r05_gather_rate.jsonl) 2> gpurun_out/r05_s2_gather.err
(timeout 300 python -m pytest tests/test_gpu_sell_native.py tests/test_gpu_fused_steps.py -x -q -k "borrowed_plan or dying or reweighted_view" 2>&1 | tail -15) > gpurun_out/r05_s2_tests.log 2>&1
tail -5 gpurun_out/r05_s2_tests.log; wc -l gpurun_out/r05_gather_rate.jsonl; cat gpurun_out/r05_s2_gather.err | tail -3
