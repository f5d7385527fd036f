#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "topk or nearest or kmeans or ncl or NCL" 2>&1 | tail -6) > gpurun_out/r05_s11_tests.log 2>&1
tail -3 gpurun_out/r05_s11_tests.log
(timeout 300 python devtools/topk_probe.py short 2>&1 | tail -5) > gpurun_out/r05_s11_topk.log 2>&1
cat gpurun_out/r05_s11_topk.log
