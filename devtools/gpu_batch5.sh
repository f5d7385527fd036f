#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python devtools/microbench/run8.py > gpurun_out/bf16x3_probe.jsonl 2>&1
cat gpurun_out/bf16x3_probe.jsonl
timeout 1200 python -m pytest tests -m gpu -q -k "kmeans or fused_training_step or device_planner or ncl" > gpurun_out/tests_batch5.log 2>&1
tail -8 gpurun_out/tests_batch5.log
