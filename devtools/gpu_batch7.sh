#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python devtools/split_probe.py > gpurun_out/split_probe.jsonl 2>&1
grep score gpurun_out/split_probe.jsonl
timeout 1200 python -m pytest tests -m gpu -q -k "score or full_sort or config1 or config3" > gpurun_out/tests_batch7.log 2>&1
tail -5 gpurun_out/tests_batch7.log
