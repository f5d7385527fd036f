#!/bin/bash
# r04 session 7: fused SGL step: tests, timing, per-kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_steps.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/s7_tests.log
cat gpurun_out/s7_tests.log
timeout 300 python devtools/sgl_graphed_only.py 2>&1 | tail -3
bash devtools/kstats.sh sgl_fused devtools/sgl_graphed_only.py fused_graphed > gpurun_out/s7_sgl_kstats.txt 2>&1
cat gpurun_out/s7_sgl_kstats.txt
