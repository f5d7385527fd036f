#!/bin/bash
# r04 session 8: one-pass InfoNCE: tests + SGL step timing + nce stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_steps.py -m gpu -x -q -k "infonce or sgl" 2>&1 | tail -15 > gpurun_out/s8_tests.log
cat gpurun_out/s8_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "nce or lse or sgl or SGL" 2>&1 | tail -5
timeout 300 python devtools/sgl_graphed_only.py 2>&1 | tail -2
timeout 300 python devtools/nce_stats.py 2>&1 | tail -6
