#!/bin/bash
# r04 session 5: fused NGCF step: tests, timing, per-kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ngcf_fused.py -m gpu -x -q 2>&1 | tail -25 > gpurun_out/s5_tests.log
cat gpurun_out/s5_tests.log
timeout 300 python devtools/ngcf_step.py ngcf fused 2>&1 | tail -3
MSG_DROPOUT=0.1 timeout 300 python devtools/ngcf_step.py ngcf fused 2>&1 | tail -3
bash devtools/kstats.sh ngcf_fused devtools/ngcf_step.py ngcf fused > gpurun_out/s5_ngcf_kstats.txt 2>&1
cat gpurun_out/s5_ngcf_kstats.txt
