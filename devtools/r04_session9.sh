#!/bin/bash
# r04 session 9: native table Adam in the fused steps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_fused_steps.py -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s9_tests.log
cat gpurun_out/s9_tests.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or Fused or graphed or train" 2>&1 | tail -3
timeout 300 python devtools/sgl_graphed_only.py fused_graphed 2>&1 | tail -1
timeout 300 python devtools/ngcf_step.py ngcf fused 2>&1 | tail -1
MSG_DROPOUT=0.1 timeout 300 python devtools/ngcf_step.py ngcf fused 2>&1 | tail -1
