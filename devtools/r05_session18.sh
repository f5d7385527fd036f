#!/bin/bash
# r05 session 18: the wave index as a scalar in the NGCF, training-step, InfoNCE and binned kernels — tests and the figures they move
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_steps.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python devtools/r05_det_cost.py 2>&1 | tail -5
timeout 300 python devtools/ngcf_step.py ngcf 2>&1 | tail -4
