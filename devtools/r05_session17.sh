#!/bin/bash
# r05 session 17: option "deterministic" — the scatters in both modes, bit stability of the fused steps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused_steps.py -x -q -m gpu 2>&1 | tail -15
