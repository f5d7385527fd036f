#!/bin/bash
# r05 session 15: the top-k kernel — parity tests of the top-k alone, timing, per-kernel durations
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_sort_topk or score_and_topk" 2>&1 | tail -4
timeout 300 python devtools/r05_topk_whatif.py time "$1" 2>&1 | tail -1
bash devtools/kstats.sh topk2 devtools/r05_topk_whatif.py prof 2>&1 | grep -i "topk"
