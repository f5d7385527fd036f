#!/bin/bash
# r05 session 13: what-if breakdowns of the scoring and top-k kernels + per-kernel durations of the top-k pipeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python devtools/r05_score_whatif.py 2>&1 | tail -12
timeout 300 python devtools/r05_topk_whatif.py 2>&1 | tail -12
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_topk -o topk -- python "$GRAFT_REPO_ROOT/devtools/r05_topk_whatif.py" prof >/dev/null 2>&1)
f=$(find /tmp/prof_topk -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r05_topk_kernel_stats.csv && head -8 "$f" | cut -c1-200
