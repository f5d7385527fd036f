#!/bin/bash
# r05 session 14: store twins (second set) + per-kernel durations of the top-k pipeline
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python devtools/microbench/run15.py 2>&1 | tail -80
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_topk -o topk -- python "$GRAFT_REPO_ROOT/devtools/r05_topk_whatif.py" prof > /tmp/prof.log 2>&1; tail -3 /tmp/prof.log
f=$(find /tmp/prof_topk -name "*kernel_stats.csv" | head -1); echo "stats: $f"; [ -n "$f" ] && cp "$f" "$GRAFT_REPO_ROOT/gpurun_out/r05_topk_kernel_stats.csv" && head -8 "$f" | cut -c1-200
