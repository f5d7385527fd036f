#!/bin/bash
# r04 session 6: per-kernel stats of the graphed SGL step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
bash devtools/kstats.sh sgl_graphed devtools/sgl_graphed_only.py > gpurun_out/s6_sgl_kstats.txt 2>&1
cat gpurun_out/s6_sgl_kstats.txt
