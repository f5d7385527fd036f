#!/bin/bash
# rocprofv3 evidence for profiles/: (1) --kernel-trace --stats of the default bench command,
# (2) PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs; kernel-trace only, as the pool requires).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
BENCH="python $REPO/bench.py --steps 100 --warmup 10 --cpu-seconds 0 --no-extras --eager"
timeout -k 3 240 rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof/stats -o bench -- $BENCH > $REPO/gpurun_out/prof/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout -k 3 240 rocprofv3 --kernel-trace --pmc $c -f csv -d $REPO/gpurun_out/prof/pmc_$tag -o bench -- $BENCH > $REPO/gpurun_out/prof/pmc_$tag.log 2>&1
done
cd $REPO
find gpurun_out/prof -name "*.csv" | head -20
python devtools/summarize_profile.py gpurun_out/prof || true
