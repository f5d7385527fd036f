#!/bin/bash
# Sweep-kernel evidence: timing of every variant, then PMC passes (separate runs, kernel-trace only) joined per variant.
# usage: devtools/sweep_session.sh [workload] [dim] [variants]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
WL=${1:-gowalla}; DIM=${2:-64}; VAR=${3:-}
OUT=$REPO/gpurun_out/sweep_${WL}_d${DIM}
mkdir -p $OUT
export TMPDIR=/tmp
VARG=""; [ -n "$VAR" ] && VARG="--variants $VAR"
timeout 900 python devtools/sweep_probe.py --workload $WL --dim $DIM $VARG --out $OUT/timing.jsonl > $OUT/timing.log 2>&1
tail -30 $OUT/timing.log
cd /tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $c -f csv -d $OUT/pmc_$tag -o p -- python $REPO/devtools/sweep_probe.py --workload $WL --dim $DIM $VARG --pmc-run --iters 12 --manifest $OUT/pmc_$tag/manifest.json > $OUT/pmc_$tag.log 2>&1
done
cd $REPO
python devtools/sweep_probe.py --summarize $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_TCC_HIT_sum_TCC_MISS_sum | tee $OUT/pmc_summary.txt
# keep the merge-back small: drop the raw rocprof trees, keep logs + summaries
find $OUT -name "*.csv" -size +2M -delete
