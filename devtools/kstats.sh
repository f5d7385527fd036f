#!/bin/bash
# usage: devtools/kstats.sh <tag> <python script and args...>  -> per-kernel rocprofv3 stats (top ${KTOP:-12}) for any probe
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
REPO=$PWD
tag=$1; shift
mkdir -p gpurun_out/prof/$tag/stats
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $REPO/gpurun_out/prof/$tag/stats -o k -- python $REPO/"$@" > $REPO/gpurun_out/prof/$tag/log.txt 2>&1
cd $REPO
rm -f gpurun_out/prof/$tag/stats/*/k_kernel_trace.csv gpurun_out/prof/$tag/stats/k_kernel_trace.csv
python - "$tag" "${KTOP:-12}" <<'PY'
import csv, glob, sys
for f in glob.glob(f"gpurun_out/prof/{sys.argv[1]}/stats/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:int(sys.argv[2])]:
        print(f'{r["Name"][:100]:100s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:8.1f} pct {r["Percentage"]}')
PY
