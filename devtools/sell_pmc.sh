#!/bin/bash
# What bounds the library's column-slab kernel at the end of r03: PMC passes (separate rocprofv3 runs, kernel-trace only)
# over the bench command, summarised per kernel instantiation -> gpurun_out/sell_pmc/summary.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; export TMPDIR=/tmp
OUT=$REPO/gpurun_out/sell_pmc
mkdir -p $OUT
cd /tmp
BENCH="python $REPO/bench.py --steps 60 --warmup 10 --cpu-seconds 0 --no-extras --eager"
i=0
for c in "TA_BUSY_avr TCC_BUSY_avr" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" \
         "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "GRBM_GUI_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES"; do
  i=$((i+1)); d=$OUT/p$i
  rm -rf $d
  timeout -k 3 90 rocprofv3 --kernel-trace --pmc $c -f csv -d $d -o p -- $BENCH > $d.log 2>&1 || tail -3 $d.log
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/sell_pmc/p*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "sell_spmm_kernel" in k:
            name = k[k.index("sell_spmm_kernel"):].split("(")[0]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, d in acc.items():
        for c, v in d.items():
            v = v[len(v) // 4:]
            out[name][c] = sum(v) / len(v)
json.dump(out, open("gpurun_out/sell_pmc/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -size +1M -delete
