#!/bin/bash
# PMC passes (separate rocprofv3 runs, kernel-trace only) over variants of the slab prototype and the library's binned kernel.
# usage: devtools/slab/pmc.sh "<variants, comma separated; 'binned' = the library kernel>"
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; export TMPDIR=/tmp
VARS=${1:-binned,w16_g256,w32_g256}
OUT=$REPO/gpurun_out/slab_pmc
mkdir -p $OUT
cd /tmp
i=0
# (FETCH_SIZE and WRITE_SIZE do not fit one pass: "exceeds the capabilities of the hardware"; a failed pass aborts within
# seconds but rocprofv3 then lingers, hence the short timeouts)
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TA_BUSY_avr TCC_BUSY_avr" \
         "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  for v in ${VARS//,/ }; do
    d=$OUT/p${i}_$v
    rm -rf $d
    timeout -k 3 75 rocprofv3 --kernel-trace --pmc $c -f csv -d $d -o p -- python $REPO/devtools/slab/run.py --pmc-run --variants $v --out $OUT/run.jsonl > $d.log 2>&1 || tail -3 $d.log
  done
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json, os
out = collections.defaultdict(dict)
for f in glob.glob("gpurun_out/slab_pmc/p*_*/**/*counter_collection.csv", recursive=True):
    var = f.split("/")[2].split("_", 1)[1]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "slab_spmm_kernel" in k or "spmm_binned_kernel" in k:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        v = v[len(v) // 3:]
        out[var][c] = sum(v) / len(v)
for var, d in out.items():
    if "FETCH_SIZE" in d:
        d["traffic_MB"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / 1e6
    if "TCC_HIT_sum" in d:
        d["l2_hit"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    print(var, json.dumps({k: round(v, 3) for k, v in sorted(d.items())}))
json.dump(out, open("gpurun_out/slab_pmc/summary.json", "w"), indent=1)
PY
find $OUT -name "*.csv" -size +1M -delete
