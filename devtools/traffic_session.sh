#!/bin/bash
# Fabric traffic and L2 hit rate of the SpMM layer for every shape a roofline figure is quoted for -> profiles/traffic.json.
# One rocprofv3 run per counter set (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only, short timeouts
# (a pass that cannot be configured aborts within seconds but rocprofv3 then lingers).
# usage: devtools/traffic_session.sh ["workload:dim ..."]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; export TMPDIR=/tmp
SHAPES=${1:-"gowalla:128 yelp2018:64 amazon-book:64 g-1.3m:64"}  # (gowalla:64 = the bench command itself: devtools/profile_session.sh)
OUT=$REPO/gpurun_out/traffic
mkdir -p $OUT
cd /tmp
for sh in $SHAPES; do
  wl=${sh%%:*}; dim=${sh##*:}
  i=0
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1)); d=$OUT/${wl}_d${dim}_p$i
    rm -rf $d
    timeout -k 3 150 rocprofv3 --kernel-trace --pmc $c -f csv -d $d -o p -- python $REPO/devtools/traffic_probe.py --workload $wl --dim $dim > $d.log 2>&1 || tail -3 $d.log
  done
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json, os, re
table_path = "profiles/traffic.json"
try:
    table = json.load(open(table_path))
except Exception:
    table = {}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
names = {}
for f in glob.glob("gpurun_out/traffic/*_p*/**/*counter_collection.csv", recursive=True):
    m = re.match(r"(.+)_d(\d+)_p\d+$", f.split("/")[2])
    wl, dim = m.group(1), int(m.group(2))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "spmm_binned_kernel" in k or "spmm_sweep_kernel" in k or "spmm_generic_kernel" in k or "sell_spmm_kernel" in k:
            acc[(wl, dim)][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("gpurun_out/traffic/*_p1.log"):
    for line in open(f):
        if line.startswith("{"):
            j = json.loads(line); names[(j["workload"], j["dim"])] = j["kernel"]
for (wl, dim), d in sorted(acc.items()):
    m = {c: sum(v[len(v) // 3:]) / len(v[len(v) // 3:]) for c, v in d.items()}
    if (wl, dim) not in names or "FETCH_SIZE" not in m or "WRITE_SIZE" not in m:
        print("incomplete:", wl, dim, m); continue
    rec = {"traffic": (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024, "fetch_size_kib": m["FETCH_SIZE"], "write_size_kib": m["WRITE_SIZE"]}
    if "TCC_HIT_sum" in m:
        rec["l2_hit"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    table[f"{wl}:d{dim}:{names[(wl, dim)]}"] = rec
    print(wl, dim, names[(wl, dim)], {k: round(v, 4) for k, v in rec.items()})
table["_note"] = ("bytes per launch of the plain SpMM layer: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (MI355X guide, gfx950: FETCH_SIZE tallies "
                  "128-B requests at 64 B), l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS); separate rocprofv3 --pmc passes of "
                  "devtools/traffic_probe.py, mean over the launches after the first third (devtools/traffic_session.sh)")
table.pop("_bench_command_note", None)
json.dump(table, open(table_path, "w"), indent=1)
PY
find $OUT -name "*.csv" -size +1M -delete
cp profiles/traffic.json $OUT/traffic.json  # (gpurun merges gpurun_out/ back, not profiles/)
