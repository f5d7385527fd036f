#!/usr/bin/env python
"""Condense the rocprofv3 CSVs of devtools/profile_session.sh into small files fit for profiles/."""
import csv, glob, json, os, sys, collections

root = sys.argv[1]
out = {}
# kernel stats
for f in glob.glob(os.path.join(root, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    out["kernel_stats"] = rows
    print("kernel stats:")
    for r in rows[:8]:
        print("  ", {k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")})
# PMC: per-dispatch counter values, averaged over the timed dispatches of the spmm kernel
pmc = {}
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "spmm_binned" in r.get("Kernel_Name", "") or "sell_spmm" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        v = v[len(v) // 4:]  # skip warm-up dispatches
        pmc[k] = {"mean": sum(v) / len(v), "n": len(v), "min": min(v), "max": max(v)}
out["pmc_spmm_per_launch"] = pmc
print("pmc:", json.dumps(pmc, indent=1))
if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
    # MI355X_MICROARCH.md §HBM: counters are in KiB-ish units of 1024 B; on gfx950 FETCH_SIZE tallies 128-B requests
    # at 64 B, so the read side is doubled before comparing with a byte count.
    traffic = (2 * pmc["FETCH_SIZE"]["mean"] + pmc["WRITE_SIZE"]["mean"]) * 1024
    out["traffic_bytes_per_launch"] = traffic
    print("traffic bytes per spmm launch (2*FETCH + WRITE)*1024 =", traffic)
    # the table bench.py reads: workload : width : kernel -> bytes per launch, averaged over the launches of the bench
    # command (two plain layers + the layer that carries the fused mean)
    tpath = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    try:
        table = json.load(open(tpath))
    except Exception:
        table = {}
    key = sys.argv[2] if len(sys.argv) > 2 else "gowalla:d64:sell_spmm_kernel<32, 2, true>"
    rec = {"traffic": traffic, "fetch_size_kib": pmc["FETCH_SIZE"]["mean"], "write_size_kib": pmc["WRITE_SIZE"]["mean"]}
    if "TCC_HIT_sum" in pmc and "TCC_MISS_sum" in pmc:
        rec["l2_hit"] = pmc["TCC_HIT_sum"]["mean"] / (pmc["TCC_HIT_sum"]["mean"] + pmc["TCC_MISS_sum"]["mean"])
    table[key] = rec
    table["_bench_command_note"] = ("the key above is refreshed by devtools/profile_session.sh from PMC passes of `python bench.py --steps 100 "
                                    "--warmup 10 --cpu-seconds 0 --no-extras --eager`: mean over all SpMM launches of that command")
    json.dump(table, open(tpath, "w"), indent=1)
json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
