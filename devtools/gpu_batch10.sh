#!/bin/bash
# counters for the LDS-DMA BiGNN dense kernel (separate --pmc passes, kernel-trace only)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
cd /tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT" \
           "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $REPO/gpurun_out/prof/bignn_$tag
  timeout 600 rocprofv3 --kernel-trace --pmc $set -f csv -d $REPO/gpurun_out/prof/bignn_$tag -o p -- python $REPO/devtools/bignn_probe.py > /dev/null 2>&1
done
cd $REPO
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in glob.glob("gpurun_out/prof/bignn_*/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "bignn_dense" in k:
            acc[k.split("(")[0] + "|grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {}).update({c: round(sum(x) / len(x)) for c, x in v.items()})
for k, v in out.items():
    print(k, json.dumps(v))
json.dump(out, open("gpurun_out/bignn_pmc.json", "w"), indent=1)
PY
