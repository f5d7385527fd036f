#!/bin/bash
# usage: devtools/pmc.sh <tag> <kernel-name-substring> <python script and args...>  -> SQ counters per kernel (mean per dispatch)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; export TMPDIR=/tmp
tag=$1; pat=$2; shift; shift
mkdir -p gpurun_out/prof/$tag
cd /tmp
rm -rf $REPO/gpurun_out/prof/$tag/sq*
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS -f csv -d $REPO/gpurun_out/prof/$tag/sq1 -o p -- python $REPO/"$@" > $REPO/gpurun_out/prof/$tag/sq1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d $REPO/gpurun_out/prof/$tag/sq2 -o p -- python $REPO/"$@" > $REPO/gpurun_out/prof/$tag/sq2.log 2>&1
cd $REPO
python - "$tag" "$pat" <<'PY'
import csv, glob, collections, sys
tag, pat = sys.argv[1], sys.argv[2]
for d in ("sq1", "sq2"):
    for f in glob.glob(f"gpurun_out/prof/{tag}/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat in k:
                acc[k[:60] + "|grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n=", len(next(iter(v.values()))))
PY
tail -2 gpurun_out/prof/$tag/sq2.log
