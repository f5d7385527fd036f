#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD; mkdir -p gpurun_out/prof; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bignn or ngcf or score" 2>&1 | tail -3
python devtools/dense_probe.py 2>&1 | grep kind
cd /tmp
rm -rf $REPO/gpurun_out/prof/dense_*
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_WR -f csv -d $REPO/gpurun_out/prof/dense_sq -o p -- python $REPO/devtools/dense_probe.py > $REPO/gpurun_out/prof/dense_sq.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -f csv -d $REPO/gpurun_out/prof/dense_sq2 -o p -- python $REPO/devtools/dense_probe.py > $REPO/gpurun_out/prof/dense_sq2.log 2>&1
cd $REPO
python - <<'PY'
import csv, glob, collections
for d in ("dense_sq", "dense_sq2"):
    for f in glob.glob(f"gpurun_out/prof/{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "score_kernel" in k or "bignn_dense" in k or "Cijk" in k:
                acc[k[:60] + "|grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()}, "n=", len(next(iter(v.values()))))
PY
tail -3 gpurun_out/prof/dense_sq2.log
