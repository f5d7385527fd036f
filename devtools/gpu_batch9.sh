#!/bin/bash
# LDS-DMA BiGNN dense kernel: parity, clock trace, interleaved graph-replay timing against the general kernel, NGCF step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; REPO=$PWD
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "bignn or ngcf or NGCF" > gpurun_out/tests_batch9.log 2>&1
tail -3 gpurun_out/tests_batch9.log
python devtools/microbench/run10.py > gpurun_out/bignn_trace.jsonl 2>&1
rm -f gpurun_out/bignn_probe.jsonl
timeout 600 python devtools/bignn_probe.py 2>&1 | grep -E "kind|Error|error"
timeout 600 python devtools/ngcf_step.py 2>&1 | tail -8
