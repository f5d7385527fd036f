#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 900 python devtools/r05_colsplit_probe.py 2>&1 | tail -20) > gpurun_out/r05_s12_colsplit.log 2>&1
cat gpurun_out/r05_s12_colsplit.log | cut -c1-420
