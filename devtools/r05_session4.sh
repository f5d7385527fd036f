#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 300 python devtools/r05_whatif.py gowalla 2>&1 | tail -40) > gpurun_out/r05_s4_whatif.log 2>&1
(timeout 300 python devtools/r05_whatif.py amazon-book 2>&1 | tail -40) >> gpurun_out/r05_s4_whatif.log 2>&1
tail -3 gpurun_out/r05_s4_whatif.log
