#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 500 python devtools/r05_probe.py gowalla,yelp2018,amazon-book 64 2>&1 | tail -80) > gpurun_out/r05_s5_probe.log 2>&1
tail -2 gpurun_out/r05_s5_probe.log
