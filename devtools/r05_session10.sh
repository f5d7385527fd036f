#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_s10_bench_n1.json) 2> gpurun_out/r05_s10_bench_n1.err
echo "n1 rc=$?"; head -c 1500 gpurun_out/r05_s10_bench_n1.json; echo
(timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r05_s10_bench_n2.json) 2> gpurun_out/r05_s10_bench_n2.err
echo "n2 rc=$?"; head -c 600 gpurun_out/r05_s10_bench_n2.json; echo
(timeout 900 python bench.py --gpus 4 --steps 10 --warmup 3 --shard hybrid > gpurun_out/r05_s10_bench_n4_hybrid.json) 2> gpurun_out/r05_s10_bench_n4_hybrid.err
echo "n4 rc=$?"; head -c 600 gpurun_out/r05_s10_bench_n4_hybrid.json; echo
tail -3 gpurun_out/r05_s10_bench_n4_hybrid.err
