#!/bin/bash
# end-of-round evidence: the whole GPU suite, smoke, rocprofv3 kernel stats + PMC of the bench command and of the other
# shapes' plain layer (-> profiles/traffic.json, which the bench lines then quote), the bench line at the driver's flags and
# at the defaults, `--gpus 2` started the way the driver starts it
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/final
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/final/tests.log 2>&1
tail -14 gpurun_out/final/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -3 gpurun_out/final/smoke.log
bash devtools/profile_session.sh > gpurun_out/final/profile_session.log 2>&1
tail -6 gpurun_out/final/profile_session.log
bash devtools/traffic_session.sh > gpurun_out/final/traffic_session.log 2>&1
tail -6 gpurun_out/final/traffic_session.log
cp profiles/traffic.json gpurun_out/final/traffic.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_n1_driver_flags.json 2> gpurun_out/final/bench_n1_driver_flags.err
echo "driver-flags rc=$?"; head -c 700 gpurun_out/final/bench_n1_driver_flags.json; echo
timeout 600 python bench.py --no-extras > gpurun_out/final/bench_n1_default.json 2> gpurun_out/final/bench_n1_default.err
echo "default rc=$?"; head -c 500 gpurun_out/final/bench_n1_default.json; echo
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/final/bench_n2.json 2> gpurun_out/final/bench_n2.err
echo "n2 rc=$?"; head -c 500 gpurun_out/final/bench_n2.json; echo
timeout 900 python bench.py --gpus 4 --steps 10 --warmup 3 --shard hybrid > gpurun_out/final/bench_n4_hybrid.json 2> gpurun_out/final/bench_n4_hybrid.err
echo "n4 hybrid rc=$?"; head -c 400 gpurun_out/final/bench_n4_hybrid.json; echo
timeout 300 python devtools/r06_lse_f16_probe.py > gpurun_out/final/lse_f16_probe.jsonl 2> gpurun_out/final/lse_f16_probe.err
tail -2 gpurun_out/final/lse_f16_probe.jsonl | cut -c1-300
find gpurun_out/prof gpurun_out/traffic -name "*.csv" -size +2M -delete
