#!/bin/bash
# end-of-round evidence: the whole GPU suite, smoke, the bench line at the driver's flags and at the defaults, `--gpus 2`
# started the way the driver starts it, rocprofv3 kernel stats + PMC of the bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/final
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/final/tests.log 2>&1
tail -14 gpurun_out/final/tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -3 gpurun_out/final/smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_n1_driver_flags.json 2> gpurun_out/final/bench_n1_driver_flags.err
echo "driver-flags rc=$?"; head -c 700 gpurun_out/final/bench_n1_driver_flags.json; echo
timeout 600 python bench.py --no-extras > gpurun_out/final/bench_n1_default.json 2> gpurun_out/final/bench_n1_default.err
echo "default rc=$?"; head -c 500 gpurun_out/final/bench_n1_default.json; echo
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/final/bench_n2.json 2> gpurun_out/final/bench_n2.err
echo "n2 rc=$?"; head -c 500 gpurun_out/final/bench_n2.json; echo
bash devtools/profile_session.sh > gpurun_out/final/profile_session.log 2>&1
tail -6 gpurun_out/final/profile_session.log
find gpurun_out/prof -name "*.csv" -size +2M -delete
