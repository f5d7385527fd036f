#!/bin/bash
# end-of-round evidence: the whole GPU suite, smoke, the default bench line, rocprofv3 kernel stats + PMC of the bench command
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/tests_final.log 2>&1
tail -14 gpurun_out/tests_final.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke_final.log 2>&1; tail -3 gpurun_out/smoke_final.log
timeout 600 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 1500 gpurun_out/bench_final.json
bash devtools/profile_session.sh > gpurun_out/profile_session.log 2>&1
tail -25 gpurun_out/profile_session.log
cp profiles/traffic.json gpurun_out/traffic.json
find gpurun_out/prof -name "*.csv" -size +2M -delete
