#!/bin/bash
# One GPU-box session: parity tests, bench, tuning sweep, rocprofv3 kernel stats.  Everything that should
# come back goes under gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== smoke ==" ; timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== pytest -m gpu ==" ; timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
echo "== bench ==" ; timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench.log
if [ "${1:-}" != "notune" ]; then
  echo "== tune ==" ; timeout 1500 python devtools/tune_spmm.py --quick --big 2>&1 | tail -60 > gpurun_out/tune_tail.log
fi
echo "== rocprof ==" 
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof" -o bench -- python "$OLDPWD/bench.py" --steps 100 --warmup 10 --cpu-seconds 0 --no-extras > "$OLDPWD/gpurun_out/rocprof_bench.log" 2>&1 )
find gpurun_out/prof -name "*stats*" | head
