#!/bin/bash
# r04 GPU session 3: targeted tests, N = 2 bench line, rocprofv3 stats + PMC passes (-> profiles/traffic.json keys of the r04 kernel names)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/s3
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_sell_native.py -x -q -m gpu -k "column or native or launch_forms or reasons or raw_ctypes" 2>&1 | tail -6 > gpurun_out/s3/tests.log
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size" 2>&1 | tail -4 >> gpurun_out/s3/tests.log
cat gpurun_out/s3/tests.log
timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/s3/bench_n2.json 2> gpurun_out/s3/bench_n2.err
echo "n2 rc=$?"; python - <<'PY'
import json
try:
    r = json.load(open("gpurun_out/s3/bench_n2.json"))
    print({k: r.get(k) for k in ("value", "n_gpus", "measured", "ms_per_step")}, r.get("column_sharding"))
except Exception as e:
    print("n2 parse error", e)
PY
bash devtools/profile_session.sh > gpurun_out/s3/profile_session.log 2>&1
tail -12 gpurun_out/s3/profile_session.log
bash devtools/traffic_session.sh > gpurun_out/s3/traffic_session.log 2>&1
tail -8 gpurun_out/s3/traffic_session.log
cp profiles/traffic.json gpurun_out/s3/traffic.json
find gpurun_out/prof gpurun_out/traffic -name "*.csv" -size +2M -delete
