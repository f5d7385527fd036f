#!/bin/bash
# r04 session 4: the no-twin plain layer (tests incl. config #5) + per-kernel stats of the NGCF training step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_sell_native.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s4_tests.log
cat gpurun_out/s4_tests.log
bash devtools/kstats.sh ngcf_step devtools/ngcf_step.py ngcf graph > gpurun_out/s4_ngcf_kstats.txt 2>&1
cat gpurun_out/s4_ngcf_kstats.txt
tail -2 gpurun_out/prof/ngcf_step/log.txt
