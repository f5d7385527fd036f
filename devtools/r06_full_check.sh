#!/bin/bash
# full GPU suite + the bench at the driver's flags -> gpurun_out/r06_gpu_tests_<tag>.log, r06_bench_<tag>.json
tag=${1:-screen}
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r06_gpu_tests_$tag.log
cat gpurun_out/r06_gpu_tests_$tag.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_$tag.json 2> gpurun_out/r06_bench_$tag.err
tail -3 gpurun_out/r06_bench_$tag.err
python - "$tag" <<'PY'
import json, sys
r = json.load(open(f"gpurun_out/r06_bench_{sys.argv[1]}.json"))
print(r["value"], r["ms_per_step"], r["roofline"]["frac"])
for k, v in r["extras"].items():
    if "topk" in k: print(k, v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "note"})
for k, v in r["extras"].items():
    if k.startswith("driver_epoch"):
        print({m: e.get("epoch_s") for m, e in v.items() if isinstance(e, dict)})
PY
