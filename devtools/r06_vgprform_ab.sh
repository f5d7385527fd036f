#!/bin/bash
# A/B: the product library against the same sources built with -mllvm -amdgpu-mfma-vgpr-form=1 on every file: the bench's extras
for tag in product vgprform product vgprform; do
  if [ $tag = vgprform ]; then export RBGNN_LIB=$PWD/devtools/microbench/librbgnn_vgprform.so; else unset RBGNN_LIB; fi
  timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/r06_ab_$tag.json 2> /dev/null
  python - $tag <<'PY'
import json, sys
r = json.load(open(f"gpurun_out/r06_ab_{sys.argv[1]}.json"))
e = r["extras"]
keys = ["score_gemm_us(4096 users x all items)", "full_sort_topk_exact_passes_us", "train_step_fused_us(batch 2048, fwd + BPR + bwd + Adam)", "ngcf_forward_us", "ngcf_fused_step_graphed_us", "sgl_fused_step_graphed_us"]
out = {"lib": sys.argv[1], "prop_per_s": round(r["value"], 1)}
for k in keys: out[k.split("(")[0]] = round(e.get(k, 0), 1) if isinstance(e.get(k), float) else e.get(k)
for k, v in e.items():
    if k.startswith("driver_epoch"): out["epochs"] = {m: x.get("epoch_s") for m, x in v.items() if isinstance(x, dict)}
print(json.dumps(out))
PY
done
