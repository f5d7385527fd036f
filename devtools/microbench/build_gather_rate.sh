#!/bin/bash
# gather_rate = the random-row-gather microbenchmark (profiles/r05_gather_rate.jsonl); the binary travels with gpurun, not with git
cd "$(dirname "$0")" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ls -la gather_rate
