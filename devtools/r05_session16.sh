#!/bin/bash
# r05 session 16: two item tiles per iteration in the top-k main pass
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for it in 1 2 1 2; do timeout 300 python devtools/r05_topk_whatif.py time "item_tiles=$it" $it 2>&1 | tail -2; done
