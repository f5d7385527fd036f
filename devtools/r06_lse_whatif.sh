#!/bin/bash
# what-if builds of lse.hip (RBG_LSE_WHATIF bit mask, see the top of the file): devtools/microbench/librbgnn_lsewi_<k>.so, built HERE (hipcc cross-compiles);
# on the GPU box: for k in ...; do RBGNN_LIB=devtools/microbench/librbgnn_lsewi_${TAG:-}$k.so python devtools/r06_lse_whatif_time.py; done
set -e
cd "$(dirname "$0")/../recbole-gnn_amd/csrc"
mkdir -p _obj_wi
for k in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DRBG_LSE_WHATIF=${k%%s*} -DRBG_LSE_STAGGER=${STAGGER:-0} -c lse.hip -o _obj_wi/lse_${TAG:-}$k.o &
done
wait
for k in "$@"; do
  objs=$(ls _obj/*.o | grep -v "_obj/lse.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs _obj_wi/lse_${TAG:-}$k.o -o ../../devtools/microbench/librbgnn_lsewi_${TAG:-}$k.so -lpthread -ldl
done
