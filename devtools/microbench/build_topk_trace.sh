#!/bin/bash
# librbgnn_topktrace.so = the product's objects with topk.o replaced by the RBG_TOPK_TRACE build of the same source
set -e
cd "$(dirname "$0")"
C=../../recbole-gnn_amd/csrc
make -C $C -j10 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I../../include -I$C -c topk_trace.hip -o /tmp/topk_trace.o
OBJ=$(ls $C/_obj/*.o | grep -v "/topk.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ /tmp/topk_trace.o -o librbgnn_topktrace.so -lpthread -ldl
ls -la librbgnn_topktrace.so
