#!/bin/bash
# librbgnn_screentrace.so = the product's objects with topk_screen.o replaced by the RBG_SCREEN_DBG build of the same source
set -e
cd "$(dirname "$0")"
C=../../recbole-gnn_amd/csrc
make -C $C -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 -I../../include -I$C -c topk_screen_trace.hip -o /tmp/topk_screen_trace.o
OBJ=$(ls $C/_obj/*.o | grep -v "/topk_screen.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ /tmp/topk_screen_trace.o -o librbgnn_screentrace.so -lpthread -ldl
ls -la librbgnn_screentrace.so
