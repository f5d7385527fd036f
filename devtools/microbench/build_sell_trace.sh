#!/bin/bash
# librbgnn_selltrace.so = the product's objects with sell.o replaced by the RBG_SELL_TRACE build of the same source
set -e
cd "$(dirname "$0")"
C=../../recbole-gnn_amd/csrc
make -C $C -j10 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I../../include -I$C -c sell_trace.hip -o /tmp/sell_trace.o
OBJ=$(ls $C/_obj/*.o | grep -v "/sell.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ /tmp/sell_trace.o -o librbgnn_selltrace.so -lpthread -ldl
ls -la librbgnn_selltrace.so
