#!/bin/bash
# librbgnn_scoretrace.so = the product's objects with score.o replaced by the RBG_SCORE_TRACE build of the same source
set -e
cd "$(dirname "$0")"
C=../../recbole-gnn_amd/csrc
make -C $C -j10 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I../../include -I$C -c score_trace.hip -o /tmp/score_trace.o
OBJ=$(ls $C/_obj/*.o | grep -v "/score.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ /tmp/score_trace.o -o librbgnn_scoretrace.so -lpthread -ldl
ls -la librbgnn_scoretrace.so
