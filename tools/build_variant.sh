#!/bin/bash
# usage: tools/build_variant.sh NAME "-DE3_MSG_X=1 ..." [SOURCE]  -> scalable-e3-gnn_amd/lib/exp/libe3gnn_NAME.so
# Experiment builds of one source of the library (development tool; SOURCE = e3_msg_fused (default) | e3_msg_ws | ...):
# the other objects are reused from csrc/build.
set -e
name=$1; flags=$2; src=${3:-e3_msg_fused}
cd "$(dirname "$0")/../scalable-e3-gnn_amd/csrc"
mkdir -p build_exp ../lib/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize $flags \
  -c $src.hip -o build_exp/${src}_$name.o
objs=$(ls build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/libe3gnn_$name.so $objs build_exp/${src}_$name.o
echo built ../lib/exp/libe3gnn_$name.so
