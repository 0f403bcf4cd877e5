#!/bin/bash
# Build a timing-experiment variant of the library: tools/build_variant.sh NAME [-DEXP_... ...]
# -> diffalign_amd/libdiffalign_hip_NAME.so (use with DA_LIB=...).  Only k_edge_layer.hip is recompiled.
set -e
cd "$(dirname "$0")/../diffalign_amd"
NAME=$1; shift
mkdir -p csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c csrc/k_edge_layer.hip -o csrc/_obj/k_edge_layer_$NAME.o
OBJS=$(ls csrc/_obj/*.o | grep -v "k_edge_layer")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdiffalign_hip_$NAME.so $OBJS csrc/_obj/k_edge_layer_$NAME.o
echo built libdiffalign_hip_$NAME.so
