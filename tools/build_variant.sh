#!/bin/bash
# Build a timing-experiment variant of the library: tools/build_variant.sh NAME FILE.hip [-DEXP_... ...]
# FILE.hip is a file in diffalign_amd/csrc, or SRC=/abs/path/to/modified_copy.hip replaces it for this build.
# -> diffalign_amd/libdiffalign_hip_NAME.so (use with DA_LIB=...).  Only FILE.hip is recompiled with the extra flags.
# Compare variants only inside ONE gpurun call (tools/run_variants.sh): boxes differ by a few per cent.
set -e
cd "$(dirname "$0")/../diffalign_amd"
NAME=$1; FILE=$2; shift; shift
BASE=${FILE%.hip}
mkdir -p csrc/_var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Icsrc "$@" -c ${SRC:-csrc/$FILE} -o csrc/_var/${BASE}_$NAME.o
OBJS=$(ls csrc/_obj/*.o | grep -v "/$BASE.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libdiffalign_hip_$NAME.so $OBJS csrc/_var/${BASE}_$NAME.o
echo built libdiffalign_hip_$NAME.so
