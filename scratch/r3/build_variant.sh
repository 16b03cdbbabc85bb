#!/bin/bash
# A/B partner of the product library for ONE kernel source: scratch/lib/libmvsnerf_hip_<name>.so = the product objects with <file>.hip
# recompiled with extra flags.    bash scratch/r3/build_variant.sh <name> <file.hip> "<flags>"
# Select it with MVS_LIB=scratch/lib/libmvsnerf_hip_<name>.so in the scripts that honour it (scratch/r3/h3_ab.py, train_prof.py).
set -e
name=$1; src=$2; flags=$3
cd "$(dirname "$0")/../../mvsnerf_amd/csrc"
make -s -j8 > /dev/null
mkdir -p build/var ../../scratch/lib
obj=build/var/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed $flags -c $src -o $obj
others=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -o ../../scratch/lib/libmvsnerf_hip_$name.so
echo built scratch/lib/libmvsnerf_hip_$name.so
