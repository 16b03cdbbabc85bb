#!/bin/bash
# A/B partner of the product library for ONE kernel source: scratch/lib/libmvsnerf_hip_<name>.so = the product objects with <base>.hip replaced by
# <source file> compiled with extra flags.    bash scratch/r6/build_variant.sh <name> <base.hip> <source file> "<flags>"
# Select it with MVS_LIB=scratch/lib/libmvsnerf_hip_<name>.so in the scripts that honour it (scratch/r6/h3_ab.py, h3_census.py).
set -e
name=$1; base=$2; src=$(realpath $3); flags=$4
cd "$(dirname "$0")/../../mvsnerf_amd/csrc"
make -s -j8 > /dev/null
mkdir -p build/var ../../scratch/lib
obj=build/var/${base%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -I. $flags -c $src -o $obj
others=$(ls build/*.o | grep -v "build/${base%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -o ../../scratch/lib/libmvsnerf_hip_$name.so
echo built scratch/lib/libmvsnerf_hip_$name.so
