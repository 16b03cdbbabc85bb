#!/bin/bash
# scratch/r5/build_variant.sh <name> <extra flags...>: every code object with -O3 + the given flags -> scratch/lib/libmvsnerf_hip_<name>.so
# (A/B on one box: copy it over mvsnerf_amd/lib/libmvsnerf_hip.so of the box's scratch copy - never in the tree).  "r4flags" = plain -O3, the plane sweep without SLP.
set -e
name=$1; shift
cd "$(dirname "$0")/../../mvsnerf_amd/csrc"
mkdir -p build_$name ../../scratch/lib
for f in *.hip; do
  extra="$@"
  if [ "$name" = r4flags ] && [ $f = planesweep.hip ]; then extra="-fno-slp-vectorize"; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed $extra -c $f -o build_$name/${f%.hip}.o 2>/dev/null &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map build_$name/*.o -o ../../scratch/lib/libmvsnerf_hip_$name.so
ls -la ../../scratch/lib/libmvsnerf_hip_$name.so
