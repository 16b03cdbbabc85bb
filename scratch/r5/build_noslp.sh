#!/bin/bash
# Every code object of the library WITHOUT the SLP vectoriser's packed fp32 arithmetic (v_pk_*_f32): scratch/lib/libmvsnerf_hip_noslp.so.
# A/B on one box: copy it over mvsnerf_amd/lib/libmvsnerf_hip.so of the box's scratch copy (never in the tree).
set -e
cd "$(dirname "$0")/../../mvsnerf_amd/csrc"
mkdir -p build_noslp ../../scratch/lib
for f in *.hip; do
  o=build_noslp/${f%.hip}.o
  if [ ! -f $o ] || [ $f -nt $o ] || [ -n "$(find . -maxdepth 1 -name '*.h' -newer $o)" ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -fno-slp-vectorize -c $f -o $o &
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map build_noslp/*.o -o ../../scratch/lib/libmvsnerf_hip_noslp.so
ls -la ../../scratch/lib/libmvsnerf_hip_noslp.so
