#!/bin/bash
# Occupancy variants of the (unpacked) plane sweep: the product objects with planesweep.o swapped.  usage: build_sweep_variants.sh
set -e
cd "$(dirname "$0")"
C=../../../mvsnerf_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -I$C -I../../../include -fno-slp-vectorize"
OTHERS=$(ls $C/build/*.o | grep -v planesweep.o)
for w in 4 5 6; do
  ( /opt/rocm/bin/hipcc $F -DPSW_VARIANT_WAVES_PER_EU=$w -c planesweep_variant.hip -o sw_occ$w.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map $OTHERS sw_occ$w.o -o lib_sw_occ$w.so
    /opt/rocm/bin/hipcc $F -DPSW_VARIANT_WAVES_PER_EU=$w -S --cuda-device-only planesweep_variant.hip -o sw_occ$w.s 2>/dev/null
    echo "occ$w: $(grep -A40 'amdhsa_kernel _Z17planesweep_kernel' sw_occ$w.s | grep -E 'next_free_vgpr|private_segment_fixed' | tr -s ' \t' ' ' | tr '\n' ';')"
    rm -f sw_occ$w.o sw_occ$w.s ) &
done
wait
ls lib_sw_*.so
