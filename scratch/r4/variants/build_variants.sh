#!/bin/bash
# Variant libraries for the shared-GPU probes: the product objects with ONE source swapped for a variant.  usage: build_variants.sh
set -e
cd "$(dirname "$0")"
# encoder_variant.hip is generated (the encoder.hip of the revision before the sweep moved + the variant macros); the variant libraries then hold the OLD sweep
# twice (this object and the product's planesweep.o) - link against objects of that revision, or use build_sweep_variants.sh for variants of the current sweep
[ -f encoder_variant.hip ] || python make_encoder_variant.py
C=../../../mvsnerf_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -I$C -I../../../include"
OTHERS=$(ls $C/build/*.o | grep -v encoder.o)
build() {  # name, flags
  /opt/rocm/bin/hipcc $F $2 -c encoder_variant.hip -o enc_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map $OTHERS enc_$1.o -o lib_$1.so
  rm -f enc_$1.o
}
build base ""                                  & 
build 4waves "-DPSW_VARIANT_4WAVES"            &
build occ4 "-DPSW_VARIANT_WAVES_PER_EU=4"      &
wait
ls -la lib_*.so
# aggressor-side variants of the fp16x3 conv0 (its results are then garbage: only the sweep next to it is checked)
OTH2=$(ls $C/build/*.o | grep -v conv_f16x3.o)
buildag() {
  /opt/rocm/bin/hipcc $F $2 -c conv_f16x3_variant.hip -o ag_$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map $OTH2 ag_$1.o -o lib_ag_$1.so
  rm -f ag_$1.o
}
buildag nodma "-DAG_NO_DMA" &
buildag nomfma "-DAG_NO_MFMA" &
wait
ls -la lib_ag_*.so
