#!/bin/bash
# Is the round-4 plane-sweep mismatch next to 16-bit MFMA waves (profiles/r04_pk_mfma_hazard.txt) a read of uninitialised LDS?  Builds the sweep WITH packed fp32
# instructions (no -fno-slp-vectorize: the victim build of round 4) in three forms - as it is, every LDS word zeroed at kernel start, every LDS word NaN at kernel start -
# links each into a copy of the library and runs scratch/r5/psw_uninit_probe.py on a GPU box.
set -e
cd "$(dirname "$0")/../.." && git apply scratch/r5/psw_lds_fill.patch && trap "git checkout mvsnerf_amd/csrc/planesweep.hip" EXIT
cd mvsnerf_amd/csrc
mkdir -p ../../scratch/lib /tmp/psw
OTH=$(ls build/*.o | grep -v planesweep.o)
build() { # name, extra flags
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed $2 -c planesweep.hip -o /tmp/psw/$1.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map $OTH /tmp/psw/$1.o -o ../../scratch/lib/libmvsnerf_hip_psw_$1.so
}
build pk ""
build pk_zero "-DPSW_LDS_FILL=0"
build pk_nan "-DPSW_LDS_FILL=0x7fc00000"
build nopk_nan "-fno-slp-vectorize -DPSW_LDS_FILL=0x7fc00000"
ls ../../scratch/lib/
