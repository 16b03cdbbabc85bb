#!/bin/bash
# Builds the round-5 prototypes into scratch/r5_prep/libr5.so (gfx950; uses the product's headers, nothing of it is linked into libmvsnerf_hip.so)
set -e
cd "$(dirname "$0")"
C=../../mvsnerf_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-pass-failed -I$C -I../../include conv_f16x3_tiled.hip -o libr5.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-pass-failed -I$C -I../../include -DR5_PIPELINED conv_f16x3_tiled.hip -o libr5_pipelined.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-pass-failed -I$C -I../../include -DR5_PIPELINED -DR5_HOIST conv_f16x3_tiled.hip -o libr5_hoist.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-pass-failed -I$C -I../../include -DR5_PIPELINED -DR5_HOIST -S --cuda-device-only conv_f16x3_tiled.hip -o /tmp/r5h.s 2>/dev/null
for v in NO_MFMA NO_STAGE NO_PREFETCH; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result -Wno-pass-failed -I$C -I../../include -DR5_PIPELINED -DR5_$v conv_f16x3_tiled.hip -o libr5_$v.so & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-pass-failed -I$C -I../../include -DR5_PIPELINED -S --cuda-device-only conv_f16x3_tiled.hip -o /tmp/r5p.s 2>/dev/null
grep -E "amdhsa_kernel |next_free_vgpr|private_segment_fixed" /tmp/r5p.s | tr -s '\t ' ' ' | grep -A2 tiled
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-pass-failed -I$C -I../../include -S --cuda-device-only conv_f16x3_tiled.hip -o /tmp/r5.s 2>/dev/null
grep -E "amdhsa_kernel |next_free_vgpr|private_segment_fixed|accum_offset" /tmp/r5.s | tr -s '\t ' ' '
