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

# the shipped bf16 tiled convolution with the hoisted prefetch addressing: libmvsnerf_hip_bf16hoist.so = the product objects with conv3d_bf16.o swapped
# (point mvsnerf_amd._lib.LIB_PATH at it before the first lib() call, e.g. PSW_LIB of scratch/r4/race_probe2.py); static instruction counts inside the tile loops:
OTH=$(ls $C/build/*.o | grep -v conv3d_bf16.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -I$C -I../../include -DR5_HOIST -c conv3d_bf16_variant.hip -o /tmp/c3b_h.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/exports.map $OTH /tmp/c3b_h.o -o libmvsnerf_hip_bf16hoist.so
for d in "" "-DR5_HOIST"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result -Wno-pass-failed -I$C -I../../include $d -S --cuda-device-only conv3d_bf16_variant.hip -o "/tmp/c3b$d.s" 2>/dev/null; done
python3 - <<'PY'
import re
for fn in ('/tmp/c3b.s', '/tmp/c3b-DR5_HOIST.s'):
    txt = open(fn).read()
    for kern in re.findall(r'\n(_ZN12_GLOBAL__N_122conv_bf16_tiled_kernel\w+):', txt):
        body = txt.split('\n' + kern + ':')[1].split('s_endpgm')[0].split('\n')
        best = None
        for i, l in enumerate(body):
            m = re.match(r'(\.LBB\d+_\d+):.*Loop Header', l)
            if m:
                for j in range(len(body) - 1, i, -1):
                    if re.search(r's_c?branch\w* ' + re.escape(m.group(1)) + r'\b', body[j]):
                        n = sum('v_mfma' in x for x in body[i:j])
                        if best is None or n > best[0]: best = (n, body[i:j])
                        break
        n, b = best
        print(fn.split('/')[-1], kern[36:60], 'tile loop: mfma', n, 'other VALU', sum(1 for x in b if re.match(r'\s+v_', x) and 'v_mfma' not in x),
              'quarter-rate int', sum(1 for x in b if re.search(r'v_mul_lo_u32|v_mul_hi_u32|v_mad_u64_u32', x)), 'SALU', sum(1 for x in b if re.match(r'\s+s_', x)))
PY
