#!/bin/sh
# usage: check_isa.sh <libmvsnerf_hip.so> <scratch dir> [arch = gfx950]        (ALLOW_NO_ISA_CHECK=1: go on without a disassembler)
# Fails if any gfx950 code object bundled in the library holds `v_pk_{fma,mul,add}_f32 ... op_sel:[x,1...]`: src1's HIGH half feeding the LOW result.
# That operand select is the one packed fp32 form measured to return wrong values (lanes 48-63 only) while ANOTHER wave of the same SIMD issues
# v_mfma_f32_16x16x32_{f16,bf16} - from another stream or process; no other matrix shape does it; the unselected forms, op_sel on
# src0 / src2, op_sel_hi and neg are clean (scratch/keep/pk_probe.hip + pk_hog.hip, profiles/r05_pk_fma_opsel_reproducer.txt).  The compiler picks the form, so the
# check is on the shipped bits.
set -e
LIB="$1"; DIR="$2"; ARCH="${3:-gfx950}"; ARCH="${ARCH%%:*}"      # (a feature suffix such as gfx950:xnack- names the same code objects)
OBJDUMP="${OBJDUMP:-/opt/rocm/lib/llvm/bin/llvm-objdump}"
if [ ! -x "$OBJDUMP" ]; then      # no disassembler on this box: the library is NOT shipped unchecked unless the builder says so
    if [ "$ALLOW_NO_ISA_CHECK" = "1" ]; then
        echo "check_isa: $OBJDUMP not found - ISA check SKIPPED (ALLOW_NO_ISA_CHECK=1)"
        exit 0
    fi
    echo "check_isa: $OBJDUMP not found: cannot check the code objects for the packed-fp32 op_sel form (set OBJDUMP, or ALLOW_NO_ISA_CHECK=1 to build without the check)"
    exit 1
fi
rm -rf "$DIR"; mkdir -p "$DIR"
cp "$LIB" "$DIR/lib.so"
( cd "$DIR" && "$OBJDUMP" --offloading lib.so > /dev/null 2>&1 )
n_obj=0; bad=0
for f in "$DIR"/lib.so.*"$ARCH"*; do
    [ -f "$f" ] || continue
    n_obj=$((n_obj + 1))
    n=$("$OBJDUMP" -d "$f" | grep -cE "v_pk_(fma|mul|add)_f32.*op_sel:\[[01],1" || true)
    if [ "$n" != "0" ]; then
        echo "check_isa: $n packed fp32 instruction(s) with op_sel on src1 in $(basename "$f"):"
        "$OBJDUMP" -d "$f" | awk '/^[0-9a-f]+ <.*>:$/{k=$2} /v_pk_(fma|mul|add)_f32.*op_sel:\[[01],1/{c[k]++} END{for (k in c) print "    " c[k], k}'
        bad=1
    fi
done
if [ "$n_obj" = "0" ]; then echo "check_isa: no $ARCH code object found in $LIB"; exit 1; fi
rm -f "$DIR"/lib.so "$DIR"/lib.so.*
[ "$bad" = "0" ] && echo "check_isa: $n_obj code objects, no packed fp32 op_sel on src1"
exit $bad
