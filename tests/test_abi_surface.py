"""CPU-only: the C-ABI library builds/loads, exports every symbol include/mvsnerf_hip.h declares, and the ctypes
table binds exactly that set.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess

from mvsnerf_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "mvsnerf_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvsnerf_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_header_symbols():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    names = _declared()
    assert len(names) >= 20
    l = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(l, n)]
    assert not missing, f"declared in the header but not exported: {missing}"
    assert sorted(_lib.SIGNATURES) == names, (set(names) ^ set(_lib.SIGNATURES))
    assert _lib.lib().mvsnerf_abi_version() == 12
    # the dynamic symbol table is the header and nothing else (csrc/exports.map): no kernel host stubs, no C++-mangled helpers, no
    # A/B switches or diagnostics state
    exported = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exp_names = sorted(ln.split()[-1] for ln in exported.splitlines() if ln.strip())
    assert exp_names == names, sorted(set(exp_names) ^ set(names))[:10]
    # pure host-side queries work without a GPU
    # 32-point layout (12 feature k-steps x 4 blocks x 64 lanes, ...)
    n32 = 12 * 256 + 8192 * 2 + 16384 * 6 + 68 * 128 + 1416
    assert _lib.lib().mvsnerf_mlp_packed_floats(20) == n32
    assert _lib.lib().mvsnerf_mlp_packed_floats(21) == 0


def test_only_gfx950_code_objects():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={_lib.LIB_PATH}"],
                         capture_output=True, text=True).stdout
    if out.strip():
        assert "gfx950" in out and not re.search(r"gfx9(0[0-9a]|4[0-9])\b", out)


def test_product_path_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "mvsnerf_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("# oracle", ""), f"{f} mentions the oracle"


def test_plane_sweep_code_object_has_no_packed_fp32_arithmetic():
    """csrc/planesweep.hip must be built without packed fp32 instructions: next to 16-bit MFMA waves of another stream or process they computed wrong lanes
    (tests/test_gpu_costream.py is the GPU side).  The Makefile carries the flag and greps the code object; this compiles the file to ISA with
    the Makefile's own flags (hipcc cross-compiles without a GPU) and looks again."""
    csrc = os.path.join(ROOT, "mvsnerf_amd", "csrc")
    mk = open(os.path.join(csrc, "Makefile")).read()
    rule = mk[mk.index("build/planesweep.o:"):]
    assert "-fno-slp-vectorize" in rule.split("\n\n")[0], "the plane sweep's build rule lost -fno-slp-vectorize"
    flags = re.search(r"^FLAGS\s*=\s*(.*)$", mk, flags=re.M).group(1).replace("$(ARCH)", "gfx950").split()
    out = os.path.join(csrc, "build", "planesweep_check.s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *flags, "-fno-slp-vectorize", "-S", "--cuda-device-only",
                        os.path.join(csrc, "planesweep.hip"), "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    isa = open(out).read()
    os.remove(out)
    assert "planesweep_kernel" in isa and "planesweep_if_kernel" in isa
    packed = re.findall(r"\bv_pk_\w+_f32\b", isa)
    assert not packed, f"{len(packed)} packed fp32 instructions in the plane sweep's code object: {sorted(set(packed))}"


def test_no_code_object_selects_the_high_half_of_src1_in_packed_fp32():
    """`v_pk_{fma,mul,add}_f32 ... op_sel:[x,1...]` (src1's high half feeding the low result) is the packed fp32 form that returned wrong values in lanes
    48-63 next to the fp16x3 conv0 of another stream (scratch/r5/pk_probe.hip, profiles/r05_pk_fma_opsel_reproducer.txt); the compiler picks the form, so
    the shipped bits are disassembled: csrc/check_isa.sh, which the Makefile also runs after linking."""
    csrc = os.path.join(ROOT, "mvsnerf_amd", "csrc")
    assert "check_isa.sh $@" in open(os.path.join(csrc, "Makefile")).read(), "the link rule lost its ISA check"
    r = subprocess.run(["sh", os.path.join(csrc, "check_isa.sh"), _lib.LIB_PATH, os.path.join(csrc, "build", "isa_test")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    n = int(re.search(r"check_isa: (\d+) code objects", r.stdout).group(1))
    assert n >= 18, r.stdout


def test_isa_check_rejects_a_library_that_holds_the_form(tmp_path):
    """The checker itself: a ten-line kernel with `v_pk_fma_f32 ... op_sel:[0,1,0]` (inline asm) linked into a shared object must fail check_isa.sh with the
    kernel's name in the report; the same kernel with the select on src0 must pass."""
    csrc = os.path.join(ROOT, "mvsnerf_amd", "csrc")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    for sel, want_rc in (("op_sel:[0,1,0]", 1), ("op_sel:[1,0,0]", 0)):
        src = tmp_path / f"k{want_rc}.hip"
        src.write_text('#include <hip/hip_runtime.h>\n'
                       'typedef float f32x2 __attribute__((ext_vector_type(2)));\n'
                       '__global__ void isa_check_probe_kernel(f32x2* p) {\n'
                       '    f32x2 a = p[threadIdx.x], b = p[threadIdx.x + 64], c = p[threadIdx.x + 128];\n'
                       f'    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 {sel}" : "=v"(c) : "v"(a), "v"(b), "v"(c));\n'
                       '    p[threadIdx.x] = c;\n'
                       '}\n')
        lib = tmp_path / f"libk{want_rc}.so"
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", str(src), "-o", str(lib)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        r = subprocess.run(["sh", os.path.join(csrc, "check_isa.sh"), str(lib), str(tmp_path / f"isa{want_rc}")], capture_output=True, text=True)
        assert r.returncode == want_rc, r.stdout + r.stderr
        if want_rc:
            assert "isa_check_probe_kernel" in r.stdout and "1 packed fp32 instruction" in r.stdout, r.stdout
