// What the 16-bit-operand MLP kernels share (mlp_bf16.hip: v_mfma_f32_32x32x16_bf16 and its split variants; mlp_f16x3.hip:
// v_mfma_f32_32x32x16_f16): the k-step geometry of a 32x32x16 instruction on the transposed-layer scheme of mlp_layout.h, the
// weight-pack argument block, accumulator initialisation from the fragment-ordered bias vectors and the in-register positional
// encoding.  Both instructions use the same operand layout (lane = row/column l & 31, k elements 8 * (l >> 5) .. + 7), so the
// k-maps below serve both.
#pragma once
#include "common.h"
#include "mlp_layout.h"

namespace mlp {

constexpr int B_PE_STEPS = 4;        // 64 padded embedding inputs / 16
constexpr int B_ACT_STEPS = 8;       // 128 / 16
constexpr int B_VIEW_STEPS = 9;      // 128 feature + 3 dir (+13 zero) / 16

__host__ __device__ inline int b_feat_steps(int F) { return ((F / 2) + 7) / 8; }
__host__ __device__ inline size_t b_seg(int steps, int nb) { return (size_t)steps * nb * 64 * 8; }      // bf16 elements

// input column of (element t of the lane half h); t = 8*step + j
__host__ __device__ inline int b_col(int kmap, int t, int h, int F)
{
    switch (kmap) {
    case K_PE:    return t < PE_STEPS ? kmap_col(K_PE, t, h, F) : -1;
    case K_FEAT:  return t < F / 2 ? h * (F / 2) + t : -1;
    case K_ACT:   return t < 64 ? act_n(t, h) : -1;
    case K_VIEWS: return t < 64 ? act_n(t, h) : t == 64 ? WIDTH + h : t == 65 ? (h ? -1 : WIDTH + 2) : -1;
    }
    return -1;
}

struct PackBArgs { const float* w[11]; int F; };

template <int NBLK>
__device__ __forceinline__ void init_acc_b(f32x16 (&acc)[NBLK], const float* __restrict__ vec_h)
{
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(vec_h + b * 16 + r4 * 4);
            acc[b][r4 * 4 + 0] = v[0]; acc[b][r4 * 4 + 1] = v[1]; acc[b][r4 * 4 + 2] = v[2]; acc[b][r4 * 4 + 3] = v[3];
        }
}

// sin / cos of x * 2^f (Embedder, models.py:47-51) on the transcendental unit.  v_sin_f32 takes REVOLUTIONS and is good to 1.25e-7 absolute over a period
// (scratch/r6/sin_probe.hip, pe_probe.hip; the polynomial routine of mlp.hip: ~22 instructions per value, this: 7); what matters is the range reduction, done exactly:
// u = x / (2 pi) is kept as an unevaluated sum hi + lo (two-constant product with the rounding error of the first recovered by FMA: ~45 bits), hi * 2^f is exact,
// its fractional part is exact, and lo * 2^f joins by one FMA - the argument of v_sin_f32 is off by half an ulp of a number below one (1.5e-8 revolutions).
// |x| is clamped to 2^15 (beyond it 2^9 x / 2 pi has no fraction bits).
struct PeArg { float hi, lo; };
__device__ __forceinline__ PeArg pe_arg(float x)
{
#pragma clang fp contract(off)      // hi must be the ROUNDED product: a fused x * (C_HI * 2^f) + 1/4 downstream would count its rounding error twice (4.8e-5 at f = 9)
    x = fminf(fmaxf(x, -32768.0f), 32768.0f);
    constexpr float C_HI = 0x1.45f306p-3f, C_LO = 0x1.b9391p-28f;          // 1 / (2 pi) = C_HI + C_LO + O(2^-52)
    PeArg a;
    a.hi = x * C_HI;
    asm("" : "+v"(a.hi));                                                    // (and opaque, so that no later pass re-derives it from x)
    a.lo = __builtin_fmaf(x, C_LO, __builtin_fmaf(x, C_HI, -a.hi));
    return a;
}
__device__ __forceinline__ float pe_sc(PeArg a, int f, int want_cos)
{
#pragma clang fp contract(off)
    const float two_f = (float)(1 << f);
    const float t = a.hi * two_f;                                            // exact
    const float fr = t - __builtin_rintf(t);                                 // exact, in [-0.5, 0.5]: where v_sin_f32 is at its best (1.7e-7; on [0.5, 1): 4e-7)
    const float r = __builtin_fmaf(a.lo, two_f, fr);
    // cos(2 pi r) = sin(2 pi (1/4 - |r|)), again inside [-1/4, 1/4].  (Adding the quarter turn BEFORE the reduction - t + 1/4 - is not exact where the sum crosses a
    // power of two: 4.8e-5 at f = 9, scratch/r6/pe_probe.hip.)
    return __builtin_amdgcn_sinf(want_cos ? 0.25f - __builtin_fabsf(r) : r);
}

// the polynomial routine of mlp.hip (rounds 1-5): what the bf16 kernels keep (their parity test pins them to a torch emulation of bf16 rounding within 2e-3,
// and one rounding boundary crossed by a 1e-7 difference in an input moves an output by more)
__device__ __forceinline__ float pe_sc_poly(float x, int want_cos)
{
    x = fminf(fmaxf(x, -65536.0f), 65536.0f);
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(k, -1.5703125f, x);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188e-8f, r);
    const float r2 = r * r;
    const float sn = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
    const float cs = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f) * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k + want_cos;
    const float v = (q & 1) ? cs : sn;
    return (q & 2) ? -v : v;
}

// operand t of the lane half: t = 0: (x, y), 1: (z, 0), t >= 2: (sin, cos) of coordinate (t - 2) % 3 at frequency 2^((t - 2) / 3)
template <bool HW_SIN>
__device__ __forceinline__ float pe_op_t(int t, int half, float px, float py, float pz)
{
    if (t == 0) return half ? py : px;
    if (t == 1) return half ? 0.0f : pz;
    const int j = t - 2, f = j / 3, c = j - 3 * f;
    const float x = c == 0 ? px : c == 1 ? py : pz;
    return HW_SIN ? pe_sc(pe_arg(x), f, half) : pe_sc_poly(x * (float)(1 << f), half);
}
__device__ __forceinline__ float pe_op(int t, int half, float px, float py, float pz) { return pe_op_t<false>(t, half, px, py, pz); }
__device__ __forceinline__ float pe_op_hw(int t, int half, float px, float py, float pz) { return pe_op_t<true>(t, half, px, py, pz); }

// v_sin_f32 is a transcendental: an instruction that consumes its result needs one wait state, which the compiler inserts for the instructions it knows - not for
// the ones inside an asm statement.  Call this between pe_op_hw() values and an asm that reads them (first fp16x3 version: wrong pieces wherever a v_cvt_pk followed
// its v_sin directly; sigma off by 1e-4).
__device__ __forceinline__ void trans_fence8(float (&t)[8])
{
    asm volatile("s_nop 0" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
}

// A wave's feature operands (32 points x F floats, the lane (m, half) wants columns [half F/2, (half + 1) F/2) of point m) fetched as WHOLE 16-byte vectors
// - the block is contiguous when feat_stride == F - through a wave-private LDS stage of 32 * F * 4 bytes, instead of F / 2 dword loads per lane at an 80-byte
// stride (~20 cache lines per wave-instruction, one scalar branch + wait per element as the compiler lays them out: the longest stretch of a wave's prologue).
// Returns false - nothing touched - when the block is not contiguous / aligned / fully inside the tensor or F > 32; the caller then loads per lane.
template <int MAXV>
__device__ __forceinline__ bool stage_features(float (&fv)[MAXV], const float* __restrict__ feat, int feat_stride, int F, int64_t first_point, int64_t P,
                                               char* __restrict__ stage, int lane)
{
    if (feat_stride != F || F > 32 || first_point + 32 > P || ((reinterpret_cast<uintptr_t>(feat) | (uintptr_t)(first_point * F * 4)) & 15)) return false;
    const f32x4* __restrict__ g = reinterpret_cast<const f32x4*>(feat + first_point * F);
    f32x4* __restrict__ st4 = reinterpret_cast<f32x4*>(stage);
    const int n4 = 8 * F;                                                     // 16-byte vectors of the block (<= 256)
    f32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = k * 64 + lane; v[k] = g[i < n4 ? i : 0]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int i = k * 64 + lane; if (i < n4) st4[i] = v[k]; }
    const float* __restrict__ mine = reinterpret_cast<const float*>(stage) + (lane & 31) * F + (lane >> 5) * (F / 2);
    const int n_half = F / 2;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) fv[i] = mine[i < n_half ? i : 0];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) fv[i] = i < n_half ? fv[i] : 0.0f;
    return true;
}

}  // namespace mlp

// mlp_f16x3.hip: the two-piece fp16 split behind mvsnerf_mlp_{packed_split_elems, pack_split, fwd_split}(n_split = MVSNERF_SPLIT_FP16)
size_t mvs_mlp_f16x3_elems(int F);
int mvs_mlp_f16x3_pack(const float* const w[11], int F, void* packed, hipStream_t st);
int mvs_mlp_f16x3_fwd(const void* packed_h, const float* packed_f32, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                      const float* dirs, int dirs_stride, int64_t P, int S, int alpha_only, float* raw, hipStream_t st, int* guard = nullptr);
