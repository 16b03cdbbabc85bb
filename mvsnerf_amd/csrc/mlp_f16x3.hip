// "fp16x3": fp32-grade results of the fused Embedder + Renderer_ours forward (models.py:145-222) from THREE v_mfma_f32_32x32x16_f16 per
// product.  Inference only.  What a no-grad rendering() runs by default (ops.MLP_PRECISION = "auto") as a GUARDED sequence: the kernel reports
// what it cannot represent (a non-finite value, a non-finite weight) through the guard word and the fp32-MFMA kernel of mlp.hip, enqueued right
// behind it and predicated on that word, recomputes the batch (include/mvsnerf_hip.h, "guarded 16-bit sequences").
// ops.set_mlp_precision("fp16x3") is the unguarded kernel alone; the headline of bench.py stays on the fp32-MFMA kernel.
//
// Every fp32 operand is written as the sum of two fp16 pieces, both rounded to nearest:
//     a = a0 + a1 + r,   a0 = fp16(a),  a1 = fp16(a - a0),   |r| <= 2^-22 |a|      (fp16 carries 11 significant bits; a - a0 is exact)
//     a * w  ~=  a0*w0 + a0*w1 + a1*w0                                              (dropped: a1*w1 <= 2^-22 |a*w|)
// The piece products are exact in fp32 and the matrix core accumulates them in fp32, so a product is off by ~3 * 2^-22 - the same order as
// the roundings of an fp32 kernel (scratch/keep/f16x3_numerics.py on the shipped weights: sigma 2.4e-6 from the float64 result, the torch
// fp32 path 2.9e-6, the two-piece BF16 split 1.2e-4).  The three-piece bf16 split of mlp_bf16.hip ("bf16x6") needs six instructions of the
// same rate for that.
//
// EXPONENT MANAGEMENT (round 6).  fp16 has five exponent bits: a second piece below 2^-14 is a subnormal (absolute resolution 2^-24) and an operand above
// 65504 does not exist.  Both ends are handled by exact power-of-two scales instead of a fallback:
//   * weights, at pack time: every K-block of every layer (pts_bias | layer 0 | layers 1-4 | layer 5's encoding columns | layer 5's h columns | feature_linear |
//     views' feature columns | views' direction columns) whose largest |w| lies outside [2^-4, 2^12] is stored times 2^kw (largest element in [2^4, 2^5));
//     the eleven exponents travel in the status tail behind the packed planes;
//   * activations, in the kernel: per point and layer (`condition` below).
// relu(W (s h) + s b) = s relu(W h + b): a scale, once applied, is inherited by every later layer; the kernel tracks the point's cumulative exponent (`lsc`),
// feeds bias vectors, encoding and view direction times the scale their accumulators carry, and divides in the two heads.  In range (the shipped network on
// every input of the test suite) all of it is a handful of wave-uniform branches that are not taken.
//
// Structure: the transposed-layer scheme of mlp_layout.h - 32 points per wave, the C/D fragment of one layer IS the B operand of the
// next, activations never leave the register file (fp32; split into their fp16 pieces per k-step inside the GEMM loop) - with EIGHT waves
// (256 points) per workgroup sharing each layer's weights: a layer is one 64 KB slab (hi plane | lo plane), two slabs alternate in LDS,
// the next one arrives by LDS-DMA while the current one is multiplied (one barrier per layer), 132 KB of LDS = one workgroup = two
// waves per SIMD.  Per k-step and output block a wave reads one hi and one lo weight fragment (ds_read_b128) for three MFMAs.
//
// What bounds it (round 6, profiles/r06_mfma_power_probe.txt): POWER.  A loop of nothing but v_mfma_f32_32x32x16_f16 on random operands sustains 1.58 PFLOP/s on
// this chip (the same loop on zeros: 2.40; the clock drops from 2.4 to ~1.6 GHz at unchanged cycles per instruction), with this kernel's ds_reads and VALU beside
// it 1.43-1.46: 0.57-0.63 of the nominal 2.5 PFLOP/s is the ceiling of ANY fp16 matrix kernel on real data here, and re-arranging who stalls when (rounds 3-5: anti-phase
// wave groups, block groupings, priorities; round 6: a stall-free software-pipelined stream) moves cycles but not joules.
#include <type_traits>
#include "common.h"
#include "lds_dma.h"
#include "mlp_layout.h"
#include "mlp_b16_dev.h"

using namespace mlp;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int H3_WAVES = 8;
constexpr int H3_THREADS = 64 * H3_WAVES;
constexpr int H3_BUF_BYTES = 65536;                                  // a 128x128 layer: hi plane 32 KB + lo plane 32 KB
constexpr int H3_LDS_BYTES = 2 * H3_BUF_BYTES + V_TOTAL * 4;

constexpr float H_MAX = 65504.0f;                                    // largest finite fp16
constexpr float F_FINITE = 3.0e38f;
// status tail behind the packed weights, 64 bytes: fp16 element [0] != 0: a weight was not finite; bytes 4 .. 47: the largest |w| of the eleven K-blocks (fp32
// bit patterns, written by the first pack pass); bytes 48 .. 58: their scale exponents kw (int8)
constexpr int H3_TAIL = 32;
constexpr int H3_SEGS = 11;
enum Seg { SEG_BIAS = 0, SEG_L0 = 1, SEG_L1 = 2, SEG_L5A = 6, SEG_L5B = 7, SEG_FEAT = 8, SEG_VF = 9, SEG_VD = 10 };
constexpr float WW_LO = 0.0625f, WW_HI = 4096.0f;                    // a K-block whose largest |w| is inside [2^-4, 2^12] is stored as it is
constexpr int WW_TARGET = 5;                                         // ... outside: times 2^kw, largest element in [2^4, 2^5)

__host__ __device__ inline int seg_kw(unsigned max_bits)
{
    float m;
    __builtin_memcpy(&m, &max_bits, 4);
    if (!(m <= F_FINITE) || m == 0.0f || (m >= WW_LO && m <= WW_HI)) return 0;
    int e;
    (void)frexpf(m, &e);                                             // m in [2^(e-1), 2^e)
    const int k = WW_TARGET - e;
    return k < -100 ? -100 : k > 100 ? 100 : k;
}

// packed buffer (fp16 elements), in the order the kernel streams it; every slab = [hi plane | lo plane] of its segment(s)
struct LayoutH { size_t s0, l1, l5a, l5b, feat, views, total; int fsteps; };
__host__ __device__ inline LayoutH layout_h(int F)
{
    LayoutH L;
    L.fsteps = b_feat_steps(F);
    size_t o = 0;
    L.s0 = o;    o += 2 * b_seg(L.fsteps, 4) + 2 * b_seg(B_PE_STEPS, 4);       // pts_bias weights, then layer 0
    L.l1 = o;    o += 4 * 2 * b_seg(B_ACT_STEPS, 4);                            // layers 1..4
    L.l5a = o;   o += 2 * b_seg(B_PE_STEPS, 4);
    L.l5b = o;   o += 2 * b_seg(B_ACT_STEPS, 4);
    L.feat = o;  o += 2 * b_seg(B_ACT_STEPS, 4);
    L.views = o; o += 2 * b_seg(B_VIEW_STEPS, 2);
    L.total = o;                                                    // followed by the H3_TAIL status elements
    return L;
}

// the K-blocks as (weight tensor, leading dimension, first column, columns, rows)
struct SegSrc { int widx, ld, col0, ncol, rows; };
__host__ __device__ inline SegSrc seg_src(int seg, int F)
{
    switch (seg) {
    case SEG_BIAS: return {6, F, 0, F, WIDTH};
    case SEG_L0:   return {0, PE_DIM, 0, PE_DIM, WIDTH};
    case SEG_L5A:  return {5, WIDTH + PE_DIM, 0, PE_DIM, WIDTH};
    case SEG_L5B:  return {5, WIDTH + PE_DIM, PE_DIM, WIDTH, WIDTH};
    case SEG_FEAT: return {7, WIDTH, 0, WIDTH, WIDTH};
    case SEG_VF:   return {9, WIDTH + 3, 0, WIDTH, WIDTH / 2};
    case SEG_VD:   return {9, WIDTH + 3, WIDTH, 3, WIDTH / 2};
    default:       return {seg - SEG_L1 + 1, WIDTH, 0, WIDTH, WIDTH};          // layers 1..4
    }
}

// pass 1: the largest |w| per K-block (bit patterns of non-negative floats order like unsigned integers; a NaN orders above infinity)
__global__ __launch_bounds__(256) void mlp_h3_wmax_kernel(PackBArgs a, unsigned* __restrict__ wmax)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (int seg = 0; seg < H3_SEGS; ++seg) {
        const SegSrc q = seg_src(seg, a.F);
        unsigned m = 0;
        for (int i = tid; i < q.rows * q.ncol; i += nt) {
            const float w = a.w[q.widx][(size_t)(i / q.ncol) * q.ld + q.col0 + i % q.ncol];
            m = max(m, __float_as_uint(fabsf(w)) & 0x7fffffffu);
        }
        for (int o = 32; o; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
        if ((threadIdx.x & 63) == 0 && m) atomicMax(wmax + seg, m);
    }
}

// one segment: hi plane at dst[0 .. n), lo plane at dst[n .. 2n), n = steps * nb * 512; element order of a plane = pack_b_segment's; columns below col_split
// times 2^kw, from col_split on times 2^kw2 (views: feature | direction)
__device__ inline void pack_h_planes(_Float16* __restrict__ dst, const float* __restrict__ W, int ld, int col_off, int kmap,
                                     int steps, int nb, int F, int tid, int nthreads, _Float16* __restrict__ bad, int kw, int col_split = 1 << 30, int kw2 = 0)
{
    const int n = steps * nb * 64 * 8;
    for (int i = tid; i < n; i += nthreads) {
        const int j = i & 7, lane = (i >> 3) & 63, rest = i >> 9;          // rest = s*nb + b
        const int b = rest % nb, s = rest / nb;
        const int col = b_col(kmap, 8 * s + j, lane >> 5, F);
        const int row = b * 32 + (lane & 31);
        float w = col < 0 ? 0.0f : ldexpf(W[(size_t)row * ld + col_off + col], col >= col_split ? kw2 : kw);
        if (!(fabsf(w) <= H_MAX)) *bad = (_Float16)1.0f;                 // not finite: the guarded sequence then always takes the fp32 kernel
        w = fminf(fmaxf(w, -H_MAX), H_MAX);
        const _Float16 hi = (_Float16)w;
        dst[i] = hi;
        dst[n + i] = (_Float16)(w - (float)hi);
    }
}

// pass 2
__global__ __launch_bounds__(256) void mlp_pack_h3_kernel(PackBArgs a, _Float16* __restrict__ packed)
{
    const LayoutH L = layout_h(a.F);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const size_t act = 2 * b_seg(B_ACT_STEPS, 4);
    _Float16* tail = packed + L.total;
    const unsigned* wmax = reinterpret_cast<const unsigned*>(tail) + 1;
    int kw[H3_SEGS];
    for (int sg = 0; sg < H3_SEGS; ++sg) kw[sg] = seg_kw(wmax[sg]);
    if (tid == 0) for (int sg = 0; sg < H3_SEGS; ++sg) reinterpret_cast<signed char*>(tail)[48 + sg] = (signed char)kw[sg];
    pack_h_planes(packed + L.s0, a.w[6], a.F, 0, K_FEAT, L.fsteps, 4, a.F, tid, nt, tail, kw[SEG_BIAS]);
    pack_h_planes(packed + L.s0 + 2 * b_seg(L.fsteps, 4), a.w[0], PE_DIM, 0, K_PE, B_PE_STEPS, 4, a.F, tid, nt, tail, kw[SEG_L0]);
    for (int l = 1; l <= 4; ++l) pack_h_planes(packed + L.l1 + (l - 1) * act, a.w[l], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt, tail, kw[SEG_L1 + l - 1]);
    pack_h_planes(packed + L.l5a, a.w[5], WIDTH + PE_DIM, 0, K_PE, B_PE_STEPS, 4, a.F, tid, nt, tail, kw[SEG_L5A]);
    pack_h_planes(packed + L.l5b, a.w[5], WIDTH + PE_DIM, PE_DIM, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt, tail, kw[SEG_L5B]);
    pack_h_planes(packed + L.feat, a.w[7], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt, tail, kw[SEG_FEAT]);
    pack_h_planes(packed + L.views, a.w[9], WIDTH + 3, 0, K_VIEWS, B_VIEW_STEPS, 2, a.F, tid, nt, tail, kw[SEG_VF], WIDTH, kw[SEG_VD]);
}

// ------------------------------------------------------------------------------------------ kernel
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// the two fp16 pieces of eight fp32 values, packed two per register
struct BP { u32x4 hi, lo; };
__device__ __forceinline__ f16x8 as_h8(u32x4 v) { return __builtin_bit_cast(f16x8, v); }

// Pieces of two values in three instructions: hi = fp16(v) for both (v_cvt_pk_f16_f32, round to nearest even), lo = fp16(v - hi) straight from the packed hi
// pieces (v_fma_mix{lo,hi}_f16: fp32 FMA of an fp16 source, result rounded to fp16; v - hi is exact in fp32).  hipcc's own lowering of the same arithmetic takes six
// (cvt_pk, two cvt_f32_f16, two subtractions, cvt_pk).  Three statements, so that the GEMM loop can put each into another gap between two matrix instructions.
// The compiler does not see into the statements: a matrix instruction that reads a piece needs two wait states behind the write - split_pad() where a consumer
// follows at once; inside gemm_h the last write of a k-step's pieces is a matrix instruction and a ds_read away from their first reader.
__device__ __forceinline__ void split_p0(const float* v8, int j, BP& d) { unsigned r; asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(v8[2 * j]), "v"(v8[2 * j + 1])); d.hi[j] = r; }
__device__ __forceinline__ void split_p1(const float* v8, int j, BP& d) { unsigned r; asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(d.hi[j]), "v"(v8[2 * j])); d.lo[j] = r; }
__device__ __forceinline__ void split_p2(const float* v8, int j, BP& d) { unsigned r = d.lo[j]; asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(d.hi[j]), "v"(v8[2 * j + 1])); d.lo[j] = r; }
__device__ __forceinline__ void split_pad(BP& d) { asm volatile("s_nop 1" : "+v"(d.hi), "+v"(d.lo)); }
__device__ __forceinline__ BP split8h(const float* v8)
{
    BP r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { split_p0(v8, j, r); split_p1(v8, j, r); split_p2(v8, j, r); }
    split_pad(r);
    return r;
}

// DEV probe (scratch/r6/build_variant.sh; never defined in the product build): -DH3_NO_DMA fetches no weight slab at all (garbage results) - the
// floor of MFMA + VALU + barriers.  Measured without effect and removed: per-workgroup rotation of the piece order, non-temporal DMA loads.
__device__ __forceinline__ void slab_dma(char* __restrict__ dst, const _Float16* __restrict__ src, size_t n_elems, int wave, int lane)
{
#if defined(H3_NO_DMA)
    (void)dst; (void)src; (void)n_elems; (void)wave; (void)lane;
#else
    lds_dma<H3_WAVES>(dst, src, (int)(n_elems >> 9), wave, lane);      // 1 KB (512 fp16) per wave-instruction (lds_dma.h)
#endif
}

__device__ __forceinline__ void slab_sync()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// acc[nb] += W[block nb] * act over STEPS k-steps of 16.  The instruction stream of a wave is laid out by hand (sched_barrier around every matrix instruction)
// because a wave shares its SIMD's matrix pipe with ONE other wave, the older of the two wins the arbitration, and the younger runs a third of every layer alone:
// whatever its own stream loses there, the pair loses (profiles/r06_mlp_f16x3_census.txt: round 5's compiler-scheduled stream ran 54 cycles per matrix
// instruction alone, the first hand-laid one - fillers bunched behind every third - 42).
//   * per (k-step, output block): three products on one accumulator, smallest first (lo x hi, hi x lo, hi x hi);
//   * in the gap behind EVERY matrix instruction one or two fillers of later work - an in-order wave issues nothing while it waits for the pipe, but up to ~6
//     instructions behind a matrix instruction are free (scratch/r6/mfma_probe.hip): the lo weight fragment of the chain after next (its registers were read for the
//     last time by the instruction just issued), the hi fragment behind the chain's third, and one instruction of the NEXT k-step's activation split per gap;
//   * nothing a matrix instruction reads was issued less than four matrix instructions (128 pipe cycles) before it.
// A wave runs its GEMMs at priority H3_GEMM_PRIO and everything else at 0.
// first: the pieces of k-step 0 (split by the caller, ahead of the layer barrier).  prep(s, j, part, dst): instruction `part` (0..2) of pair j of k-step s - or, for
// operands split beforehand, the whole k-step at (j, part) == (0, 0).
#ifndef H3_GEMM_PRIO
#define H3_GEMM_PRIO 1
#endif
template <int STEPS, int NBLK, typename PREP>
__device__ __forceinline__ void gemm_h(const char* __restrict__ w_hi, const char* __restrict__ w_lo, f32x16 (&acc)[NBLK], int lane, const BP& first, PREP prep)
{
    static_assert(NBLK == 2 || NBLK == 4, "two or four output blocks");
    // one lane base per plane; every fragment of the segment is then an immediate offset of the ds_read (<= 65520 from the hi-plane base)
    const f16x8* __restrict__ fh = reinterpret_cast<const f16x8*>(w_hi) + lane;
    const f16x8* __restrict__ fl = reinterpret_cast<const f16x8*>(w_lo) + lane;
    constexpr int N = STEPS * NBLK;
    f16x8 ah[2], al[2];
    BP b[2];
    b[0] = first;
#if defined(H3_NO_FRAG)      // DEV probe: one fragment pair for the whole GEMM (garbage results): the stream without its ds_reads
    ah[0] = ah[1] = fh[0]; al[0] = al[1] = fl[0];
#else
#pragma unroll
    for (int i = 0; i < 2 && i < N; ++i) { al[i] = fl[i * 64]; ah[i] = fh[i * 64]; }
#endif
    __builtin_amdgcn_s_setprio(H3_GEMM_PRIO);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int s = i / NBLK, nb = i % NBLK, r = i & 1;
#if defined(H3_NO_SPLIT)     // DEV probe: the pieces of k-step 0 serve every k-step (garbage results): the stream without its split VALU
        const f16x8 bh = as_h8(b[0].hi), bl = as_h8(b[0].lo);
        const bool more = false;
#else
        const f16x8 bh = as_h8(b[s & 1].hi), bl = as_h8(b[s & 1].lo);
        const bool more = s + 1 < STEPS;
#endif
        BP& n = b[(s + 1) & 1];
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[r], bh, acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#if !defined(H3_NO_FRAG)
        if (i + 2 < N) al[r] = fl[(i + 2) * 64];
#endif
        if (more) {
            if (NBLK == 4) { prep(s + 1, nb, 0, n); if (nb == 3) prep(s + 1, nb, 1, n); }
            else { prep(s + 1, 2 * nb, 0, n); prep(s + 1, 2 * nb + 1, 0, n); if (nb == 1) { prep(s + 1, 2, 1, n); prep(s + 1, 3, 1, n); } }
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[r], bl, acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            if (NBLK == 4) prep(s + 1, nb, nb == 3 ? 2 : 1, n);
            else if (nb == 0) { prep(s + 1, 0, 1, n); prep(s + 1, 1, 1, n); }
            else { prep(s + 1, 2, 2, n); prep(s + 1, 3, 2, n); }
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[r], bh, acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#if !defined(H3_NO_FRAG)
        if (i + 2 < N) ah[r] = fh[(i + 2) * 64];
#endif
        if (more) {
            if (NBLK == 4) { if (nb != 3) prep(s + 1, nb, 2, n); }
            else if (nb == 0) { prep(s + 1, 0, 2, n); prep(s + 1, 1, 2, n); }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
}

// operands split beforehand (encoding, features, view direction): the whole k-step at j == 0
#define H3_PREP_FROM(ARR) [&](int s, int j, int part, BP& d) { if (j == 0 && part == 0) d = (ARR)[s]; }

constexpr float WIN_LO = 0.125f, WIN_HI = 4096.0f;      // a K-vector whose largest element is inside [2^-3, 2^12] is split as it is
constexpr int WIN_TARGET = 5;                            // ... outside, it is scaled into [2^4, 2^5)
constexpr int SC_LIM = 120;                              // |log2| of the cumulative scale (the multiplicative modulation of six layers takes features of 3e5 to 1e36)
__device__ __forceinline__ float pow2f(int e) { return __builtin_amdgcn_ldexpf(1.0f, min(max(e, -126), 126)); }

template <bool ALPHA_ONLY>
__global__ __launch_bounds__(H3_THREADS) void mlp_fwd_f16x3_kernel(
    const _Float16* __restrict__ wq, const float* __restrict__ packed_f32, int F, const float* __restrict__ ndc, int ndc_stride,
    const float* __restrict__ feat, int feat_stride, const float* __restrict__ dirs, int dirs_stride,
    int64_t P, int S, float* __restrict__ raw, int* __restrict__ guard)
{
    extern __shared__ __attribute__((aligned(16))) char lds_h[];
    char* buf0 = lds_h;
    char* buf1 = lds_h + H3_BUF_BYTES;
    float* vec = reinterpret_cast<float*>(lds_h + 2 * H3_BUF_BYTES);
    const LayoutH L = layout_h(F);
    const Layout LF = layout(F);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int64_t p_raw = ((int64_t)blockIdx.x * H3_WAVES + wave) * 32 + (lane & 31);
    const bool live = p_raw < P;
    const int64_t p = live ? p_raw : P - 1;
    constexpr size_t ACT_PLANE = (size_t)B_ACT_STEPS * 4 * 512, PE_PLANE = (size_t)B_PE_STEPS * 4 * 512;     // fp16 elements of one plane
    constexpr size_t VIEW_PLANE = (size_t)B_VIEW_STEPS * 2 * 512;
    const size_t feat_plane = b_seg(L.fsteps, 4);
#ifdef H3_CENSUS      // DEV probe: shader-clock stamps of this wave's phases, 32 per tile, behind the results (scratch/r6/h3_census.py allocates them)
    unsigned* cen = reinterpret_cast<unsigned*>(raw + P * (ALPHA_ONLY ? 1 : 4)) + ((int64_t)blockIdx.x * H3_WAVES + wave) * 32;
    int cen_i = 0;
#define H3_STAMP() do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (lane == 0 && cen_i < 32) cen[cen_i] = (unsigned)t__; ++cen_i; } while (0)
#else
#define H3_STAMP() do {} while (0)
#endif
    H3_STAMP();                                   // 0: wave start

    // slab 0 = pts_bias weights + layer 0 (contiguous in the packed buffer) -> buf0.  (Layer 1's slab follows behind the first barrier: every workgroup of the
    // launch asks the L2 for the same bytes at the same moment, and the wait for slab 0 is the longest single wait of a wave's life.)
    slab_dma(buf0, wq + L.s0, L.l1 - L.s0, wave, lane);
    // The two waves of a SIMD share one matrix pipe and the OLDER one (waves 0..3 of the workgroup, dispatched first; priority 1 makes it explicit) wins every
    // arbitration between two matrix streams (a wave raises its priority for the length of a GEMM, so that a matrix stream also beats the partner's epilogue): measured, the pair runs its GEMMs one after the other (profiles/r06_mlp_f16x3_census.txt).  The layer barrier is therefore placed differently
    // for the two: an older wave runs [GEMM, epilogue, barrier], a younger one [GEMM, barrier, epilogue] - the barrier falls when the younger wave's GEMM ends, the
    // older wave (epilogue long done) starts the next layer's GEMM at once and the younger wave's epilogue runs beside it instead of in front of it.
    const bool young = wave >= H3_WAVES / 2;
    for (int i = tid; i < V_TOTAL; i += H3_THREADS) vec[i] = packed_f32[LF.vec + i];
    const float px = ndc[p * ndc_stride + 0], py = ndc[p * ndc_stride + 1], pz = ndc[p * ndc_stride + 2];
    // the view direction of the point's ray: asked for HERE, used by the last GEMM (loaded there, its latency sat in front of that GEMM)
    float dir0 = 0.0f, dir1 = 0.0f;
    if (!ALPHA_ONLY) {
        const int64_t ray = ((uint64_t)p >> 32) == 0 ? (int64_t)((unsigned)p / (unsigned)S) : p / S;      // (a 64-bit division is ~10x a 32-bit one)
        dir0 = half ? dirs[ray * dirs_stride + 1] : dirs[ray * dirs_stride + 0];
        dir1 = half ? 0.0f : dirs[ray * dirs_stride + 2];
    }
    float fv[24];                                 // F/2 <= 20 feature operands of this lane half
    // (buf1 is idle until layer 1's slab is asked for behind the first barrier: 4 KB of it per wave stage the features - mlp_b16_dev.h)
    if (!stage_features(fv, feat, feat_stride, F, ((int64_t)blockIdx.x * H3_WAVES + wave) * 32, P, buf1 + wave * 4096, lane)) {
        const float* fp = feat + p * feat_stride + half * (F / 2);
        // (the compiler turns this into one scalar branch + one load per element; issuing all 24 unconditionally - padding slots re-reading element 0 - measured
        // SLOWER, 34.4 -> 41.9 us in the bf16 kernel: a 64-lane dword load at an 80-byte stride is ~20 cache lines per instruction, the address unit is what waits)
#pragma unroll
        for (int i = 0; i < 24; ++i) fv[i] = i < F / 2 ? fp[i] : 0.0f;
    }

    // ---- exponent management.  A K-vector (the operands one point hands to a GEMM: lanes m and m + 32) is split into fp16 pieces as it is while its largest
    // magnitude lies in [WIN_LO, WIN_HI]: the second piece of that element then has >= 10 significant bits (22 in all) and nothing is near fp16's 65504.  Outside the
    // window the point's vector is multiplied by a power of two (exact), which every later layer inherits (relu(W (s h) + s b) = s relu(W h + b)): 2^lsc is the
    // point's cumulative scale (sc) - additive bias vectors enter the accumulators times sc, the positional encoding of the skip layer and the view direction are scaled
    // by sc, the two heads divide by it.  Rare by design (the shipped network never leaves the window on config 2's batch): every use of sc sits behind a
    // wave-uniform `any lane scaled?` branch, so that the common path executes what the unscaled kernel executed.  What still trips the guard: a non-finite value
    // and a weight outside fp16's range (pack time).
    int lsc = 0;
    bool bad = false;
    // the weight blocks' scale exponents (pack time; wave-uniform, in SGPRs): block g of the packed buffer holds W * 2^kw(g)
    const int* kwp = reinterpret_cast<const int*>(wq + L.total) + 12;
    const int kw_a = __builtin_amdgcn_readfirstlane(kwp[0]), kw_b = __builtin_amdgcn_readfirstlane(kwp[1]), kw_c = __builtin_amdgcn_readfirstlane(kwp[2]);
    auto kw = [&](int g) { const int word = g < 4 ? kw_a : g < 8 ? kw_b : kw_c; return (int)(signed char)(word >> (8 * (g & 3))); };
    // a GEMM whose weights carry 2^k hands on activations that carry it too
    auto inherit = [&](int k) {
        lsc += k;
    };
    float h[64];                                  // the current activations (times sc), 64 fp32 values per lane; k-step s of the next GEMM splits values 8s .. 8s+7
    // lmax: this lane's largest |h|; with_unit: the K-vector also holds operands of magnitude ~1 that enter times sc * 2^unit_exp (encoding, view direction)
    auto condition = [&](float lmax, bool with_unit, int unit_exp) {
        float m = fmaxf(lmax, __shfl_xor(lmax, 32));
        if (with_unit) m = fmaxf(m, pow2f(lsc + unit_exp));
        bad = bad || !(m <= F_FINITE);
        const bool need = (m < WIN_LO || m > WIN_HI) && m > 0.0f && m <= F_FINITE;
        if (__ballot(need)) {
            int k = need ? WIN_TARGET - __builtin_amdgcn_frexp_expf(m) : 0;      // m * 2^k in [2^4, 2^5)
            k = min(max(k, max(-SC_LIM - lsc, -126)), min(SC_LIM - lsc, 126));      // (one step moves at most what one fp32 factor holds)
            const float f = pow2f(k);
#pragma unroll
            for (int q = 0; q < 64; ++q) h[q] *= f;
            lsc += k;
        }
    };
    auto report = [&]() {
        if (guard) {
            if (bad) guard[0] = 1;
            if (blockIdx.x == 0 && tid == 0 && (float)wq[L.total] != 0.0f) guard[0] = 1;      // a weight was clamped at pack time
        }
    };
    // accumulators start as the layer's bias vector (fragment order, LDS) times the point's scale
    auto init_acc4 = [&](f32x16 (&acc)[4], const float* vec_h, int k) {        // k: the scale exponent of the weights that will be accumulated onto it
        init_acc_b<4>(acc, vec_h);
        if (k != 0 || __ballot(lsc != 0)) {
            const float f = pow2f(lsc + k);
#pragma unroll
            for (int q = 0; q < 64; ++q) acc[q >> 4][q & 15] *= f;
        }
    };

    BP pe[B_PE_STEPS];                            // positional-encoding pieces (layer 0; stashed until layer 5)
    float bias[64];                               // pts_bias(feat): the multiplicative modulation of every pts_linears layer
    auto act = [&](int s, int j, int part, BP& d) {
        const float* v8 = h + 8 * (s & 7);
        if (part == 0) split_p0(v8, j, d); else if (part == 1) split_p1(v8, j, d); else split_p2(v8, j, d);
    };
    // epilogue of a modulated ReLU layer: h = relu(acc * bias); returns the lane's largest activation
    auto finish = [&](f32x16 (&acc)[4]) {
        float lmax = 0.0f;
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            h[q] = fmaxf(acc[q >> 4][q & 15] * bias[q], 0.0f);
            lmax = fmaxf(lmax, h[q]);
        }
        return lmax;
    };

    // Between two layers a wave prepares everything that does not need the next slab BEFORE the layer barrier - the accumulators (bias vector times the scale they
    // will carry) and the pieces of k-step 0 - so that behind the barrier only the first fragment reads stand between it and the matrix pipe.
    f32x16 acc[4];
    BP first;
    {   // bias = pts_bias(feat); the features are conditioned like every other K-vector (scale 2^kf, local to this GEMM)
        float fmx = 0.0f;
#pragma unroll
        for (int i = 0; i < 24; ++i) fmx = fmaxf(fmx, fabsf(fv[i]));
        fmx = fmaxf(fmx, __shfl_xor(fmx, 32));
        bad = bad || !(fmx <= F_FINITE);
        const bool need = (fmx < WIN_LO || fmx > WIN_HI) && fmx > 0.0f && fmx <= F_FINITE;
        float rf = 1.0f;
        const int kwb = kw(SEG_BIAS);
        const bool scaled = kwb != 0 || __ballot(need);
        int k = 0;
        if (scaled) {
            k = need ? min(max(WIN_TARGET - __builtin_amdgcn_frexp_expf(fmx), -SC_LIM), SC_LIM) : 0;
            const float f = pow2f(k);
            rf = pow2f(-k - kwb);
#pragma unroll
            for (int i = 0; i < 24; ++i) fv[i] *= f;
        }
        BP fb[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) fb[s] = split8h(fv + 8 * s);
        slab_sync();                                                                  // slab 0 and the vectors have landed
        H3_STAMP();
        slab_dma(buf1, wq + L.l1, 2 * ACT_PLANE, wave, lane);                         // layer 1 -> buf1
        init_acc_b<4>(acc, vec + V_BIASG + half * 64);
        if (scaled) {
            const float fa = pow2f(k + kwb);
#pragma unroll
            for (int q = 0; q < 64; ++q) acc[q >> 4][q & 15] *= fa;
        }
        const char* wh = buf0;
        const char* wl = buf0 + feat_plane * 2;
        if (L.fsteps == 1) gemm_h<1, 4>(wh, wl, acc, lane, fb[0], H3_PREP_FROM(fb));
        else if (L.fsteps == 2) gemm_h<2, 4>(wh, wl, acc, lane, fb[0], H3_PREP_FROM(fb));
        else gemm_h<3, 4>(wh, wl, acc, lane, fb[0], H3_PREP_FROM(fb));
        H3_STAMP();
#pragma unroll
        for (int q = 0; q < 64; ++q) bias[q] = acc[q >> 4][q & 15] * rf;
    }
    // the encoding, computed between the first two GEMMs: ~900 VALU instructions that run beside the partner wave's pts_bias / layer-0 matrix work
#pragma unroll
    for (int s = 0; s < B_PE_STEPS; ++s) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = pe_op_hw(8 * s + j, half, px, py, pz);
        trans_fence8(t8);                                                             // (v_sin_f32 -> asm consumer: mlp_b16_dev.h)
        pe[s] = split8h(t8);
    }
    // The encoding pieces wait for layer 5 in the wave's private segment (32 dwords per lane, stored here, loaded once behind layer 4): between the two a wave
    // holds h + accumulators + modulation = 192 registers, and the 32 this frees are the fragments and pieces in flight that keep its matrix stream from stalling.
    // (A placed store / load pair, not a register-allocator spill - the kernel's .vgpr_spill_count is 0.  The array's address is passed through an empty asm so
    // that it stays a memory object; the memory-clobbering waits of slab_sync() between store and load keep the values from being forwarded in registers.)
    u32x4 pe_stash_mem[2 * B_PE_STEPS];
    __attribute__((address_space(5))) u32x4* pe_stash = (__attribute__((address_space(5))) u32x4*)pe_stash_mem;
    asm volatile("" : "+v"(pe_stash));
    {   // layer 0 (same slab)
        init_acc4(acc, vec + V_L0 + half * 64, kw(SEG_L0));
        const char* wh = buf0 + feat_plane * 4;
        gemm_h<B_PE_STEPS, 4>(wh, wh + PE_PLANE * 2, acc, lane, pe[0], H3_PREP_FROM(pe));
        H3_STAMP();
        if (young) { slab_sync(); slab_dma(buf0, wq + L.l1 + 2 * ACT_PLANE, 2 * ACT_PLANE, wave, lane); }      // layer 2 -> buf0
#pragma unroll
        for (int s = 0; s < B_PE_STEPS; ++s) { pe_stash[2 * s] = pe[s].hi; pe_stash[2 * s + 1] = pe[s].lo; }
        inherit(kw(SEG_L0));
        condition(finish(acc), false, 0);
        init_acc4(acc, vec + V_L0 + 128 + half * 64, kw(SEG_L1));
        first = split8h(h);
        H3_STAMP();
        if (!young) { slab_sync(); slab_dma(buf0, wq + L.l1 + 2 * ACT_PLANE, 2 * ACT_PLANE, wave, lane); }
        H3_STAMP();
    }
    // layers 1..4: slabs alternate buf1, buf0, buf1, buf0; behind layer l's GEMM its buffer takes the slab after next (layer l + 2; layer 5's encoding part;
    // layer 5's h part).  Layer 4 is peeled off the rolled loop: behind its GEMM the encoding comes back from the stash (inside the loop its 32 registers would be
    // live through every iteration).
    auto hidden_layer = [&](int layer, auto last) {
        constexpr bool LAST = decltype(last)::value;
        char* cur = (layer & 1) ? buf1 : buf0;
        auto refill = [&]() {
            slab_sync();
            if (LAST) slab_dma(cur, wq + L.l5b, 2 * ACT_PLANE, wave, lane);
            else if (layer == 3) slab_dma(cur, wq + L.l5a, 2 * PE_PLANE, wave, lane);
            else slab_dma(cur, wq + L.l1 + (size_t)(layer + 1) * 2 * ACT_PLANE, 2 * ACT_PLANE, wave, lane);
        };
        gemm_h<B_ACT_STEPS, 4>(cur, cur + ACT_PLANE * 2, acc, lane, first, act);
        H3_STAMP();
        if (LAST) {                                                                   // the encoding comes back while the epilogue runs
#pragma unroll
            for (int s = 0; s < B_PE_STEPS; ++s) { pe[s].hi = pe_stash[2 * s]; pe[s].lo = pe_stash[2 * s + 1]; }
        }
        if (young) refill();
        inherit(kw(SEG_L1 + layer - 1));
        condition(finish(acc), LAST, kw(SEG_L5B) - kw(SEG_L5A));                     // layer 5's K-vector = [encoding | h4]
        init_acc4(acc, vec + V_L0 + 128 * (layer + 1) + half * 64, LAST ? kw(SEG_L5B) : kw(SEG_L1 + layer));
        if (!LAST) first = split8h(h);
        H3_STAMP();
        if (!young) refill();
        H3_STAMP();
    };
#pragma unroll 1
    for (int layer = 1; layer <= 3; ++layer) hidden_layer(layer, std::false_type());
    hidden_layer(4, std::true_type());
    float sigma;
    {   // layer 5 on cat([pts, h4]): L5a in buf1, L5b in buf0
        // the encoding enters times sc * 2^(kw5b - kw5a) - its own weights carry 2^kw5a, the accumulators 2^kw5b: exact on the fp16 pieces while that exponent
        // is fp16's (condition(.., true, ..) keeps it inside the window or below it; far below, the encoding's share of the sum is below fp32's resolution anyway)
        const int kw5 = kw(SEG_L5B), pe_exp = lsc + kw5 - kw(SEG_L5A);
        if (__ballot(pe_exp != 0)) {
            const _Float16 s16 = (_Float16)pow2f(min(max(pe_exp, -24), 15));
#pragma unroll
            for (int s = 0; s < B_PE_STEPS; ++s) {
                pe[s].hi = __builtin_bit_cast(u32x4, as_h8(pe[s].hi) * s16);
                pe[s].lo = __builtin_bit_cast(u32x4, as_h8(pe[s].lo) * s16);
            }
        }
        gemm_h<B_PE_STEPS, 4>(buf1, buf1 + PE_PLANE * 2, acc, lane, pe[0], H3_PREP_FROM(pe));
        first = split8h(h);                                                           // k-step 0 of the h part
        H3_STAMP();
        slab_sync();                                                                  // no epilogue between the two parts: the same place for both waves
        H3_STAMP();
        if (!ALPHA_ONLY) slab_dma(buf1, wq + L.feat, 2 * ACT_PLANE, wave, lane);
        gemm_h<B_ACT_STEPS, 4>(buf0, buf0 + ACT_PLANE * 2, acc, lane, first, act);
        H3_STAMP();
        if (!ALPHA_ONLY && young) { slab_sync(); slab_dma(buf0, wq + L.views, 2 * VIEW_PLANE, wave, lane); }
        // alpha_linear on the fp32 activations (before they are split)
        inherit(kw5);
        const float lmax = finish(acc);
        const float* wa = vec + V_WA + half * 64;
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 64; ++q) part = fmaf(wa[q], h[q], part);
        part += __shfl_xor(part, 32);
        sigma = fmaxf(__builtin_amdgcn_ldexpf(part, -lsc) + vec[V_BA], 0.0f);
        if (ALPHA_ONLY) {
            float m = fmaxf(lmax, __shfl_xor(lmax, 32));
            bad = bad || !(m <= F_FINITE);
        } else {
            condition(lmax, false, 0);
            init_acc4(acc, vec + V_FEAT + half * 64, kw(SEG_FEAT));
            first = split8h(h);
        }
        H3_STAMP();
        if (!ALPHA_ONLY && !young) { slab_sync(); slab_dma(buf0, wq + L.views, 2 * VIEW_PLANE, wave, lane); }
        H3_STAMP();
    }
    // (the point index is recomputed from an opaque copy of the thread index where the results are stored: kept from the prologue it costs two registers for the
    // whole kernel - with 256 in use, the ones that spill)
    auto point_again = [&]() {
        int t2 = threadIdx.x;
        asm volatile("" : "+v"(t2));
        return ((int64_t)blockIdx.x * H3_WAVES + (t2 >> 6)) * 32 + (t2 & 31);
    };
    if (ALPHA_ONLY) {
        const int64_t q_raw = point_again();
        if (q_raw < P && half == 0) raw[q_raw] = sigma;
        report();
        return;
    }
    f32x16 av[2];
    {   // feature_linear (buf1, no activation)
        gemm_h<B_ACT_STEPS, 4>(buf1, buf1 + ACT_PLANE * 2, acc, lane, first, act);
        H3_STAMP();
        if (young) slab_sync();                                                       // (the views slab has landed)
        inherit(kw(SEG_FEAT));
        float lmax = 0.0f;
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            h[q] = acc[q >> 4][q & 15];
            lmax = fmaxf(lmax, fabsf(h[q]));
        }
        condition(lmax, true, kw(SEG_VF) - kw(SEG_VD));                               // views' K-vector = [feature | view direction]
    }
    report();
    const int64_t q_raw = point_again();
    {   // views_linears[0] + rgb head
        float dl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int kwv = kw(SEG_VF);
        const float dsc = pow2f(lsc + kwv - kw(SEG_VD));       // the direction's own weights carry 2^kw(SEG_VD), the accumulators 2^kwv
        dl[0] = dir0 * dsc;
        dl[1] = dir1 * dsc;
        BP d8[1];
        d8[0] = split8h(dl);
        init_acc_b<2>(av, vec + V_VIEWS + half * 32);
        if (kwv != 0 || __ballot(lsc != 0)) {
            const float f = pow2f(lsc + kwv);
#pragma unroll
            for (int q = 0; q < 32; ++q) av[q >> 4][q & 15] *= f;
        }
        first = split8h(h);
        H3_STAMP();
        if (!young) slab_sync();
        H3_STAMP();
        gemm_h<B_VIEW_STEPS, 2>(buf0, buf0 + VIEW_PLANE * 2, av, lane, first,
                                [&](int s, int j, int part, BP& d) { if (s < 8) act(s, j, part, d); else if (j == 0 && part == 0) d = d8[0]; });
        H3_STAMP();
        inherit(kwv);
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* wr = vec + V_WR + c * 64 + half * 32;
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 32; ++q) part = fmaf(wr[q], fmaxf(av[q >> 4][q & 15], 0.0f), part);
            part += __shfl_xor(part, 32);
            rgb[c] = 1.0f / (1.0f + expf(-(__builtin_amdgcn_ldexpf(part, -lsc) + vec[V_BR + c])));
        }
        if (q_raw < P && half == 0) *reinterpret_cast<f32x4*>(raw + q_raw * 4) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
        H3_STAMP();
    }
}

}  // namespace

// internal entries behind mvsnerf_mlp_{packed_split_elems, pack_split, fwd_split}(n_split = MVSNERF_SPLIT_FP16) in mlp_bf16.hip
size_t mvs_mlp_f16x3_elems(int F) { return layout_h(F).total + H3_TAIL; }

int mvs_mlp_f16x3_pack(const float* const w[11], int F, void* packed, hipStream_t st)
{
    PackBArgs a;
    for (int i = 0; i < 11; ++i) { if (!w[i]) return MVSNERF_EINVAL; a.w[i] = w[i]; }
    a.F = F;
    _Float16* tail = reinterpret_cast<_Float16*>(packed) + layout_h(F).total;
    hipError_t e = hipMemsetAsync(tail, 0, H3_TAIL * sizeof(_Float16), st);      // status: all weights finite; maxima 0
    if (e != hipSuccess) return (int)e;
    mlp_h3_wmax_kernel<<<64, 256, 0, st>>>(a, reinterpret_cast<unsigned*>(tail) + 1);
    MVS_LAUNCH_CHECK();
    mlp_pack_h3_kernel<<<64, 256, 0, st>>>(a, reinterpret_cast<_Float16*>(packed));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

int mvs_mlp_f16x3_fwd(const void* packed_h, const float* packed_f32, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                      const float* dirs, int dirs_stride, int64_t P, int S, int alpha_only, float* raw, hipStream_t st, int* guard)
{
    static unsigned long long cap_a = 0, cap_b = 0;         // per-device bit masks (common.h)
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_f16x3_kernel<false>), H3_LDS_BYTES, &cap_a)) return rc_;
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_f16x3_kernel<true>), H3_LDS_BYTES, &cap_b)) return rc_;
    const _Float16* wq = reinterpret_cast<const _Float16*>(packed_h);
    const unsigned grid = mvs_cdiv(P, 32 * H3_WAVES);
    if (alpha_only)
        mlp_fwd_f16x3_kernel<true><<<grid, H3_THREADS, H3_LDS_BYTES, st>>>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, guard);
    else
        mlp_fwd_f16x3_kernel<false><<<grid, H3_THREADS, H3_LDS_BYTES, st>>>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, guard);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
