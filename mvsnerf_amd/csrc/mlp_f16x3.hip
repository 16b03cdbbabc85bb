// "fp16x3": fp32-grade results of the fused Embedder + Renderer_ours forward (models.py:145-222) from THREE v_mfma_f32_32x32x16_f16 per
// product.  Inference only.  What a no-grad rendering() runs by default (ops.MLP_PRECISION = "auto") as a GUARDED sequence: the kernel reports
// every value that left fp16's range through the guard word and the fp32-MFMA kernel of mlp.hip, enqueued right behind it and predicated on that
// word, recomputes the batch (include/mvsnerf_hip.h, "guarded 16-bit sequences").  ops.set_mlp_precision("fp16x3") is the unguarded kernel alone
// (saturating); the headline of bench.py stays on the fp32-MFMA kernel.
//
// Every fp32 operand is written as the sum of two fp16 pieces, both rounded to nearest:
//     a = a0 + a1 + r,   a0 = fp16(a),  a1 = fp16(a - a0),   |r| <= 2^-22 |a|      (fp16 carries 11 significant bits; a - a0 is exact)
//     a * w  ~=  a0*w0 + a0*w1 + a1*w0                                              (dropped: a1*w1 <= 2^-22 |a*w|)
// The piece products are exact in fp32 and the matrix core accumulates them in fp32, so a product is off by ~3 * 2^-22 - the same order as
// the roundings of an fp32 kernel (scratch/r3/f16x3_numerics.py on the shipped weights: sigma 2.4e-6 from the float64 result, the torch
// fp32 path 2.9e-6, the two-piece BF16 split 1.2e-4).  The three-piece bf16 split of mlp_bf16.hip ("bf16x6") needs six instructions of the
// same rate for that.  Price: fp16's range.  The pieces of operands below 2^-3 are fp16 subnormals - gfx950's matrix cores take them as
// they are (tests/test_gpu_raymarch.py would show a flush as a 1e-3 error) - and an operand above 65504 saturates (the
// activations of the shipped network stay below 200); the bf16 splits have fp32's range and remain for networks that need it.
// LOG2_SA / LOG2_SW pre-scale activations / weights by exact powers of two (undone in the fp32 epilogues) should a network need another window.
//
// Structure: the transposed-layer scheme of mlp_layout.h - 32 points per wave, the C/D fragment of one layer IS the B operand of the
// next, activations never leave the register file (fp32; split into their fp16 pieces per k-step inside the GEMM loop) - with EIGHT waves
// (256 points) per workgroup sharing each layer's weights: a layer is one 64 KB slab (hi plane | lo plane), two slabs alternate in LDS,
// the next one arrives by LDS-DMA while the current one is multiplied (one barrier per layer), 132 KB of LDS = one workgroup = two
// waves per SIMD.  Per k-step and output block a wave reads one hi and one lo weight fragment (ds_read_b128) for three MFMAs.
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

constexpr int LOG2_SA = 0, LOG2_SW = 0;                              // operand scales 2^LOG2_SA (activations), 2^LOG2_SW (weights)
constexpr float SA = (float)(1 << LOG2_SA), SW = (float)(1 << LOG2_SW);
constexpr float H_MAX = 65504.0f;                                    // largest finite fp16
constexpr int H3_TAIL = 8;                                           // status elements behind the packed weights (16 bytes); [0] != 0: a weight was clamped

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
    L.total = o;                                                    // followed by H3_TAIL status elements (pack: "a weight was clamped")
    return L;
}

// one segment: hi plane at dst[0 .. n), lo plane at dst[n .. 2n), n = steps * nb * 512; element order of a plane = pack_b_segment's
__device__ inline void pack_h_planes(_Float16* __restrict__ dst, const float* __restrict__ W, int ld, int col_off, int kmap,
                                     int steps, int nb, int F, int tid, int nthreads, _Float16* __restrict__ clamped)
{
    const int n = steps * nb * 64 * 8;
    for (int i = tid; i < n; i += nthreads) {
        const int j = i & 7, lane = (i >> 3) & 63, rest = i >> 9;          // rest = s*nb + b
        const int b = rest % nb, s = rest / nb;
        const int col = b_col(kmap, 8 * s + j, lane >> 5, F);
        const int row = b * 32 + (lane & 31);
        float w = col < 0 ? 0.0f : W[(size_t)row * ld + col_off + col] * SW;
        if (!(fabsf(w) <= H_MAX)) *clamped = (_Float16)1.0f;             // outside fp16's range (or NaN): the guarded sequence then always takes the fp32 kernel
        w = fminf(fmaxf(w, -H_MAX), H_MAX);
        const _Float16 hi = (_Float16)w;
        dst[i] = hi;
        dst[n + i] = (_Float16)(w - (float)hi);
    }
}

__global__ __launch_bounds__(256) void mlp_pack_h3_kernel(PackBArgs a, _Float16* __restrict__ packed)
{
    const LayoutH L = layout_h(a.F);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const size_t act = 2 * b_seg(B_ACT_STEPS, 4);
    pack_h_planes(packed + L.s0, a.w[6], a.F, 0, K_FEAT, L.fsteps, 4, a.F, tid, nt, packed + L.total);
    pack_h_planes(packed + L.s0 + 2 * b_seg(L.fsteps, 4), a.w[0], PE_DIM, 0, K_PE, B_PE_STEPS, 4, a.F, tid, nt, packed + L.total);
    for (int l = 1; l <= 4; ++l) pack_h_planes(packed + L.l1 + (l - 1) * act, a.w[l], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt, packed + L.total);
    pack_h_planes(packed + L.l5a, a.w[5], WIDTH + PE_DIM, 0, K_PE, B_PE_STEPS, 4, a.F, tid, nt, packed + L.total);
    pack_h_planes(packed + L.l5b, a.w[5], WIDTH + PE_DIM, PE_DIM, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt, packed + L.total);
    pack_h_planes(packed + L.feat, a.w[7], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt, packed + L.total);
    pack_h_planes(packed + L.views, a.w[9], WIDTH + 3, 0, K_VIEWS, B_VIEW_STEPS, 2, a.F, tid, nt, packed + L.total);
}

// ------------------------------------------------------------------------------------------ kernel
struct HL { f16x8 hi, lo; };

// the two fp16 pieces of 8 fp32 values (v_cvt_pk_f16_f32 rounds to nearest; the remainder a - a0 is exact in fp32)
__device__ __forceinline__ HL split8h(const float* v)
{
    HL r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)v[j];
        r.hi[j] = h;
        r.lo[j] = (_Float16)(v[j] - (float)h);
    }
    return r;
}

// DEV probe (scratch/r3/build_variant.sh; never defined in the product build): -DH3_NO_DMA fetches no weight slab at all (garbage results) - the
// floor of MFMA + VALU + barriers: 77.4 us against 82.7 us with the slabs, i.e. the L2 -> LDS stream (256 MB per launch) is NOT what bounds the
// kernel.  Measured without effect and removed: per-workgroup rotation of the piece order, non-temporal DMA loads.
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

// acc[nb] += W[block nb] * act over STEPS k-steps of 16: per step and block pair two hi + two lo weight fragments, six MFMAs on two
// independent accumulators, smallest piece products first.  (Groups of four blocks - 12 MFMAs between fragment loads - measured the same, 85.7 vs
// 84.7 us, with more spilled registers; the two waves of a SIMD taking turns in s_setprio per k-step: 87.6 vs 84.9 us.)
template <int STEPS, int NBLK, typename BFN>
__device__ __forceinline__ void gemm_h(const char* __restrict__ w_hi, const char* __restrict__ w_lo, f32x16 (&acc)[NBLK], int lane, BFN bfn)
{
    static_assert(NBLK % 2 == 0, "output blocks are taken in pairs");
    // one lane base per plane; every fragment of the segment is then an immediate offset of the ds_read (<= 65520 from the hi-plane base)
    const f16x8* __restrict__ fh = reinterpret_cast<const f16x8*>(w_hi) + lane;
    const f16x8* __restrict__ fl = reinterpret_cast<const f16x8*>(w_lo) + lane;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const HL b = bfn(s);
#pragma unroll
        for (int nb = 0; nb < NBLK; nb += 2) {
            f16x8 ah[2], al[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                ah[e] = fh[(s * NBLK + nb + e) * 64];
                al[e] = fl[(s * NBLK + nb + e) * 64];
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) acc[nb + e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[e], b.hi, acc[nb + e], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 2; ++e) acc[nb + e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[e], b.lo, acc[nb + e], 0, 0, 0);
#pragma unroll
            for (int e = 0; e < 2; ++e) acc[nb + e] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[e], b.hi, acc[nb + e], 0, 0, 0);
        }
    }
}

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
#ifdef H3_CENSUS      // DEV probe: shader-clock stamps of this wave's phases, 32 per tile, behind the results (scratch/r3/h3_census.py allocates them)
    unsigned* cen = reinterpret_cast<unsigned*>(raw + P * (ALPHA_ONLY ? 1 : 4)) + ((int64_t)blockIdx.x * H3_WAVES + wave) * 32;
    int cen_i = 0;
#define H3_STAMP() do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (lane == 0 && cen_i < 32) cen[cen_i] = (unsigned)t__; ++cen_i; } while (0)
#else
#define H3_STAMP() do {} while (0)
#endif
    H3_STAMP();                                   // 0: wave start

    // slab 0 = pts_bias weights + layer 0 (contiguous in the packed buffer)
    slab_dma(buf0, wq + L.s0, L.l1 - L.s0, wave, lane);
    // bias vectors enter the accumulators, which hold SA*SW times the true sums; the head vectors (from V_WA on) stay as they are
    for (int i = tid; i < V_TOTAL; i += H3_THREADS) vec[i] = packed_f32[LF.vec + i] * (i < V_WA ? SA * SW : 1.0f);
    const float px = ndc[p * ndc_stride + 0], py = ndc[p * ndc_stride + 1], pz = ndc[p * ndc_stride + 2];
    float fv[24];                                 // F/2 <= 20 feature operands of this lane half
    {
        const float* fp = feat + p * feat_stride + half * (F / 2);
#pragma unroll
        for (int i = 0; i < 24; ++i) fv[i] = i < F / 2 ? fp[i] * SA : 0.0f;
    }
    // guard: the largest magnitude this lane hands to an fp16 split (features; layer outputs before their clamp).  Positions and view directions
    // are bounded by construction.  Reported once, at the end (guard != NULL: the caller enqueues the fp32 kernel behind this one, predicated on it).
    float gmax = 0.0f;
#pragma unroll
    for (int i = 0; i < 24; ++i) gmax = fmaxf(gmax, fabsf(fv[i]));
    // ... and the other end of fp16: a second piece below 2^-14 is a subnormal (absolute resolution 2^-24), so an operand x carries an absolute error of up
    // to 2^-25 whatever its size - fp32 grade only while the layer's inputs are not ALL small.  `tiny` is set when the largest activation a layer hands on,
    // taken over the wave's 32 points, is non-zero and below 2^-7 (the layer's products then carry > 2^-18 of its scale): reported like an overflow, the
    // fp32 kernel takes the batch (a network whose hidden activations are 1e-4 of the shipped one's: tests/test_gpu_fp16x3.py).
    bool tiny = false;
    auto layer_scale = [&](float lmax) {                          // lmax: the largest |activation| among this lane's 64 values of the layer
        gmax = fmaxf(gmax, lmax);
        // wave-uniform, two compares and no cross-lane traffic: no lane reaches 2^-7, some lane is above zero (a wave-level max by six shuffles per
        // layer cost 3.8 % of the kernel: they sit in the epilogue, the part of a layer nothing overlaps)
        tiny = tiny || (__ballot(lmax >= SA * 0.0078125f) == 0 && __ballot(lmax > 0.0f) != 0);
    };
    auto report = [&]() {
        if (guard) {
            if (gmax > H_MAX || tiny) guard[0] = 1;
            if (blockIdx.x == 0 && tid == 0 && (float)wq[L.total] != 0.0f) guard[0] = 1;      // a weight was clamped at pack time
        }
    };
    HL pe[B_PE_STEPS];                            // positional-encoding operands (reused by layer 5)
#pragma unroll
    for (int s = 0; s < B_PE_STEPS; ++s) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = pe_op(8 * s + j, half, px, py, pz) * SA;
        pe[s] = split8h(t8);
    }
    float bias[64];                               // pts_bias(feat) / SW: relu(acc * bias) is then SA times the true activation
    // the current activations, 64 fp32 values per lane; k-step s of the next GEMM takes values q = 8s .. 8s+7 as its B operand and splits them
    // there, between the MFMAs (splitting once in the layer epilogue instead - 64 registers of pieces - measured 86.8 against 84.7 us)
    float h[64];
    auto act = [&](int s) { return split8h(h + 8 * (s & 7)); };
    auto put = [&](int s, const float* v8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) h[8 * s + j] = v8[j];
    };
    // epilogue of a modulated ReLU layer; the clamp only matters where fp16 would overflow
    auto finish = [&](f32x16 (&acc)[4]) {
        float lmax = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = 8 * s + j;
                const float x = acc[q >> 4][q & 15] * bias[q];
                lmax = fmaxf(lmax, x);
                v8[j] = __builtin_amdgcn_fmed3f(x, 0.0f, H_MAX);
            }
            put(s, v8);
        }
        layer_scale(lmax);
    };

    slab_sync();
    H3_STAMP();
    slab_dma(buf1, wq + L.l1, 2 * ACT_PLANE, wave, lane);
    {   // bias = pts_bias(feat)
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_BIASG + half * 64);
        auto fb = [&](int s) { return split8h(fv + 8 * s); };
        const char* wh = buf0;
        const char* wl = buf0 + feat_plane * 2;
        if (L.fsteps == 1) gemm_h<1, 4>(wh, wl, acc, lane, fb);
        else if (L.fsteps == 2) gemm_h<2, 4>(wh, wl, acc, lane, fb);
        else gemm_h<3, 4>(wh, wl, acc, lane, fb);
        H3_STAMP();
#pragma unroll
        for (int q = 0; q < 64; ++q) bias[q] = acc[q >> 4][q & 15] * (1.0f / (SA * SW * SW));
    }
    {   // layer 0
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + half * 64);
        const char* wh = buf0 + feat_plane * 4;
        gemm_h<B_PE_STEPS, 4>(wh, wh + PE_PLANE * 2, acc, lane, [&](int s) { return pe[s]; });
        H3_STAMP();
        finish(acc);
        H3_STAMP();
    }
    // layers 1..4: slabs alternate buf1, buf0, buf1, buf0
#pragma unroll 1
    for (int layer = 1; layer <= 4; ++layer) {
        char* cur = (layer & 1) ? buf1 : buf0;
        char* nxt = (layer & 1) ? buf0 : buf1;
        slab_sync();
        H3_STAMP();
        if (layer < 4) slab_dma(nxt, wq + L.l1 + (size_t)layer * 2 * ACT_PLANE, 2 * ACT_PLANE, wave, lane);
        else slab_dma(nxt, wq + L.l5a, 2 * PE_PLANE, wave, lane);                    // after layer 4 (in buf0): L5a -> buf1
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + 128 * layer + half * 64);
        gemm_h<B_ACT_STEPS, 4>(cur, cur + ACT_PLANE * 2, acc, lane, act);
        H3_STAMP();
        finish(acc);
        H3_STAMP();
    }
    float sigma;
    {   // layer 5 on cat([pts, h4]): L5a in buf1, L5b -> buf0
        f32x16 acc[4];
        slab_sync();
        H3_STAMP();
        slab_dma(buf0, wq + L.l5b, 2 * ACT_PLANE, wave, lane);
        init_acc_b<4>(acc, vec + V_L0 + 128 * 5 + half * 64);
        gemm_h<B_PE_STEPS, 4>(buf1, buf1 + PE_PLANE * 2, acc, lane, [&](int s) { return pe[s]; });
        H3_STAMP();
        slab_sync();
        H3_STAMP();
        if (!ALPHA_ONLY) slab_dma(buf1, wq + L.feat, 2 * ACT_PLANE, wave, lane);
        gemm_h<B_ACT_STEPS, 4>(buf0, buf0 + ACT_PLANE * 2, acc, lane, act);
        H3_STAMP();
        // alpha_linear on the fp32 activations (before they are split), then the split for feature_linear
        const float* wa = vec + V_WA + half * 64;
        float part = 0.0f, lmax = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = 8 * s + j;
                const float x = acc[q >> 4][q & 15] * bias[q];
                lmax = fmaxf(lmax, x);
                v8[j] = __builtin_amdgcn_fmed3f(x, 0.0f, H_MAX);
                part = fmaf(wa[q], v8[j], part);
            }
            if (!ALPHA_ONLY) put(s, v8);
        }
        if (ALPHA_ONLY) gmax = fmaxf(gmax, lmax); else layer_scale(lmax);      // (the sigma head reads these values in fp32: only feature_linear splits them)
        part += __shfl_xor(part, 32);
        sigma = fmaxf(fmaf(part, 1.0f / SA, vec[V_BA]), 0.0f);
        H3_STAMP();
    }
    if (ALPHA_ONLY) {
        if (live && half == 0) raw[p_raw] = sigma;
        report();
        return;
    }
    {   // feature_linear (buf1, no activation), then views -> buf0
        f32x16 acc[4];
        slab_sync();
        H3_STAMP();
        slab_dma(buf0, wq + L.views, 2 * VIEW_PLANE, wave, lane);
        init_acc_b<4>(acc, vec + V_FEAT + half * 64);
        gemm_h<B_ACT_STEPS, 4>(buf1, buf1 + ACT_PLANE * 2, acc, lane, act);
        H3_STAMP();
        float lmax = 0.0f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = 8 * s + j;
                const float x = acc[q >> 4][q & 15] * (1.0f / SW);
                lmax = fmaxf(lmax, fabsf(x));
                v8[j] = __builtin_amdgcn_fmed3f(x, -H_MAX, H_MAX);
            }
            put(s, v8);
        }
        layer_scale(lmax);
    }
    report();
    {   // views_linears[0] + rgb head
        const int64_t ray = p / S;
        float dl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        dl[0] = (half ? dirs[ray * dirs_stride + 1] : dirs[ray * dirs_stride + 0]) * SA;
        dl[1] = half ? 0.0f : dirs[ray * dirs_stride + 2] * SA;
        const HL d8 = split8h(dl);
        f32x16 acc[2];
        slab_sync();
        H3_STAMP();
        init_acc_b<2>(acc, vec + V_VIEWS + half * 32);
        gemm_h<B_VIEW_STEPS, 2>(buf0, buf0 + VIEW_PLANE * 2, acc, lane, [&](int s) { return s < 8 ? act(s) : d8; });
        H3_STAMP();
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* wr = vec + V_WR + c * 64 + half * 32;
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 32; ++q) part = fmaf(wr[q], fmaxf(acc[q >> 4][q & 15], 0.0f), part);
            part += __shfl_xor(part, 32);
            rgb[c] = 1.0f / (1.0f + expf(-fmaf(part, 1.0f / (SA * SW), vec[V_BR + c])));
        }
        if (live && half == 0) *reinterpret_cast<f32x4*>(raw + p_raw * 4) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
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
    hipError_t e = hipMemsetAsync(reinterpret_cast<_Float16*>(packed) + layout_h(F).total, 0, H3_TAIL * sizeof(_Float16), st);      // status: nothing clamped
    if (e != hipSuccess) return (int)e;
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
