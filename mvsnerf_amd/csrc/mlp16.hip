// DEV BUILD ONLY (`make dev`, -DMVSNERF_DEV_KNOBS): a measured-and-dropped schedule, not compiled into libmvsnerf_hip.so.
#ifdef MVSNERF_DEV_KNOBS
// 16-point-tile variant of the fused Embedder + Renderer_ours forward (inference): same network, same fp32 MFMA arithmetic as
// mlp.hip, on v_mfma_f32_16x16x4_f32 instead of v_mfma_f32_32x32x2_f32.
//
// Why it exists: a wave that owns 32 points holds 192 fp32 values per lane (layer input, layer output, per-point bias), which
// caps the CU at two waves per SIMD; with 16 points per wave the same three arrays are 96 values per lane and up to four waves
// fit.  Measured (mvsnerf_tune "mlp_variant" = 4): 0.258-0.262 ms per 1024x128 batch whether it runs at two or at four waves per
// SIMD, i.e. the same ~80 % matrix-pipe utilisation as the 32-point kernel (0.251-0.262 ms): occupancy is not what bounds the
// 32-point kernel.  Kept as a selectable variant (same results to fp32 rounding, half the registers per wave).
//
// Layout (the same register-chaining idea): the C/D fragment of a 16x16 block puts, in lane (n = lane&15, g = lane>>4), the
// rows 4g+r, r = 0..3, of column (point) n.  With 8 row blocks per 128-wide layer a lane holds outputs
//     n16(q, g) = 16*(q>>2) + 4*g + (q&3),   q = 4*block + r = 0..31
// and the B operand of k-step t (4 contraction indices, one per lane group g) wants act[k_t(g)][n] in the same lane: choosing
// k_t(g) = n16(t, g) makes register t of a layer's output the B operand of step t of the next layer.  Weights are re-ordered
// once (pack16) so that the A fragments of k-steps 4j..4j+3 of block b are one float4 per lane, lane-linear in LDS.
// A workgroup = 8 waves = 128 points shares the weight slabs (LDS-DMA double buffering as in mlp_fwd_pipe_kernel); 6 waves per
// workgroup load the four SIMDs unevenly (two of them carry twice the MFMA work) and ran at 0.389 ms.
#include "common.h"
#include "lds_dma.h"
#include "mlp_layout.h"

using namespace mlp;

namespace mlp16 {

constexpr int PE_ST = 16;        // 64 padded embedding inputs / 4
constexpr int ACT_ST = 32;       // 128 / 4
constexpr int VIEW_ST = 36;      // 32 (feature) + 1 (dir xyz + pad), rounded up to a multiple of 4

__host__ __device__ inline int n16(int q, int g) { return 16 * (q >> 2) + 4 * g + (q & 3); }

// which input column feeds (k-step t, lane group g); -1 = zero padding
__host__ __device__ inline int kcol(int kmap, int t, int g, int F)
{
    switch (kmap) {
    case K_PE:    // t = 0: (x, y, z, pad); t >= 1: frequency/coordinate pair k = 2(t-1) + (g>>1), sin for even g, cos for odd g
        if (t == 0) return g < 3 ? g : -1;
        if (t >= PE_ST) return -1;
        return ((g & 1) ? 33 : 3) + 2 * (t - 1) + (g >> 1);
    case K_FEAT:  { const int c = 4 * t + g; return c < F ? c : -1; }
    case K_ACT:   return t < ACT_ST ? n16(t, g) : -1;
    case K_VIEWS: return t < ACT_ST ? n16(t, g) : (t == ACT_ST && g < 3 ? WIDTH + g : -1);
    }
    return -1;
}

__host__ __device__ inline int feat_st(int F) { return (((F + 3) / 4) + 3) & ~3; }
__host__ __device__ inline size_t seg(int steps, int nb) { return (size_t)steps * nb * 64; }

// vector block, fragment order: [4 groups][32 registers] per 128-wide vector
constexpr int W_BIASG = 0;
constexpr int W_L0 = 128;                 // + 128*i
constexpr int W_FEAT = 128 * 7;
constexpr int W_VIEWS = 128 * 8;          // [4][16]
constexpr int W_WA = W_VIEWS + 64;        // [4][32] alpha_linear weight
constexpr int W_BA = W_WA + 128;          // alpha bias (+3 pad)
constexpr int W_WR = W_BA + 4;            // [3][4][16] rgb_linear weight
constexpr int W_BR = W_WR + 192;          // rgb bias (+1 pad)
constexpr int W_TOTAL = W_BR + 4;

struct Layout16 { size_t biasw, l0, l1, l5a, l5b, feat, views, vec, total; int fst; };
__host__ __device__ inline Layout16 layout16(int F)
{
    Layout16 L;
    L.fst = feat_st(F);
    size_t o = 0;
    L.biasw = o; o += seg(L.fst, 8);
    L.l0 = o;    o += seg(PE_ST, 8);
    L.l1 = o;    o += 4 * seg(ACT_ST, 8);          // l1..l4 contiguous
    L.l5a = o;   o += seg(PE_ST, 8);
    L.l5b = o;   o += seg(ACT_ST, 8);
    L.feat = o;  o += seg(ACT_ST, 8);
    L.views = o; o += seg(VIEW_ST, 4);
    L.vec = o;   o += W_TOTAL;
    L.total = (o + 3) & ~(size_t)3;
    return L;
}

struct Pack16Args { const float* w[11]; const float* b[11]; int F; };

__device__ inline void pack_seg(float* __restrict__ dst, const float* __restrict__ W, int ld, int col_off, int kmap, int steps, int nb,
                                int F, int tid, int nt)
{
    const int total = steps * nb * 64;
    for (int i = tid; i < total; i += nt) {
        const int j = i & 3, lane = (i >> 2) & 63, rest = i >> 8;            // rest = t4*nb + b
        const int b = rest % nb, t = (rest / nb) * 4 + j;
        const int col = kcol(kmap, t, lane >> 4, F);
        const int row = b * 16 + (lane & 15);
        dst[i] = col < 0 ? 0.0f : W[(size_t)row * ld + col_off + col];
    }
}

__global__ __launch_bounds__(256) void pack16_kernel(Pack16Args a, float* __restrict__ packed)
{
    const Layout16 L = layout16(a.F);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    pack_seg(packed + L.biasw, a.w[6], a.F, 0, K_FEAT, L.fst, 8, a.F, tid, nt);
    pack_seg(packed + L.l0, a.w[0], PE_DIM, 0, K_PE, PE_ST, 8, a.F, tid, nt);
    for (int l = 1; l <= 4; ++l) pack_seg(packed + L.l1 + (size_t)(l - 1) * seg(ACT_ST, 8), a.w[l], WIDTH, 0, K_ACT, ACT_ST, 8, a.F, tid, nt);
    pack_seg(packed + L.l5a, a.w[5], WIDTH + PE_DIM, 0, K_PE, PE_ST, 8, a.F, tid, nt);
    pack_seg(packed + L.l5b, a.w[5], WIDTH + PE_DIM, PE_DIM, K_ACT, ACT_ST, 8, a.F, tid, nt);
    pack_seg(packed + L.feat, a.w[7], WIDTH, 0, K_ACT, ACT_ST, 8, a.F, tid, nt);
    pack_seg(packed + L.views, a.w[9], WIDTH + 3, 0, K_VIEWS, VIEW_ST, 4, a.F, tid, nt);
    float* v = packed + L.vec;
    for (int i = tid; i < W_TOTAL; i += nt) {
        float x = 0.0f;
        if (i < W_VIEWS) {                       // eight [4][32] bias vectors
            const int which = i >> 7, g = (i >> 5) & 3, q = i & 31;
            const float* src = which == 0 ? a.b[6] : which <= 6 ? a.b[which - 1] : a.b[7];
            x = src[n16(q, g)];
        } else if (i < W_WA) {                   // views bias [4][16]
            const int k = i - W_VIEWS;
            x = a.b[9][n16(k & 15, k >> 4)];
        } else if (i < W_BA) {                   // alpha weight [4][32]
            const int k = i - W_WA;
            x = a.w[8][n16(k & 31, k >> 5)];
        } else if (i < W_WR) {
            x = (i == W_BA) ? a.b[8][0] : 0.0f;
        } else if (i < W_BR) {                   // rgb weight [3][4][16]
            const int k = i - W_WR, c = k >> 6, g = (k >> 4) & 3, q = k & 15;
            x = a.w[10][c * 64 + n16(q, g)];
        } else {
            const int c = i - W_BR;
            x = c < 3 ? a.b[10][c] : 0.0f;
        }
        v[i] = x;
    }
}

// ------------------------------------------------------------------------------------------ kernel
constexpr int WAVES = 8;                                   // 128 points per workgroup: two waves per SIMD, so two workgroups per CU load the four SIMDs evenly
constexpr int SLAB_F = 9216;                               // 36 KB: views = 36 k-steps x 4 blocks x 64 lanes; half a 128-wide layer = 8192
constexpr int LDS_F = 2 * SLAB_F + W_TOTAL;
constexpr int HALF = (ACT_ST / 2) * 8 * 64;               // floats of half a 128x128 layer (16 k-steps)

__device__ __forceinline__ void slab_dma(float* __restrict__ dst, const float* __restrict__ src, int n_floats, int wave, int lane)
{
    lds_dma<WAVES>(dst, src, n_floats >> 8, wave, lane);   // 1 KB per wave-instruction (lds_dma.h)
}
template <int N_FLOATS>
__device__ __forceinline__ void slab_dma_c(float* __restrict__ dst, const float* __restrict__ src, int wave, int lane)
{
    lds_dma_c<WAVES, N_FLOATS / 256>(dst, src, wave, lane);
}

__device__ __forceinline__ void slab_sync()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// acc[b] += W_frag(t, b) * bfn(t), t in [0, 4*STEPS4), NBLK output blocks of 16 rows.  The B value of a k-step is evaluated
// once and used by all blocks; blocks are processed four at a time (16 fragment registers) and rotate, so consecutive MFMAs are
// independent.
template <int STEPS4, int NBLK, typename BFN>
__device__ __forceinline__ void gemm16(const float* __restrict__ w, f32x4 (&acc)[NBLK], int lane, BFN bfn)
{
    // groups of 4 blocks x 4 k-steps = 16 MFMAs; the fragments of group i+1 are requested BEFORE the MFMAs of group i issue
    constexpr int NG = STEPS4 * (NBLK / 4);
    f32x4 a[4], an[4];
    auto load = [&](f32x4 (&f)[4], int grp) {
        const int t4 = grp / (NBLK / 4), b0 = (grp % (NBLK / 4)) * 4;
#pragma unroll
        for (int b = 0; b < 4; ++b) f[b] = *reinterpret_cast<const f32x4*>(w + ((t4 * NBLK + b0 + b) * 64 + lane) * 4);
    };
    load(a, 0);
    float bv[4];
#pragma unroll
    for (int grp = 0; grp < NG; ++grp) {
        const int t4 = grp / (NBLK / 4), b0 = (grp % (NBLK / 4)) * 4;
        if (b0 == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = bfn(t4 * 4 + j);
        }
        if (grp + 1 < NG) load(an, grp + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int b = 0; b < 4; ++b)
                acc[b0 + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[b][j], bv[j], acc[b0 + b], 0, 0, 0);
        if (grp + 1 < NG) {
#pragma unroll
            for (int b = 0; b < 4; ++b) a[b] = an[b];
        }
    }
}

template <int NBLK>
__device__ __forceinline__ void init16(f32x4 (&acc)[NBLK], const float* __restrict__ vec_g)
{
#pragma unroll
    for (int b = 0; b < NBLK; ++b) acc[b] = *reinterpret_cast<const f32x4*>(vec_g + b * 4);
}

__device__ __forceinline__ float pe_sc(float x, int want_cos)      // same routine as mlp.hip (separate TU)
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

// B operand of embedding k-step t for lane group g (see kcol(K_PE)): everything but two selects is static in t
// embedding layout [x(3) | sin(x 2^f), f-major (30) | cos (30)]  (models.py:47-51)
__device__ __forceinline__ float pe16(int t, int g, float px, float py, float pz)
{
    if (t == 0) return g == 0 ? px : g == 1 ? py : g == 2 ? pz : 0.0f;
    const int ka = 2 * (t - 1), kb = ka + 1;                                  // index = 3*f + coordinate
    const int fa = ka / 3, ca = ka - 3 * fa, fb = kb / 3, cb = kb - 3 * fb;
    const float xa = (ca == 0 ? px : ca == 1 ? py : pz) * (float)(1 << fa);
    const float xb = (cb == 0 ? px : cb == 1 ? py : pz) * (float)(1 << fb);
    return pe_sc((g >> 1) ? xb : xa, g & 1);
}

template <bool ALPHA_ONLY>
__global__ __launch_bounds__(64 * WAVES, 2) void mlp_fwd16_kernel(
    const float* __restrict__ packed, int F, const float* __restrict__ ndc, int ndc_stride,
    const float* __restrict__ feat, int feat_stride, const float* __restrict__ dirs, int dirs_stride,
    int64_t P, int S, float* __restrict__ raw)
{
    extern __shared__ __attribute__((aligned(16))) float lds16[];
    float* buf0 = lds16;
    float* buf1 = lds16 + SLAB_F;
    float* vec = lds16 + 2 * SLAB_F;
    const Layout16 L = layout16(F);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;
    const int64_t p_raw = ((int64_t)blockIdx.x * WAVES + wave) * 16 + (lane & 15);
    const bool live = p_raw < P;
    const int64_t p = live ? p_raw : P - 1;

    slab_dma(buf0, packed + L.biasw, (int)seg(L.fst, 8), wave, lane);                    // slab 0
    for (int i = tid; i < W_TOTAL; i += 64 * WAVES) vec[i] = packed[L.vec + i];
    const float px = ndc[p * ndc_stride + 0], py = ndc[p * ndc_stride + 1], pz = ndc[p * ndc_stride + 2];
    float fv[12];                                                                        // feature column 4t+g, t < 12 (F <= 40)
    {
        const float* fp = feat + p * feat_stride;
#pragma unroll
        for (int t = 0; t < 12; ++t) { const int c = 4 * t + g; fv[t] = c < F ? fp[c] : 0.0f; }
    }
    float bias[32], h[32];
    auto pe = [&](int t) { return pe16(t, g, px, py, pz); };
    auto hlo = [&](int t) { return h[t]; };
    auto hhi = [&](int t) { return h[16 + t]; };

    // ---- slab 0: bias = pts_bias(feat)
    slab_sync();
    slab_dma(buf1, packed + L.l0, (int)seg(PE_ST, 8), wave, lane);                       // slab 1
    {
        f32x4 acc[8];
        init16<8>(acc, vec + W_BIASG + g * 32);
        auto fb = [&](int t) { return fv[t < 12 ? t : 0]; };
        if (L.fst == 4) gemm16<1, 8>(buf0, acc, lane, fb);
        else if (L.fst == 8) gemm16<2, 8>(buf0, acc, lane, fb);
        else gemm16<3, 8>(buf0, acc, lane, fb);
#pragma unroll
        for (int q = 0; q < 32; ++q) bias[q] = acc[q >> 2][q & 3];
    }
    // ---- slab 1: layer 0
    slab_sync();
    slab_dma_c<HALF>(buf0, packed + L.l1, wave, lane);                                     // slab 2
    {
        f32x4 acc[8];
        init16<8>(acc, vec + W_L0 + g * 32);
        gemm16<PE_ST / 4, 8>(buf1, acc, lane, pe);
#pragma unroll
        for (int q = 0; q < 32; ++q) h[q] = fmaxf(acc[q >> 2][q & 3] * bias[q], 0.0f);
    }
    // ---- layers 1..4: two slabs each (buf0 then buf1)
#pragma unroll 1
    for (int layer = 1; layer <= 4; ++layer) {
        const float* wl = packed + L.l1 + (size_t)(layer - 1) * seg(ACT_ST, 8);
        f32x4 acc[8];
        slab_sync();
        slab_dma_c<HALF>(buf1, wl + HALF, wave, lane);
        init16<8>(acc, vec + W_L0 + 128 * layer + g * 32);
        gemm16<4, 8>(buf0, acc, lane, hlo);
        slab_sync();
        slab_dma(buf0, layer < 4 ? wl + 2 * HALF : packed + L.l5a, layer < 4 ? HALF : (int)seg(PE_ST, 8), wave, lane);
        gemm16<4, 8>(buf1, acc, lane, hhi);
#pragma unroll
        for (int q = 0; q < 32; ++q) h[q] = fmaxf(acc[q >> 2][q & 3] * bias[q], 0.0f);
    }
    // ---- layer 5 on cat([pts, h4])
    float sigma;
    {
        f32x4 acc[8];
        slab_sync();
        slab_dma_c<HALF>(buf1, packed + L.l5b, wave, lane);
        init16<8>(acc, vec + W_L0 + 128 * 5 + g * 32);
        gemm16<PE_ST / 4, 8>(buf0, acc, lane, pe);
        slab_sync();
        slab_dma_c<HALF>(buf0, packed + L.l5b + HALF, wave, lane);
        gemm16<4, 8>(buf1, acc, lane, hlo);
        slab_sync();
        if (!ALPHA_ONLY) slab_dma_c<HALF>(buf1, packed + L.feat, wave, lane);
        gemm16<4, 8>(buf0, acc, lane, hhi);
#pragma unroll
        for (int q = 0; q < 32; ++q) h[q] = fmaxf(acc[q >> 2][q & 3] * bias[q], 0.0f);
        const float* wa = vec + W_WA + g * 32;
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 32; ++q) part = fmaf(wa[q], h[q], part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        sigma = fmaxf(part + vec[W_BA], 0.0f);
    }
    if (ALPHA_ONLY) {
        if (live && g == 0) raw[p_raw] = sigma;
        return;
    }
    // ---- feature_linear
    {
        f32x4 acc[8];
        slab_sync();
        slab_dma_c<HALF>(buf0, packed + L.feat + HALF, wave, lane);
        init16<8>(acc, vec + W_FEAT + g * 32);
        gemm16<4, 8>(buf1, acc, lane, hlo);
        slab_sync();
        slab_dma(buf1, packed + L.views, (int)seg(VIEW_ST, 4), wave, lane);
        gemm16<4, 8>(buf0, acc, lane, hhi);
#pragma unroll
        for (int q = 0; q < 32; ++q) h[q] = acc[q >> 2][q & 3];
    }
    // ---- views_linears[0] + rgb head
    {
        const int64_t ray = p / S;
        const float dg = g < 3 ? dirs[ray * dirs_stride + (g < 3 ? g : 0)] : 0.0f;
        f32x4 acc[4];
        slab_sync();
        init16<4>(acc, vec + W_VIEWS + g * 16);
        gemm16<VIEW_ST / 4, 4>(buf1, acc, lane, [&](int t) { return t < ACT_ST ? h[t < ACT_ST ? t : 0] : (t == ACT_ST ? dg : 0.0f); });
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* wr = vec + W_WR + c * 64 + g * 16;
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 16; ++q) part = fmaf(wr[q], fmaxf(acc[q >> 2][q & 3], 0.0f), part);
            part += __shfl_xor(part, 16);
            part += __shfl_xor(part, 32);
            rgb[c] = 1.0f / (1.0f + expf(-(part + vec[W_BR + c])));
        }
        if (live && g == 0) *reinterpret_cast<f32x4*>(raw + p_raw * 4) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
    }
}

}  // namespace mlp16

// ------------------------------------------------------------------------------------------ entry points used by mlp.hip
size_t mvs_mlp16_packed_floats(int F) { return mlp16::layout16(F).total; }

int mvs_mlp16_pack(const float* const w[11], const float* const b[11], int F, float* packed16, hipStream_t st)
{
    mlp16::Pack16Args a;
    for (int i = 0; i < 11; ++i) { a.w[i] = w[i]; a.b[i] = b[i]; }
    a.F = F;
    mlp16::pack16_kernel<<<64, 256, 0, st>>>(a, packed16);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int mvs_mlp16_fwd(const float* packed16, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                  const float* dirs, int dirs_stride, int64_t P, int S, int alpha_only, float* raw, hipStream_t st)
{
    using namespace mlp16;
    const size_t lds_bytes = LDS_F * sizeof(float);
    static unsigned long long cap_a = 0, cap_b = 0;         // per-device bit masks (common.h)
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd16_kernel<false>), (int)(lds_bytes), &cap_a)) return rc_;
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd16_kernel<true>), (int)(lds_bytes), &cap_b)) return rc_;
    const unsigned grid = mvs_cdiv(P, 16 * WAVES);
    if (alpha_only) mlp_fwd16_kernel<true><<<grid, 64 * WAVES, lds_bytes, st>>>(packed16, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw);
    else mlp_fwd16_kernel<false><<<grid, 64 * WAVES, lds_bytes, st>>>(packed16, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

#endif   // MVSNERF_DEV_KNOBS
