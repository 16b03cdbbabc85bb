// Fused Embedder + Renderer_ours MLP for gfx950 (models.py:17-51, 145-222; renderer.py:42-63).
//
// One workgroup = 4 waves = 128 points; each wave owns 32 points for the whole network.  Every layer
// is computed transposed on v_mfma_f32_32x32x2_f32 (exact fp32 fma chains, 157 TFLOP/s peak) with the
// weights as A operand streamed through a 64 KB LDS buffer and the activations as B operand straight
// from the previous layer's accumulator registers (see mlp_layout.h) - the 86-wide input and all
// 128-wide activations of the reference (~1.5 GB of ATen traffic per 1024x128 batch) never exist in
// memory.  MFMA-bound: 251 392 FLOP per point against 12 B + 4*F B read and 16 B written.
#include "common.h"
#include "mlp_layout.h"
#include "lds_dma.h"
#include "knobs.h"

using namespace mlp;
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------ pack
struct PackArgs {
    const float* w[11];
    const float* b[11];
    int F;
};
// order of w/b: 0..5 pts_linears, 6 pts_bias, 7 feature_linear, 8 alpha_linear, 9 views_linears.0, 10 rgb_linear

__device__ inline void pack_segment(float* __restrict__ dst, const float* __restrict__ W, int ld, int col_off,
                                    int kmap, int steps, int nb, int F, int tid, int nthreads)
{
    const int total = steps * nb * 64;
    for (int i = tid; i < total; i += nthreads) {
        const int j = i & 3;
        const int lane = (i >> 2) & 63;
        const int rest = i >> 8;               // t4*nb + b
        const int b = rest % nb, t = (rest / nb) * 4 + j;
        const int col = kmap_col(kmap, t, lane >> 5, F);
        const int row = b * 32 + (lane & 31);
        dst[i] = col < 0 ? 0.0f : W[(size_t)row * ld + col_off + col];
    }
}

__global__ __launch_bounds__(256) void mlp_pack_kernel(PackArgs a, float* __restrict__ packed)
{
    const Layout L = layout(a.F);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    pack_segment(packed + L.biasw, a.w[6], a.F, 0, K_FEAT, L.fsteps, 4, a.F, tid, nt);
    pack_segment(packed + L.l0, a.w[0], PE_DIM, 0, K_PE, PE_STEPS, 4, a.F, tid, nt);
    pack_segment(packed + L.l1, a.w[1], WIDTH, 0, K_ACT, ACT_STEPS, 4, a.F, tid, nt);
    pack_segment(packed + L.l2, a.w[2], WIDTH, 0, K_ACT, ACT_STEPS, 4, a.F, tid, nt);
    pack_segment(packed + L.l3, a.w[3], WIDTH, 0, K_ACT, ACT_STEPS, 4, a.F, tid, nt);
    pack_segment(packed + L.l4, a.w[4], WIDTH, 0, K_ACT, ACT_STEPS, 4, a.F, tid, nt);
    pack_segment(packed + L.l5a, a.w[5], WIDTH + PE_DIM, 0, K_PE, PE_STEPS, 4, a.F, tid, nt);        // cat([pts, h]) models.py:205
    pack_segment(packed + L.l5b, a.w[5], WIDTH + PE_DIM, PE_DIM, K_ACT, ACT_STEPS, 4, a.F, tid, nt);
    pack_segment(packed + L.feat, a.w[7], WIDTH, 0, K_ACT, ACT_STEPS, 4, a.F, tid, nt);
    pack_segment(packed + L.views, a.w[9], WIDTH + 3, 0, K_VIEWS, VIEW_STEPS, 2, a.F, tid, nt);
    float* v = packed + L.vec;
    for (int i = tid; i < V_TOTAL; i += nt) {
        float x = 0.0f;
        if (i < V_VIEWS) {                       // eight [2][64] bias vectors
            const int which = i >> 7, h = (i >> 6) & 1, q = i & 63;
            const float* src = which == 0 ? a.b[6] : which <= 6 ? a.b[which - 1] : a.b[7];
            x = src[act_n(q, h)];
        } else if (i < V_WA) {                   // views bias [2][32]
            const int k = i - V_VIEWS;
            x = a.b[9][act_n(k & 31, k >> 5)];
        } else if (i < V_BA) {                   // alpha weight [2][64]
            const int k = i - V_WA;
            x = a.w[8][act_n(k & 63, k >> 6)];
        } else if (i < V_WR) {
            x = (i == V_BA) ? a.b[8][0] : 0.0f;
        } else if (i < V_BR) {                   // rgb weight [3][2][32]
            const int k = i - V_WR, c = k >> 6, h = (k >> 5) & 1, q = k & 31;
            x = a.w[10][c * 64 + act_n(q, h)];
        } else {
            const int c = i - V_BR;
            x = c < 3 ? a.b[10][c] : 0.0f;
        }
        v[i] = x;
    }
}

static size_t off16(int F) { return (layout(F).total + 3) & ~(size_t)3; }

extern "C" size_t mvsnerf_mlp_packed_floats(int F)
{
    if (F < 2 || F > MAX_F || (F & 1)) return 0;
    return off16(F);
}

extern "C" int mvsnerf_mlp_pack(const float* const w[11], const float* const b[11], int F, float* packed, void* stream)
{
    if (!w || !b || !packed) return MVSNERF_EINVAL;
    if (F < 2 || F > MAX_F || (F & 1)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed)) return MVSNERF_EALIGN;
    PackArgs a;
    for (int i = 0; i < 11; ++i) {
        if (!w[i] || !b[i]) return MVSNERF_EINVAL;
        a.w[i] = w[i]; a.b[i] = b[i];
    }
    a.F = F;
    mlp_pack_kernel<<<64, 256, 0, (hipStream_t)stream>>>(a, packed);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ------------------------------------------------------------------------------------------ compute
constexpr int WBUF_FLOATS = 16384;                      // 64 KB weight stage
constexpr int LDS_FLOATS = WBUF_FLOATS + V_TOTAL;       // + fragment-ordered vectors (5.5 KB)

__device__ __forceinline__ void stage_weights(float* __restrict__ wbuf, const float* __restrict__ src, int n_floats, int tid)
{
    // 256 threads x float4, fully coalesced; n_floats is a multiple of 1024
    const f32x4* s = reinterpret_cast<const f32x4*>(src);
    f32x4* d = reinterpret_cast<f32x4*>(wbuf);
    const int n4 = n_floats >> 2;
#pragma unroll 4
    for (int i = tid; i < n4; i += 256) d[i] = s[i];
}

// acc[g][b] += W_frag(t, b) * bfn(g, t) for t in [0, 4*STEPS4); G = 32-point groups per wave
// (one A fragment read from LDS feeds G MFMAs).
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `after_first_reads` runs once, between the LDS reads of the first fragment group and its MFMAs: the pipelined kernel
// issues the next slab's DMA there, so that the DMA instructions fill the LDS latency instead of preceding the reads.
template <int STEPS4, int NBLK, int G, typename BFN, typename HOOK = NoHook>
__device__ __forceinline__ void gemm_stage(const float* __restrict__ w, f32x16 (&acc)[G][NBLK], int lane, BFN bfn, HOOK after_first_reads = HOOK())
{
#pragma unroll
    for (int t4 = 0; t4 < STEPS4; ++t4) {
        f32x4 a[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b)
            a[b] = *reinterpret_cast<const f32x4*>(w + ((t4 * NBLK + b) * 64 + lane) * 4);
        if (t4 == 0) after_first_reads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float bv[G];
#pragma unroll
            for (int g = 0; g < G; ++g) bv[g] = bfn(g, t4 * 4 + j);
#pragma unroll
            for (int b = 0; b < NBLK; ++b)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    acc[g][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][j], bv[g], acc[g][b], 0, 0, 0);
        }
    }
}

template <int NBLK, int G>
__device__ __forceinline__ void init_acc(f32x16 (&acc)[G][NBLK], const float* __restrict__ vec_h)
{
    // vec_h points at this lane-half's [NBLK*16] bias fragment (LDS broadcast reads)
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(vec_h + b * 16 + r4 * 4);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                acc[g][b][r4 * 4 + 0] = v[0]; acc[g][b][r4 * 4 + 1] = v[1]; acc[g][b][r4 * 4 + 2] = v[2]; acc[g][b][r4 * 4 + 3] = v[3];
            }
        }
}

// sin or cos of x for the positional encoding (x = ndc * 2^f, f <= 9, ndc in [0,1] inside the volume): Cody-Waite
// reduction by pi/2 in three fma steps (k*C1 is exact for |k| < 2^16, i.e. |x| < 1e5), cephes minimax polynomials on
// [-pi/4, pi/4], quadrant select.  cos(x) = sin(x + pi/2) is a quadrant shift, so every lane evaluates one polynomial
// pair and picks.  Branch-free on purpose (a branch here would cut the unrolled MFMA stream into basic blocks).
// Max abs error 7.6e-8 on |x| <= 1600 (numpy float32 sin: 6.6e-8).  Arguments are clamped to +-65536, i.e. samples
// more than 128 volume-widths outside the frustum (where the encoding is physically meaningless anyway).
__device__ __forceinline__ float pe_sin_or_cos(float x, int want_cos)
{
    x = fminf(fmaxf(x, -65536.0f), 65536.0f);
    const float k = rintf(x * 0.63661977236758134f);
    float r = fmaf(k, -1.5703125f, x);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188e-8f, r);
    const float r2 = r * r;
    const float sp = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    const float sn = fmaf(sp * r2, r, r);
    const float cp = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    const float cs = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k + want_cos;
    const float v = (q & 1) ? cs : sn;
    return (q & 2) ? -v : v;
}

// positional-encoding B operand of k-step t for this lane (point coords px,py,pz; half)
__device__ __forceinline__ float pe_operand(int t, int half, float px, float py, float pz)
{
    if (t == 0) return half ? py : px;
    if (t == 1) return half ? 0.0f : pz;
    const int j = t - 2, f = j / 3, c = j - 3 * f;
    const float x = (c == 0 ? px : c == 1 ? py : pz) * (float)(1 << f);   // exact, as x*2^f in models.py:49
    return pe_sin_or_cos(x, half);
}


// ------------------------------------------------------------------------------------------ pipelined forward
// Measured on MI355X: 0.237 ms per 1024x128 batch = 139 TFLOP/s = 88 % of the 157.3 TFLOP/s fp32-MFMA peak; PMC:
// matrix pipes busy 85 % of the kernel's duration.  What the rest is: VALU instructions cost matrix-pipe issue time on this
// chip whichever wave issues them (scratch/mfma_mix.hip: one v_fma per MFMA takes 12 % off the MFMA rate), ~1 950 of them per
// 1 976 MFMAs here; launch ramp / tail of a two-round grid.  Negative results kept out of the code: (i) staggering /
// prioritising the two co-resident workgroups of a CU: no change; (ii) reading A fragments straight from L2 (no LDS stage, no
// barriers): 114 TFLOP/s; (iii) 64 points per wave at one wave per SIMD: 114 TFLOP/s.
// Same arithmetic as mlp_fwd_kernel<.., G=1, ..>, different weight logistics: the packed weights are cut into 16 slabs
// of <= 34 KB (half a 128x128 layer = 32 k-steps) that alternate between two LDS buffers.  While the MFMAs of slab i
// run, slab i+1 arrives by LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write pass); one barrier per slab.
constexpr int SLAB_FLOATS = 8704;                        // 34 KB = 34 k-steps x 4 blocks x 64 lanes (views: 68 x 2)
constexpr int PIPE_LDS_FLOATS = 2 * SLAB_FLOATS + V_TOTAL;

__device__ __forceinline__ void slab_dma(float* __restrict__ dst, const float* __restrict__ src, int n_floats, int wave, int lane)
{
    lds_dma<4>(dst, src, n_floats >> 8, wave, lane);     // 1 KB per wave-instruction, scalar base + one lane offset (lds_dma.h)
}
template <int N_FLOATS>
__device__ __forceinline__ void slab_dma_c(float* __restrict__ dst, const float* __restrict__ src, int wave, int lane)
{
    lds_dma_c<4, N_FLOATS / 256>(dst, src, wave, lane);
}

__device__ __forceinline__ void slab_sync()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces (and earlier stores) have landed
    __syncthreads();                                      // ... everybody's have, and everybody left the other buffer
}

// One tile = 128 points = one workgroup's work; `tile` is the workgroup index in the plain kernel and the loop variable of the predicated one below.
template <bool ALPHA_ONLY, bool SAVE>
__device__ __forceinline__ void mlp_fwd_pipe_tile(
    const unsigned tile, const float* __restrict__ packed, int F, const float* __restrict__ ndc, int ndc_stride,
    const float* __restrict__ feat, int feat_stride, const float* __restrict__ dirs, int dirs_stride,
    int64_t P, int S, float* __restrict__ raw, float* __restrict__ saved, long long* __restrict__ census)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    long long t_start = 0, c_start = 0;
    if (census) { t_start = wall_clock64(); c_start = __builtin_amdgcn_s_memtime(); }
    int stamp_i = 4;
    auto stamp = [&]() { if (census && threadIdx.x == 0 && stamp_i < 16) census[tile * 16 + stamp_i] = wall_clock64(); ++stamp_i; };
    float* buf0 = lds;
    float* buf1 = lds + SLAB_FLOATS;
    float* vec = lds + 2 * SLAB_FLOATS;
    constexpr int G = 1;
    const Layout L = layout(F);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int64_t p_raw = ((int64_t)tile * 4 + wave) * 32 + (lane & 31);
    const bool live = p_raw < P;
    const int64_t p = live ? p_raw : P - 1;
    float* sv = nullptr;
    if (SAVE) sv = saved + ((int64_t)tile * 4 + wave) * (SLOTS_SAVED * 64) + lane;
    auto save = [&](int slot, float v) { if (SAVE) sv[slot * 64] = v; };
    constexpr int HALF = (int)(ACT_STEPS / 2) * 4 * 64;                   // floats of half a 128x128 layer

    slab_dma(buf0, packed + L.biasw, (int)seg_floats(L.fsteps, 4), wave, lane);          // slab 0
    slab_dma_c<(int)seg_floats(PE_STEPS, 4)>(buf1, packed + L.l0, wave, lane);              // slab 1 (buf1 is free at tile start)
    for (int i = tid; i < V_TOTAL; i += 256) vec[i] = packed[L.vec + i];
    const float px = ndc[p * ndc_stride + 0], py = ndc[p * ndc_stride + 1], pz = ndc[p * ndc_stride + 2];
    float fv[MAX_F / 2];
    {
        const float* fp = feat + p * feat_stride + half * (F / 2);
#pragma unroll
        for (int i = 0; i < MAX_F / 2; ++i) fv[i] = i < F / 2 ? fp[i] : 0.0f;
    }
    float bias[64], h[64];
    auto pe = [&](int g, int t) { return pe_operand(t, half, px, py, pz); };
    auto hlo = [&](int g, int t) { return h[t]; };
    auto hhi = [&](int g, int t) { return h[32 + t]; };

    // ---- slab 0: bias = pts_bias(feat)
    slab_sync();
    stamp();                                                                                // [4] startup done
    {
        f32x16 acc[G][4];
        init_acc<4, G>(acc, vec + V_BIASG + half * 64);
        auto fb = [&](int g, int t) { return fv[t]; };
        switch (L.fsteps) {
            case 4:  gemm_stage<1, 4, G>(buf0, acc, lane, fb); break;
            case 8:  gemm_stage<2, 4, G>(buf0, acc, lane, fb); break;
            case 12: gemm_stage<3, 4, G>(buf0, acc, lane, fb); break;
            case 16: gemm_stage<4, 4, G>(buf0, acc, lane, fb); break;
            default: gemm_stage<5, 4, G>(buf0, acc, lane, fb); break;
        }
#pragma unroll
        for (int q = 0; q < 64; ++q) { bias[q] = acc[0][q >> 4][q & 15]; save(S_BM + q, bias[q]); }
        if (SAVE) {
#pragma unroll
            for (int t = 0; t < 16; ++t) save(S_FV + t, fv[t]);
#pragma unroll
            for (int t = 0; t < PE_STEPS; ++t) save(S_E + t, pe_operand(t, half, px, py, pz));
        }
    }
    stamp();                                                                                // [5] bias gemm done
    // ---- slab 1: layer 0
    slab_sync();
    slab_dma_c<HALF>(buf0, packed + L.l1, wave, lane);                                      // slab 2
    {
        f32x16 acc[G][4];
        init_acc<4, G>(acc, vec + V_L0 + half * 64);
        gemm_stage<PE_STEPS / 4, 4, G>(buf1, acc, lane, pe);
#pragma unroll
        for (int q = 0; q < 64; q += 2) {
            const f32x2 m2 = f32x2{acc[0][q >> 4][q & 15], acc[0][q >> 4][(q & 15) + 1]} * f32x2{bias[q], bias[q + 1]};   // v_pk_mul_f32
            h[q] = fmaxf(m2[0], 0.0f); h[q + 1] = fmaxf(m2[1], 0.0f);
            save(S_H + q, h[q]); save(S_H + q + 1, h[q + 1]);
        }
    }
    stamp();                                                                                // [6] layer 0 done
    // ---- layers 1..4: two slabs each (buf0 then buf1)
#pragma unroll 1
    for (int layer = 1; layer <= 4; ++layer) {
        const float* wl = packed + L.l1 + (size_t)(layer - 1) * seg_floats(ACT_STEPS, 4);
        f32x16 acc[G][4];
        slab_sync();
        init_acc<4, G>(acc, vec + V_L0 + 128 * layer + half * 64);
        gemm_stage<8, 4, G>(buf0, acc, lane, hlo, [&]() { slab_dma_c<HALF>(buf1, wl + HALF, wave, lane); });
        slab_sync();
        // next slab: first half of the next layer, or the positional-encoding part of layer 5
        gemm_stage<8, 4, G>(buf1, acc, lane, hhi, [&]() { slab_dma_c<HALF>(buf0, layer < 4 ? wl + 2 * HALF : packed + L.l5a, wave, lane); });
#pragma unroll
        for (int q = 0; q < 64; q += 2) {
            const f32x2 m2 = f32x2{acc[0][q >> 4][q & 15], acc[0][q >> 4][(q & 15) + 1]} * f32x2{bias[q], bias[q + 1]};   // v_pk_mul_f32
            h[q] = fmaxf(m2[0], 0.0f); h[q + 1] = fmaxf(m2[1], 0.0f);
            save(S_H + layer * 64 + q, h[q]); save(S_H + layer * 64 + q + 1, h[q + 1]);
        }
        stamp();                                                                            // [7..10] layers 1..4 done
    }
    // ---- layer 5 on cat([pts, h4]): slabs 10 (buf0), 11 (buf1), 12 (buf0)
    float sigma;
    {
        f32x16 acc[G][4];
        slab_sync();
        init_acc<4, G>(acc, vec + V_L0 + 128 * 5 + half * 64);
        gemm_stage<PE_STEPS / 4, 4, G>(buf0, acc, lane, pe, [&]() { slab_dma_c<HALF>(buf1, packed + L.l5b, wave, lane); });
        slab_sync();
        gemm_stage<8, 4, G>(buf1, acc, lane, hlo, [&]() { slab_dma_c<HALF>(buf0, packed + L.l5b + HALF, wave, lane); });
        slab_sync();
        gemm_stage<8, 4, G>(buf0, acc, lane, hhi, [&]() { if (!ALPHA_ONLY) slab_dma_c<HALF>(buf1, packed + L.feat, wave, lane); });
#pragma unroll
        for (int q = 0; q < 64; q += 2) {
            const f32x2 m2 = f32x2{acc[0][q >> 4][q & 15], acc[0][q >> 4][(q & 15) + 1]} * f32x2{bias[q], bias[q + 1]};   // v_pk_mul_f32
            h[q] = fmaxf(m2[0], 0.0f); h[q + 1] = fmaxf(m2[1], 0.0f);
            save(S_H + 5 * 64 + q, h[q]); save(S_H + 5 * 64 + q + 1, h[q + 1]);
        }
        const float* wa = vec + V_WA + half * 64;
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 64; ++q) part = fmaf(wa[q], h[q], part);
        part += __shfl_xor(part, 32);
        sigma = fmaxf(part + vec[V_BA], 0.0f);
    }
    stamp();                                                                                // [11] layer 5 + sigma head done
    if (ALPHA_ONLY) {
        if (live && half == 0) raw[p_raw] = sigma;
        return;
    }
    // ---- feature_linear: slabs 13 (buf1), 14 (buf0)
    {
        f32x16 acc[G][4];
        slab_sync();
        init_acc<4, G>(acc, vec + V_FEAT + half * 64);
        gemm_stage<8, 4, G>(buf1, acc, lane, hlo, [&]() { slab_dma_c<HALF>(buf0, packed + L.feat + HALF, wave, lane); });
        slab_sync();
        gemm_stage<8, 4, G>(buf0, acc, lane, hhi, [&]() { slab_dma_c<(int)seg_floats(VIEW_STEPS, 2)>(buf1, packed + L.views, wave, lane); });
#pragma unroll
        for (int q = 0; q < 64; ++q) { h[q] = acc[0][q >> 4][q & 15]; save(S_FE + q, h[q]); }
    }
    stamp();                                                                                // [12] feature_linear done
    // ---- views_linears[0] + rgb head: slab 15 (buf1)
    {
        float d0, d1, d2;
        { const int64_t ray = p / S; d0 = dirs[ray * dirs_stride + 0]; d1 = dirs[ray * dirs_stride + 1]; d2 = dirs[ray * dirs_stride + 2]; }
        f32x16 acc[G][2];
        slab_sync();
        init_acc<2, G>(acc, vec + V_VIEWS + half * 32);
        gemm_stage<VIEW_STEPS / 4, 2, G>(buf1, acc, lane, [&](int g, int t) {
            return t < ACT_STEPS ? h[t < ACT_STEPS ? t : 0] : t == ACT_STEPS ? (half ? d1 : d0) : t == ACT_STEPS + 1 ? (half ? 0.0f : d2) : 0.0f;
        });
        if (SAVE) {
#pragma unroll
            for (int q = 0; q < 32; ++q) save(S_HV + q, fmaxf(acc[0][q >> 4][q & 15], 0.0f));
            save(S_DR + 0, half ? d1 : d0);
            save(S_DR + 1, half ? 0.0f : d2);
        }
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* wr = vec + V_WR + c * 64 + half * 32;
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 32; ++q) part = fmaf(wr[q], fmaxf(acc[0][q >> 4][q & 15], 0.0f), part);
            part += __shfl_xor(part, 32);
            const float x = part + vec[V_BR + c];
            rgb[c] = 1.0f / (1.0f + expf(-x));
        }
        if (live && half == 0) *reinterpret_cast<f32x4*>(raw + p_raw * 4) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
    }
    if (census && tid == 0) {
        census[tile * 16 + 0] = t_start;
        census[tile * 16 + 1] = wall_clock64();
        census[tile * 16 + 2] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        census[tile * 16 + 3] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));     // HW_REG_XCC_ID
        census[tile * 16 + 13] = __builtin_amdgcn_s_memtime() - c_start;                      // shader-clock ticks of this workgroup
    }
}

template <bool ALPHA_ONLY, bool SAVE>
__global__ __launch_bounds__(256, 2) void mlp_fwd_pipe_kernel(
    const float* __restrict__ packed, int F, const float* __restrict__ ndc, int ndc_stride,
    const float* __restrict__ feat, int feat_stride, const float* __restrict__ dirs, int dirs_stride,
    int64_t P, int S, float* __restrict__ raw, float* __restrict__ saved, long long* __restrict__ census)
{
    mlp_fwd_pipe_tile<ALPHA_ONLY, SAVE>(blockIdx.x, packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, saved, census);
}

// Second half of a guarded 16-bit sequence (include/mvsnerf_hip.h): launched behind the fp16x3 kernel, does its work only when that kernel
// reported a value outside fp16's range.  A small persistent grid that walks the tiles: when the guard is clear - the usual case - a few
// hundred workgroups leave at once (the plain kernel's 1024 early exits, each waiting for its 69 KB of LDS, cost ~3 us of every batch);
// when it is set, the same per-tile code runs and the results are the plain kernel's bits.
template <bool ALPHA_ONLY>
__global__ __launch_bounds__(256, 2) void mlp_fwd_pipe_if_kernel(
    const float* __restrict__ packed, int F, const float* __restrict__ ndc, int ndc_stride,
    const float* __restrict__ feat, int feat_stride, const float* __restrict__ dirs, int dirs_stride,
    int64_t P, int S, float* __restrict__ raw, const int* __restrict__ run_if)
{
    if (*run_if == 0) return;
    const unsigned n_tiles = (unsigned)((P + 127) / 128);
    for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        mlp_fwd_pipe_tile<ALPHA_ONLY, false>(tile, packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, nullptr, nullptr);
        __syncthreads();                                       // the next tile's first slabs overwrite what the last GEMMs of this one read
    }
}

template <bool AO, bool SAVE>
static int launch_mlp_pipe(const float* packed, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                           const float* dirs, int dirs_stride, int64_t P, int S, float* raw, hipStream_t st, float* saved = nullptr, long long* census = nullptr,
                           const int* run_if = nullptr)
{
    const size_t lds_bytes = PIPE_LDS_FLOATS * sizeof(float);
    if (run_if) {                                       // predicated on a guard word: persistent grid (never with an activation store or a census)
        if constexpr (!SAVE) {
            static unsigned long long cap_if = 0;
            if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_pipe_if_kernel<AO>), (int)lds_bytes, &cap_if)) return rc_;
            const unsigned n_tiles = mvs_cdiv(P, 128);
            mlp_fwd_pipe_if_kernel<AO><<<n_tiles < 512u ? n_tiles : 512u, 256, lds_bytes, st>>>(packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, run_if);
            MVS_LAUNCH_CHECK();
            return MVSNERF_OK;
        } else {
            return MVSNERF_EINVAL;
        }
    }
    static unsigned long long lds_cap_set = 0;          // per-device bit mask (common.h)
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_pipe_kernel<AO, SAVE>), (int)lds_bytes, &lds_cap_set)) return rc_;
    mlp_fwd_pipe_kernel<AO, SAVE><<<mvs_cdiv(P, 128), 256, lds_bytes, st>>>(packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, saved, census);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// (The lookups fused into this kernel's prologue were measured in round 2 - 0.2574 ms/step against 0.2551 ms with the separate 11 us gather
// launch, whose 2048 waves hide the lookup latency that every workgroup's first GEMM would otherwise wait for - and are not built.)

static int mlp_fwd_checked(const float* packed, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                           const float* dirs, int dirs_stride, int64_t N, int S, int alpha_only, float* raw, void* stream, long long* census,
                           const int* run_if = nullptr)
{
    if (!packed || !ndc || !feat || !raw || N < 0 || S < 1 || feat_stride < F || ndc_stride < 3) return MVSNERF_EINVAL;
    if (!alpha_only && dirs_stride < 3) return MVSNERF_EINVAL;
    if (!alpha_only && !dirs) return MVSNERF_EINVAL;
    if (F < 2 || F > MAX_F || (F & 1)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed) || !mvs_aligned16(raw)) return MVSNERF_EALIGN;
    const int64_t P = N * S;
    if (P == 0) return MVSNERF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (alpha_only) return launch_mlp_pipe<true, false>(packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, st, nullptr, census, run_if);
    return launch_mlp_pipe<false, false>(packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, st, nullptr, census, run_if);
}

// mvsnerf_mlp_fwd predicated on a guard word (raymarch.hip: the fp32 re-run of a guarded fp16x3 batch)
int mvs_mlp_fwd_if(const float* packed, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                   const float* dirs, int dirs_stride, int64_t N, int S, int alpha_only, float* raw, const int* run_if, void* stream)
{
    return mlp_fwd_checked(packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, N, S, alpha_only, raw, stream, nullptr, run_if);
}

extern "C" int mvsnerf_mlp_fwd(const float* packed, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                               const float* dirs, int dirs_stride, int64_t N, int S, int alpha_only, float* raw, void* stream)
{
    return mlp_fwd_checked(packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, N, S, alpha_only, raw, stream, nullptr);
}

// The same launch with a per-workgroup timing record (stateless diagnostics for bench.py's sustained-clock figure): census =
// (N*S + 127) / 128 rows of 16 int64 {start, end (100 MHz wall clock), HW_ID, XCC_ID, phase stamps [4..12], shader-clock ticks [13]}.
extern "C" int mvsnerf_mlp_fwd_census(const float* packed, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                                      const float* dirs, int dirs_stride, int64_t N, int S, int alpha_only, float* raw, long long* census, void* stream)
{
    if (!census) return MVSNERF_EINVAL;
    return mlp_fwd_checked(packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, N, S, alpha_only, raw, stream, census);
}

// Training forward: identical arithmetic to mvsnerf_mlp_fwd (32 points per wave) + the activation store the
// backward pass consumes (mvsnerf_mlp_saved_floats(N*S) floats, slot format of mlp_layout.h).
extern "C" int mvsnerf_mlp_fwd_train(const float* packed, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                                     const float* dirs, int dirs_stride, int64_t N, int S, float* raw, float* saved, void* stream)
{
    if (!packed || !ndc || !feat || !dirs || !raw || !saved || N < 0 || S < 1 || feat_stride < F || ndc_stride < 3 || dirs_stride < 3) return MVSNERF_EINVAL;
    if (F < 2 || F > 32 || (F & 1)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed) || !mvs_aligned16(raw) || !mvs_aligned16(saved)) return MVSNERF_EALIGN;
    const int64_t P = N * S;
    if (P == 0) return MVSNERF_OK;
    return launch_mlp_pipe<false, true>(packed, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, (hipStream_t)stream, saved);
}

