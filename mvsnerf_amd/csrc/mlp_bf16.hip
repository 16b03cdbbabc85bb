// bf16-MFMA variant of the fused Embedder + Renderer_ours forward (BASELINE configs 3/4: "bf16", "MFMA-bf16 MLP").
// Opt-in; the fp32-MFMA kernel of mlp.hip stays the default and the headline (parity 1e-4 needs fp32 products).
//
// Same structure as the fp32 kernel - 32 points per wave, every layer transposed, the C/D fragment of one layer feeds
// the B operand of the next from registers - on v_mfma_f32_32x32x16_bf16 (16x the fp32-MFMA rate): a k-step now spans 16
// inputs, 8 per lane half, so the activation registers q = 8s..8s+7 of a lane, converted with v_cvt_pk_bf16_f32, ARE the B
// operand of step s (k-pairing n(8s+j, half), weights permuted to match at pack time).  Accumulation, biases, the
// multiplicative modulation, ReLU, the positional encoding and the two small heads stay fp32.  With the MFMA work cut 16x the
// kernel is VALU-bound (sin/cos, epilogues, conversions).
#include "common.h"
#include "mlp_layout.h"

using namespace mlp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int B_PE_STEPS = 4;        // 64 padded embedding inputs / 16
constexpr int B_ACT_STEPS = 8;       // 128 / 16
constexpr int B_VIEW_STEPS = 9;      // 128 feature + 3 dir (+13 zero) / 16

__host__ __device__ inline int b_feat_steps(int F) { return ((F / 2) + 7) / 8; }
__host__ __device__ inline size_t b_seg(int steps, int nb) { return (size_t)steps * nb * 64 * 8; }      // bf16 elements

struct LayoutB { size_t featw, l0, l1, l2, l3, l4, l5a, l5b, feat, views, total; int fsteps; };
__host__ __device__ inline LayoutB layout_b(int F)
{
    LayoutB L;
    L.fsteps = b_feat_steps(F);
    size_t o = 0;
    L.featw = o; o += b_seg(L.fsteps, 4);
    L.l0 = o;    o += b_seg(B_PE_STEPS, 4);
    L.l1 = o;    o += b_seg(B_ACT_STEPS, 4);
    L.l2 = o;    o += b_seg(B_ACT_STEPS, 4);
    L.l3 = o;    o += b_seg(B_ACT_STEPS, 4);
    L.l4 = o;    o += b_seg(B_ACT_STEPS, 4);
    L.l5a = o;   o += b_seg(B_PE_STEPS, 4);
    L.l5b = o;   o += b_seg(B_ACT_STEPS, 4);
    L.feat = o;  o += b_seg(B_ACT_STEPS, 4);
    L.views = o; o += b_seg(B_VIEW_STEPS, 2);
    L.total = o;
    return L;
}

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

__device__ inline void pack_b_segment(__bf16* __restrict__ dst, const float* __restrict__ W, int ld, int col_off, int kmap,
                                      int steps, int nb, int F, int tid, int nthreads)
{
    const int total = steps * nb * 64 * 8;
    for (int i = tid; i < total; i += nthreads) {
        const int j = i & 7, lane = (i >> 3) & 63, rest = i >> 9;          // rest = s*nb + b
        const int b = rest % nb, s = rest / nb;
        const int col = b_col(kmap, 8 * s + j, lane >> 5, F);
        const int row = b * 32 + (lane & 31);
        dst[i] = (__bf16)(col < 0 ? 0.0f : W[(size_t)row * ld + col_off + col]);
    }
}

__global__ __launch_bounds__(256) void mlp_pack_bf16_kernel(PackBArgs a, __bf16* __restrict__ packed)
{
    const LayoutB L = layout_b(a.F);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    pack_b_segment(packed + L.featw, a.w[6], a.F, 0, K_FEAT, L.fsteps, 4, a.F, tid, nt);
    pack_b_segment(packed + L.l0, a.w[0], PE_DIM, 0, K_PE, B_PE_STEPS, 4, a.F, tid, nt);
    pack_b_segment(packed + L.l1, a.w[1], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt);
    pack_b_segment(packed + L.l2, a.w[2], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt);
    pack_b_segment(packed + L.l3, a.w[3], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt);
    pack_b_segment(packed + L.l4, a.w[4], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt);
    pack_b_segment(packed + L.l5a, a.w[5], WIDTH + PE_DIM, 0, K_PE, B_PE_STEPS, 4, a.F, tid, nt);
    pack_b_segment(packed + L.l5b, a.w[5], WIDTH + PE_DIM, PE_DIM, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt);
    pack_b_segment(packed + L.feat, a.w[7], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, a.F, tid, nt);
    pack_b_segment(packed + L.views, a.w[9], WIDTH + 3, 0, K_VIEWS, B_VIEW_STEPS, 2, a.F, tid, nt);
}

// ------------------------------------------------------------------------------------------ kernel
constexpr int SLABB_BYTES = 32768;                                  // one 128x128 bf16 layer
constexpr int B_LDS_BYTES = 2 * SLABB_BYTES + V_TOTAL * 4;

__device__ __forceinline__ void slabb_dma(char* __restrict__ dst, const __bf16* __restrict__ src, size_t n_elems, int wave, int lane)
{
    const int pieces = (int)(n_elems >> 9);                          // 1 KB (512 bf16) per wave-instruction
    const char* s = reinterpret_cast<const char*>(src);
    for (int pc = wave; pc < pieces; pc += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(s + pc * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + pc * 1024), 16, 0, 0);
}

__device__ __forceinline__ void slabb_sync()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <int STEPS, int NBLK, typename BFN>
__device__ __forceinline__ void gemm_b(const char* __restrict__ w, f32x16 (&acc)[NBLK], int lane, BFN bfn)
{
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        const bf16x8 b = bfn(s);
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(w + ((s * NBLK + nb) * 64 + lane) * 16);
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
        }
    }
}

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

__device__ __forceinline__ float pe_sc(float x, int want_cos)      // same routine as mlp.hip (kept local: separate TU)
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

__device__ __forceinline__ float pe_op(int t, int half, float px, float py, float pz)
{
    if (t == 0) return half ? py : px;
    if (t == 1) return half ? 0.0f : pz;
    const int j = t - 2, f = j / 3, c = j - 3 * f;
    return pe_sc((c == 0 ? px : c == 1 ? py : pz) * (float)(1 << f), half);
}

__device__ __forceinline__ bf16x8 pack8(const float* v)
{
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (__bf16)v[j];
    return r;
}

template <bool ALPHA_ONLY>
__global__ __launch_bounds__(256, 2) void mlp_fwd_bf16_kernel(
    const __bf16* __restrict__ wq, const float* __restrict__ packed_f32, int F, const float* __restrict__ ndc, int ndc_stride,
    const float* __restrict__ feat, int feat_stride, const float* __restrict__ dirs, int dirs_stride,
    int64_t P, int S, float* __restrict__ raw)
{
    extern __shared__ __attribute__((aligned(16))) char lds_b[];
    char* buf0 = lds_b;
    char* buf1 = lds_b + SLABB_BYTES;
    float* vec = reinterpret_cast<float*>(lds_b + 2 * SLABB_BYTES);
    const LayoutB L = layout_b(F);
    const Layout LF = layout(F);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int64_t p_raw = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const bool live = p_raw < P;
    const int64_t p = live ? p_raw : P - 1;

    // slab 0 = pts_bias weights + layer 0 (contiguous in the packed buffer)
    slabb_dma(buf0, wq + L.featw, L.l1 - L.featw, wave, lane);
    for (int i = tid; i < V_TOTAL; i += 256) vec[i] = packed_f32[LF.vec + i];
    const float px = ndc[p * ndc_stride + 0], py = ndc[p * ndc_stride + 1], pz = ndc[p * ndc_stride + 2];
    float fv[24];
    {
        const float* fp = feat + p * feat_stride + half * (F / 2);
#pragma unroll
        for (int i = 0; i < 24; ++i) fv[i] = i < F / 2 ? fp[i] : 0.0f;
    }
    bf16x8 pe8[B_PE_STEPS];                      // positional-encoding operands (reused by layer 5)
#pragma unroll
    for (int s = 0; s < B_PE_STEPS; ++s) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = pe_op(8 * s + j, half, px, py, pz);
        pe8[s] = pack8(t8);
    }
    float bias[64];
    bf16x8 hb[8];
    float hf[64];                                 // fp32 copy of the current activations (epilogue scratch / heads)
    auto to_b = [&]() {
#pragma unroll
        for (int s = 0; s < 8; ++s) hb[s] = pack8(hf + 8 * s);
    };

    slabb_sync();
    slabb_dma(buf1, wq + L.l1, b_seg(B_ACT_STEPS, 4), wave, lane);
    {   // bias = pts_bias(feat)
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_BIASG + half * 64);
        auto fb = [&](int s) { return pack8(fv + 8 * s); };
        if (L.fsteps == 1) gemm_b<1, 4>(buf0, acc, lane, fb);
        else if (L.fsteps == 2) gemm_b<2, 4>(buf0, acc, lane, fb);
        else gemm_b<3, 4>(buf0, acc, lane, fb);
#pragma unroll
        for (int q = 0; q < 64; ++q) bias[q] = acc[q >> 4][q & 15];
    }
    {   // layer 0
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + half * 64);
        gemm_b<B_PE_STEPS, 4>(buf0 + b_seg(L.fsteps, 4) * 2, acc, lane, [&](int s) { return pe8[s]; });
#pragma unroll
        for (int q = 0; q < 64; ++q) hf[q] = fmaxf(acc[q >> 4][q & 15] * bias[q], 0.0f);
        to_b();
    }
    // layers 1..4: slabs alternate buf1, buf0, buf1, buf0
#pragma unroll 1
    for (int layer = 1; layer <= 4; ++layer) {
        char* cur = (layer & 1) ? buf1 : buf0;
        char* nxt = (layer & 1) ? buf0 : buf1;
        slabb_sync();
        if (layer < 4) slabb_dma(nxt, wq + L.l1 + (size_t)layer * b_seg(B_ACT_STEPS, 4), b_seg(B_ACT_STEPS, 4), wave, lane);
        else slabb_dma(nxt, wq + L.l5a, b_seg(B_PE_STEPS, 4), wave, lane);           // after layer 4 (in buf0): L5a -> buf1
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + 128 * layer + half * 64);
        gemm_b<B_ACT_STEPS, 4>(cur, acc, lane, [&](int s) { return hb[s]; });
#pragma unroll
        for (int q = 0; q < 64; ++q) hf[q] = fmaxf(acc[q >> 4][q & 15] * bias[q], 0.0f);
        to_b();
    }
    float sigma;
    {   // layer 5 on cat([pts, h4]): L5a in buf1, L5b -> buf0
        f32x16 acc[4];
        slabb_sync();
        slabb_dma(buf0, wq + L.l5b, b_seg(B_ACT_STEPS, 4), wave, lane);
        init_acc_b<4>(acc, vec + V_L0 + 128 * 5 + half * 64);
        gemm_b<B_PE_STEPS, 4>(buf1, acc, lane, [&](int s) { return pe8[s]; });
        slabb_sync();
        if (!ALPHA_ONLY) slabb_dma(buf1, wq + L.feat, b_seg(B_ACT_STEPS, 4), wave, lane);
        gemm_b<B_ACT_STEPS, 4>(buf0, acc, lane, [&](int s) { return hb[s]; });
#pragma unroll
        for (int q = 0; q < 64; ++q) hf[q] = fmaxf(acc[q >> 4][q & 15] * bias[q], 0.0f);
        const float* wa = vec + V_WA + half * 64;
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 64; ++q) part = fmaf(wa[q], hf[q], part);
        part += __shfl_xor(part, 32);
        sigma = fmaxf(part + vec[V_BA], 0.0f);
        to_b();
    }
    if (ALPHA_ONLY) {
        if (live && half == 0) raw[p_raw] = sigma;
        return;
    }
    {   // feature_linear (buf1), then views -> buf0
        f32x16 acc[4];
        slabb_sync();
        slabb_dma(buf0, wq + L.views, b_seg(B_VIEW_STEPS, 2), wave, lane);
        init_acc_b<4>(acc, vec + V_FEAT + half * 64);
        gemm_b<B_ACT_STEPS, 4>(buf1, acc, lane, [&](int s) { return hb[s]; });
#pragma unroll
        for (int q = 0; q < 64; ++q) hf[q] = acc[q >> 4][q & 15];
        to_b();
    }
    {   // views_linears[0] + rgb head
        const int64_t ray = p / S;
        float dl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        dl[0] = half ? dirs[ray * dirs_stride + 1] : dirs[ray * dirs_stride + 0];
        dl[1] = half ? 0.0f : dirs[ray * dirs_stride + 2];
        const bf16x8 d8 = pack8(dl);
        f32x16 acc[2];
        slabb_sync();
        init_acc_b<2>(acc, vec + V_VIEWS + half * 32);
        gemm_b<B_VIEW_STEPS, 2>(buf0, acc, lane, [&](int s) { return s < 8 ? hb[s < 8 ? s : 0] : d8; });
        float rgb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* wr = vec + V_WR + c * 64 + half * 32;
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 32; ++q) part = fmaf(wr[q], fmaxf(acc[q >> 4][q & 15], 0.0f), part);
            part += __shfl_xor(part, 32);
            rgb[c] = 1.0f / (1.0f + expf(-(part + vec[V_BR + c])));
        }
        if (live && half == 0) *reinterpret_cast<f32x4*>(raw + p_raw * 4) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
    }
}

}  // namespace

extern "C" size_t mvsnerf_mlp_packed_bf16_elems(int F)
{
    if (F < 2 || F > MAX_F || (F & 1)) return 0;
    return layout_b(F).total;
}

extern "C" int mvsnerf_mlp_pack_bf16(const float* const w[11], int F, void* packed_bf16, void* stream)
{
    if (!w || !packed_bf16) return MVSNERF_EINVAL;
    if (F < 2 || F > MAX_F || (F & 1)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed_bf16)) return MVSNERF_EALIGN;
    PackBArgs a;
    for (int i = 0; i < 11; ++i) { if (!w[i]) return MVSNERF_EINVAL; a.w[i] = w[i]; }
    a.F = F;
    mlp_pack_bf16_kernel<<<64, 256, 0, (hipStream_t)stream>>>(a, reinterpret_cast<__bf16*>(packed_bf16));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_mlp_fwd_bf16(const void* packed_bf16, const float* packed_f32, int F, const float* ndc, int ndc_stride,
                                    const float* feat, int feat_stride, const float* dirs, int dirs_stride,
                                    int64_t N, int S, int alpha_only, float* raw, void* stream)
{
    if (!packed_bf16 || !packed_f32 || !ndc || !feat || !raw || N < 0 || S < 1 || feat_stride < F || ndc_stride < 3) return MVSNERF_EINVAL;
    if (!alpha_only && (!dirs || dirs_stride < 3)) return MVSNERF_EINVAL;
    if (F < 2 || F > MAX_F || (F & 1)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed_bf16) || !mvs_aligned16(raw)) return MVSNERF_EALIGN;
    const int64_t P = N * S;
    if (P == 0) return MVSNERF_OK;
    hipStream_t st = (hipStream_t)stream;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_fwd_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, B_LDS_BYTES);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const __bf16* wq = reinterpret_cast<const __bf16*>(packed_bf16);
    if (alpha_only)
        mlp_fwd_bf16_kernel<true><<<mvs_cdiv(P, 128), 256, B_LDS_BYTES, st>>>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw);
    else
        mlp_fwd_bf16_kernel<false><<<mvs_cdiv(P, 128), 256, B_LDS_BYTES, st>>>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
