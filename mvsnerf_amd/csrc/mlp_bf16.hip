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
#include "lds_dma.h"
#include "mlp_layout.h"
#include "mlp_b16_dev.h"
#include "knobs.h"

using namespace mlp;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

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
    lds_dma<4>(dst, src, (int)(n_elems >> 9), wave, lane);           // 1 KB (512 bf16) per wave-instruction (lds_dma.h)
}

__device__ __forceinline__ void slabb_sync()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// acc[nb] += W[block nb] * act over STEPS k-steps of 16: one weight fragment (ds_read_b128) per matrix instruction.  The stream is laid out by hand
// (sched_barrier around every matrix instruction) like the fp16x3 kernel's: the fragment an instruction needs was requested B_AHEAD instructions earlier
// (round 5's compiler-scheduled loop waited for every fragment right after asking for it: 2 650 cycles per 32-instruction layer, profiles/r06_mlp_bf16_census.txt),
// the gap behind every instruction carries one read.
constexpr int B_AHEAD = 3;
template <int STEPS, int NBLK, typename BFN>
__device__ __forceinline__ void gemm_b(const char* __restrict__ w, f32x16 (&acc)[NBLK], int lane, BFN bfn, bool high = false)
{
    constexpr int N = STEPS * NBLK, D = B_AHEAD;
    const bf16x8* __restrict__ fw = reinterpret_cast<const bf16x8*>(w) + lane;
    bf16x8 a[D + 1];
#pragma unroll
    for (int i = 0; i < D && i < N; ++i) a[i] = fw[i * 64];
    // a matrix stream beats the other wave's epilogue (priority 0); two matrix streams of equal priority share the pipe about evenly (measured: 64 cycles per
    // instruction for BOTH waves of a SIMD)
    // (measured: priority 2 for the older waves of the 8-wave kernel - strictly sequential GEMMs - is 3 % SLOWER than equal priorities: `high` is accepted and unused)
    (void)high;
    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int s = i / NBLK, nb = i % NBLK;
        if (i + D < N) a[(i + D) % (D + 1)] = fw[(i + D) * 64];
        __builtin_amdgcn_sched_barrier(0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % (D + 1)], bfn(s), acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
}

// eight fp32 values -> bf16 (round to nearest even), two per v_cvt_pk_bf16_f32 (written element by element the compiler converts each value alone and merges
// pairs with v_perm_b32: twelve instructions instead of four)
__device__ __forceinline__ bf16x8 pack8(const float* v)
{
    typedef unsigned u32x4_b __attribute__((ext_vector_type(4)));
    u32x4_b r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned t;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(v[2 * j]), "v"(v[2 * j + 1]));
        r[j] = t;
    }
    return __builtin_bit_cast(bf16x8, r);
}

// SAVE (bf16 training forward, train_mvs_nerf_pl.py:317-318 `precision=16`): the operands the backward pass consumes are written in
// the slot format of mlp_layout.h as BF16 values (same [tile][slot][lane] order, two bytes per element): the weight-gradient GEMMs
// round them to bf16 when they load them anyway, and 0.64 GB of fp32 slots per 1024 x 128 batch made this kernel, the data-gradient
// kernel and the weight-gradient kernels stream at the HBM rate (4.6 / 5.0 / 3.5 TB/s) instead of computing.  What autocast itself
// saves for backward is 16-bit as well.
template <bool ALPHA_ONLY, bool SAVE = false>
__global__ __launch_bounds__(256, 2) void mlp_fwd_bf16_kernel(
    const __bf16* __restrict__ wq, const float* __restrict__ packed_f32, int F, const float* __restrict__ ndc, int ndc_stride,
    const float* __restrict__ feat, int feat_stride, const float* __restrict__ dirs, int dirs_stride,
    int64_t P, int S, float* __restrict__ raw, float* __restrict__ saved = nullptr)
{
    static_assert(!(SAVE && ALPHA_ONLY), "the training forward computes colours too");
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
#ifdef BF_CENSUS      // DEV probe: shader-clock stamps of this wave's phases, 32 per tile, behind the results (scratch/r6/bf_census.py allocates them)
    unsigned* cen = reinterpret_cast<unsigned*>(raw + P * (ALPHA_ONLY ? 1 : 4)) + ((int64_t)blockIdx.x * 4 + wave) * 32;
    int cen_i = 0;
#define BF_STAMP() do { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if (lane == 0 && cen_i < 32) cen[cen_i] = (unsigned)t__; ++cen_i; } while (0)
#else
#define BF_STAMP() do {} while (0)
#endif
    BF_STAMP();
    __bf16* sv = nullptr;
    if (SAVE) sv = reinterpret_cast<__bf16*>(saved) + ((int64_t)blockIdx.x * 4 + wave) * (SLOTS_SAVED * 64) + lane;
    auto save = [&](int slot, float v) { if (SAVE) sv[slot * 64] = (__bf16)v; };

    // slab 0 = pts_bias weights + layer 0 (contiguous in the packed buffer)
    slabb_dma(buf0, wq + L.featw, L.l1 - L.featw, wave, lane);
    for (int i = tid; i < V_TOTAL; i += 256) vec[i] = packed_f32[LF.vec + i];
    const float px = ndc[p * ndc_stride + 0], py = ndc[p * ndc_stride + 1], pz = ndc[p * ndc_stride + 2];
    float fv[24];                                 // F/2 <= 20 feature operands of this lane half (the store keeps the first 16: F <= 32 when training)
    {
        const float* fp = feat + p * feat_stride + half * (F / 2);
        // (the compiler turns this into one scalar branch + one load per element; issuing all 24 unconditionally - padding slots re-reading element 0 - measured
        // SLOWER, 34.4 -> 41.9 us in the bf16 kernel: a 64-lane dword load at an 80-byte stride is ~20 cache lines per instruction, the address unit is what waits)
#pragma unroll
        for (int i = 0; i < 24; ++i) fv[i] = i < F / 2 ? fp[i] : 0.0f;
    }
    // the view direction of the point's ray: asked for HERE, used by the last GEMM (loaded there its latency sat in front of the rgb head: ~1 000 cycles per wave)
    float dl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (!ALPHA_ONLY) {
        const int64_t ray = ((uint64_t)p >> 32) == 0 ? (int64_t)((unsigned)p / (unsigned)S) : p / S;      // (a 64-bit division is ~10x a 32-bit one)
        dl[0] = half ? dirs[ray * dirs_stride + 1] : dirs[ray * dirs_stride + 0];
        dl[1] = half ? 0.0f : dirs[ray * dirs_stride + 2];
    }
    bf16x8 pe8[B_PE_STEPS];                      // positional-encoding operands (layer 0, reused by layer 5); computed BEHIND the pts_bias GEMM
    bf16x8 fb8[3];                                // the feature operands
#pragma unroll
    for (int s = 0; s < 3; ++s) fb8[s] = pack8(fv + 8 * s);
    if (SAVE) {
#pragma unroll
        for (int t = 0; t < 16; ++t) save(S_FV + t, fv[t]);
    }
    float bias[64];
    bf16x8 hb[8];                                 // the current activations as bf16 B operands (k-step s = values 8s .. 8s+7 of the lane)
    // epilogue of a layer: v = act(acc [* bias]) goes straight into hb, eight values at a time (no fp32 copy of the layer lives on: with the hand-laid GEMM stream
    // the epilogue no longer overlaps the GEMM of the SAME wave - the other workgroup's wave on the SIMD does - and 64 more live registers spilled)
    auto finish_b = [&](f32x16 (&acc)[4], bool modulated_relu, int save_slot, const float* head_w, float& head_sum) {
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float v8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = 8 * s + j;
                const float x = acc[q >> 4][q & 15];
                v8[j] = modulated_relu ? fmaxf(x * bias[q], 0.0f) : x;
                save(save_slot + q, v8[j]);
                if (head_w) head_sum = fmaf(head_w[q], v8[j], head_sum);
            }
            hb[s] = pack8(v8);
        }
    };
    float no_head = 0.0f;

    slabb_sync();
    BF_STAMP();
    slabb_dma(buf1, wq + L.l1, b_seg(B_ACT_STEPS, 4), wave, lane);
    {   // bias = pts_bias(feat)
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_BIASG + half * 64);
        auto fb = [&](int s) { return fb8[s]; };
        if (L.fsteps == 1) gemm_b<1, 4>(buf0, acc, lane, fb);
        else if (L.fsteps == 2) gemm_b<2, 4>(buf0, acc, lane, fb);
        else gemm_b<3, 4>(buf0, acc, lane, fb);
        BF_STAMP();
#pragma unroll
        for (int q = 0; q < 64; ++q) { bias[q] = acc[q >> 4][q & 15]; save(S_BM + q, bias[q]); }
    }
    // the encoding (transcendental unit, exact range reduction: mlp_b16_dev.h), between the first two GEMMs: it runs beside the matrix work of the other
    // workgroup's wave on this SIMD (in front of the first barrier it measured 1 us slower)
#pragma unroll
    for (int s = 0; s < B_PE_STEPS; ++s) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { t8[j] = pe_op_hw(8 * s + j, half, px, py, pz); save(S_E + 8 * s + j, t8[j]); }
        trans_fence8(t8);
        pe8[s] = pack8(t8);
    }

    {   // layer 0
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + half * 64);
        gemm_b<B_PE_STEPS, 4>(buf0 + b_seg(L.fsteps, 4) * 2, acc, lane, [&](int s) { return pe8[s]; });
        BF_STAMP();
        finish_b(acc, true, S_H, nullptr, no_head);
        BF_STAMP();
    }
    // layers 1..4: slabs alternate buf1, buf0, buf1, buf0
#pragma unroll 1
    for (int layer = 1; layer <= 4; ++layer) {
        char* cur = (layer & 1) ? buf1 : buf0;
        char* nxt = (layer & 1) ? buf0 : buf1;
        slabb_sync();
        BF_STAMP();
        if (layer < 4) slabb_dma(nxt, wq + L.l1 + (size_t)layer * b_seg(B_ACT_STEPS, 4), b_seg(B_ACT_STEPS, 4), wave, lane);
        else slabb_dma(nxt, wq + L.l5a, b_seg(B_PE_STEPS, 4), wave, lane);           // after layer 4 (in buf0): L5a -> buf1
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + 128 * layer + half * 64);
        gemm_b<B_ACT_STEPS, 4>(cur, acc, lane, [&](int s) { return hb[s]; });
        BF_STAMP();
        finish_b(acc, true, S_H + layer * 64, nullptr, no_head);
        BF_STAMP();
    }
    float sigma;
    {   // layer 5 on cat([pts, h4]): L5a in buf1, L5b -> buf0
        f32x16 acc[4];
        slabb_sync();
        BF_STAMP();
        slabb_dma(buf0, wq + L.l5b, b_seg(B_ACT_STEPS, 4), wave, lane);
        init_acc_b<4>(acc, vec + V_L0 + 128 * 5 + half * 64);
        gemm_b<B_PE_STEPS, 4>(buf1, acc, lane, [&](int s) { return pe8[s]; });
        BF_STAMP();
        slabb_sync();
        BF_STAMP();
        if (!ALPHA_ONLY) slabb_dma(buf1, wq + L.feat, b_seg(B_ACT_STEPS, 4), wave, lane);
        gemm_b<B_ACT_STEPS, 4>(buf0, acc, lane, [&](int s) { return hb[s]; });
        BF_STAMP();
        float part = 0.0f;
        finish_b(acc, true, S_H + 5 * 64, vec + V_WA + half * 64, part);       // alpha_linear on the fp32 activations, as they are produced
        part += __shfl_xor(part, 32);
        sigma = fmaxf(part + vec[V_BA], 0.0f);
        BF_STAMP();
    }
    if (ALPHA_ONLY) {
        if (live && half == 0) raw[p_raw] = sigma;
        return;
    }
    {   // feature_linear (buf1), then views -> buf0
        f32x16 acc[4];
        slabb_sync();
        BF_STAMP();
        slabb_dma(buf0, wq + L.views, b_seg(B_VIEW_STEPS, 2), wave, lane);
        init_acc_b<4>(acc, vec + V_FEAT + half * 64);
        gemm_b<B_ACT_STEPS, 4>(buf1, acc, lane, [&](int s) { return hb[s]; });
        BF_STAMP();
        finish_b(acc, false, S_FE, nullptr, no_head);
    }
    {   // views_linears[0] + rgb head
        const bf16x8 d8 = pack8(dl);
        f32x16 acc[2];
        slabb_sync();
        BF_STAMP();
        init_acc_b<2>(acc, vec + V_VIEWS + half * 32);
        gemm_b<B_VIEW_STEPS, 2>(buf0, acc, lane, [&](int s) { return s < 8 ? hb[s < 8 ? s : 0] : d8; });
        BF_STAMP();
        if (SAVE) {
#pragma unroll
            for (int q = 0; q < 32; ++q) save(S_HV + q, fmaxf(acc[q >> 4][q & 15], 0.0f));
            save(S_DR + 0, dl[0]);
            save(S_DR + 1, dl[1]);
        }
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
        BF_STAMP();
    }
}


// ------------------------------------------------------------------------------------------ inference kernel (round 6)
// The same arithmetic with EIGHT waves (256 points) per workgroup - one slab serves twice the points of the 4-wave kernel above, which stays for the training
// forward with its activation store - and NO separate epilogue: a bf16 layer is 32 matrix instructions against ~200 VALU of finishing (modulation, ReLU,
// conversion), and the two waves of a SIMD run both phases together (census: GEMMs at 64 cycles per instruction each, then both epilogues, then the barrier:
// 3 700 cycles per layer for 2 048 of matrix work).  Here the finishing of layer l is done INSIDE the GEMM of layer l + 1: the B operand of k-step s is
// act(acc_l[8s .. 8s+7]) converted in the gaps behind the matrix instructions of k-step s - 1, straight from the previous layer's accumulators, which stay live
// (two accumulator sets alternate; the bf16 copy of the activations and the fp32 scratch of the 4-wave kernel are gone).
#ifndef BP_WAVES_N
#define BP_WAVES_N 8
#endif
constexpr int BP_WAVES = BP_WAVES_N;             // 8: one workgroup per CU; 4: two (DEV: -DBP_WAVES_N=4)
constexpr int BP_THREADS = 64 * BP_WAVES;

__device__ __forceinline__ void slabp_dma(char* __restrict__ dst, const __bf16* __restrict__ src, size_t n_elems, int wave, int lane)
{
    lds_dma<BP_WAVES>(dst, src, (int)(n_elems >> 9), wave, lane);
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b)
{
    unsigned t;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b));
    return t;
}

// acc[nb] += W[block nb] * B over STEPS k-steps.  B of k-step 0 = `first`; B of k-step s + 1 is produced pair by pair by next(s + 1, pair) -> packed bf16 pair,
// in the gaps behind the matrix instructions of k-step s (pairs 0, 1 behind the first, 2 behind the second, 3 behind the third: the last conversion is a matrix
// instruction and a ds_read away from its reader - the compiler does not know the asm writes a VGPR a matrix instruction reads).
template <int STEPS, int NBLK, typename NEXT>
__device__ __forceinline__ void gemm_bf(const char* __restrict__ w, f32x16 (&acc)[NBLK], int lane, bf16x8 first, NEXT next)
{
    typedef unsigned u32x4_b __attribute__((ext_vector_type(4)));
    constexpr int N = STEPS * NBLK, D = B_AHEAD;
    const bf16x8* __restrict__ fw = reinterpret_cast<const bf16x8*>(w) + lane;
    bf16x8 a[D + 1];
    u32x4_b b[2];
    b[0] = __builtin_bit_cast(u32x4_b, first);
#pragma unroll
    for (int i = 0; i < D && i < N; ++i) a[i] = fw[i * 64];
    __builtin_amdgcn_s_setprio(1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int s = i / NBLK, nb = i % NBLK;
        if (i + D < N) a[(i + D) % (D + 1)] = fw[(i + D) * 64];
        __builtin_amdgcn_sched_barrier(0);
        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i % (D + 1)], __builtin_bit_cast(bf16x8, b[s & 1]), acc[nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < STEPS) {
            u32x4_b& n = b[(s + 1) & 1];
            if (NBLK == 4) {
                if (nb == 0) { n[0] = next(s + 1, 0); n[1] = next(s + 1, 1); }
                else if (nb == 1) n[2] = next(s + 1, 2);
                else if (nb == 2) n[3] = next(s + 1, 3);
            } else if (nb == 0) {
#pragma unroll
                for (int j = 0; j < 4; ++j) n[j] = next(s + 1, j);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_setprio(0);
}

template <bool ALPHA_ONLY>
__global__ __launch_bounds__(BP_THREADS, BP_WAVES == 4 ? 2 : 1) void mlp_fwd_bf16_pair_kernel(
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
    const int64_t p_raw = ((int64_t)blockIdx.x * BP_WAVES + wave) * 32 + (lane & 31);
    const bool live = p_raw < P;
    const int64_t p = live ? p_raw : P - 1;
    constexpr size_t ACT = (size_t)B_ACT_STEPS * 4 * 512, PEW = (size_t)B_PE_STEPS * 4 * 512;      // bf16 elements of a slab

    slabp_dma(buf0, wq + L.featw, L.l1 - L.featw, wave, lane);                       // slab 0 = pts_bias weights + layer 0
    for (int i = tid; i < V_TOTAL; i += BP_THREADS) vec[i] = packed_f32[LF.vec + i];
    const float px = ndc[p * ndc_stride + 0], py = ndc[p * ndc_stride + 1], pz = ndc[p * ndc_stride + 2];
    float dl0 = 0.0f, dl1 = 0.0f;                                                     // the view direction: asked for here, used by the last GEMM
    if (!ALPHA_ONLY) {
        const int64_t ray = ((uint64_t)p >> 32) == 0 ? (int64_t)((unsigned)p / (unsigned)S) : p / S;      // (a 64-bit division is ~10x a 32-bit one)
        dl0 = half ? dirs[ray * dirs_stride + 1] : dirs[ray * dirs_stride + 0];
        dl1 = half ? 0.0f : dirs[ray * dirs_stride + 2];
    }
    bf16x8 fb8[3];
    {
        float fv[24];
        // (buf1 is idle until layer 1's slab is asked for behind the first barrier: 4 KB of it per wave stage the features)
        if (!stage_features(fv, feat, feat_stride, F, ((int64_t)blockIdx.x * BP_WAVES + wave) * 32, P, buf1 + wave * 4096, lane)) {
            const float* fp = feat + p * feat_stride + half * (F / 2);
#pragma unroll
            for (int i = 0; i < 24; ++i) fv[i] = i < F / 2 ? fp[i] : 0.0f;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) fb8[s] = pack8(fv + 8 * s);
    }
    float bias[64];
    bf16x8 pe8[B_PE_STEPS];
    f32x16 accA[4], accB[4];
    // operand pairs for gemm_bf: precomputed (features, encoding) or finished on the fly from the previous layer's accumulators
    auto from8 = [&](const bf16x8* arr) { return [arr](int s, int j) { return __builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, arr[s])[j]; }; };
    auto relu_of = [&](f32x16 (&prev)[4]) {
        return [&](int s, int j) {
            const int q = 8 * s + 2 * j;
            return cvt_pk_bf16(fmaxf(prev[q >> 4][q & 15] * bias[q], 0.0f), fmaxf(prev[(q + 1) >> 4][(q + 1) & 15] * bias[q + 1], 0.0f));
        };
    };
    auto first_relu = [&](f32x16 (&prev)[4]) {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = fmaxf(prev[0][j] * bias[j], 0.0f);
        return pack8(v8);
    };

    slabb_sync();
    slabp_dma(buf1, wq + L.l1, ACT, wave, lane);                                     // layer 1 -> buf1
    {   // bias = pts_bias(feat)
        init_acc_b<4>(accA, vec + V_BIASG + half * 64);
        if (L.fsteps == 1) gemm_bf<1, 4>(buf0, accA, lane, fb8[0], from8(fb8));
        else if (L.fsteps == 2) gemm_bf<2, 4>(buf0, accA, lane, fb8[0], from8(fb8));
        else gemm_bf<3, 4>(buf0, accA, lane, fb8[0], from8(fb8));
#pragma unroll
        for (int q = 0; q < 64; ++q) bias[q] = accA[q >> 4][q & 15];
    }
#pragma unroll
    for (int s = 0; s < B_PE_STEPS; ++s) {                                            // the encoding (transcendental unit)
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = pe_op_hw(8 * s + j, half, px, py, pz);
        trans_fence8(t8);
        pe8[s] = pack8(t8);
    }
    init_acc_b<4>(accA, vec + V_L0 + half * 64);
    gemm_bf<B_PE_STEPS, 4>(buf0 + b_seg(L.fsteps, 4) * 2, accA, lane, pe8[0], from8(pe8));      // layer 0 (same slab) -> A
    // layers 1..4: layer l reads the finished values of the other accumulator set; behind the GEMM in buffer X, X takes the slab after next
    bf16x8 first;
#define BP_LAYER(LAYER, CUR, PREV, NEXT_DMA)                                                                  \
    init_acc_b<4>(CUR, vec + V_L0 + 128 * (LAYER) + half * 64);                                               \
    first = first_relu(PREV);                                                                                 \
    slabb_sync();                                                                                             \
    NEXT_DMA;                                                                                                 \
    gemm_bf<B_ACT_STEPS, 4>(((LAYER) & 1) ? buf1 : buf0, CUR, lane, first, relu_of(PREV));
    BP_LAYER(1, accB, accA, slabp_dma(buf0, wq + L.l1 + 1 * ACT, ACT, wave, lane))              // layer 2 -> buf0
    BP_LAYER(2, accA, accB, slabp_dma(buf1, wq + L.l1 + 2 * ACT, ACT, wave, lane))              // layer 3 -> buf1
    BP_LAYER(3, accB, accA, slabp_dma(buf0, wq + L.l1 + 3 * ACT, ACT, wave, lane))              // layer 4 -> buf0
    BP_LAYER(4, accA, accB, slabp_dma(buf1, wq + L.l5a, PEW, wave, lane))                       // layer 5, encoding part -> buf1
#undef BP_LAYER
    // layer 5 on cat([pts, h4]) -> B: encoding part (buf1), then the h4 part (buf0) finished from A
    init_acc_b<4>(accB, vec + V_L0 + 128 * 5 + half * 64);
    first = first_relu(accA);
    slabb_sync();
    slabp_dma(buf0, wq + L.l5b, ACT, wave, lane);
    gemm_bf<B_PE_STEPS, 4>(buf1, accB, lane, pe8[0], from8(pe8));
    slabb_sync();
    if (!ALPHA_ONLY) slabp_dma(buf1, wq + L.feat, ACT, wave, lane);
    gemm_bf<B_ACT_STEPS, 4>(buf0, accB, lane, first, relu_of(accA));
    // h5 = relu(B * bias): alpha_linear reads it in fp32; feature_linear gets it finished on the fly like every other layer
    float sigma;
    {
        const float* wa = vec + V_WA + half * 64;
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 64; ++q) part = fmaf(wa[q], fmaxf(accB[q >> 4][q & 15] * bias[q], 0.0f), part);
        part += __shfl_xor(part, 32);
        sigma = fmaxf(part + vec[V_BA], 0.0f);
    }
    if (ALPHA_ONLY) {
        if (live && half == 0) raw[p_raw] = sigma;
        return;
    }
    init_acc_b<4>(accA, vec + V_FEAT + half * 64);
    first = first_relu(accB);
    slabb_sync();
    slabp_dma(buf0, wq + L.views, b_seg(B_VIEW_STEPS, 2), wave, lane);
    gemm_bf<B_ACT_STEPS, 4>(buf1, accA, lane, first, relu_of(accB));                 // feature_linear (no activation) -> A
    // views_linears[0] on [feature | direction]: the feature values go in as they are
    f32x16 av[2];
    init_acc_b<2>(av, vec + V_VIEWS + half * 32);
    {
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = accA[0][j];
        first = pack8(v8);
    }
    const unsigned d01 = cvt_pk_bf16(dl0, dl1);
    slabb_sync();
    gemm_bf<B_VIEW_STEPS, 2>(buf0, av, lane, first, [&](int s, int j) {
        if (s < 8) { const int q = 8 * s + 2 * j; return cvt_pk_bf16(accA[q >> 4][q & 15], accA[(q + 1) >> 4][(q + 1) & 15]); }
        return j == 0 ? d01 : 0u;
    });
    float rgb[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* wr = vec + V_WR + c * 64 + half * 32;
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 32; ++q) part = fmaf(wr[q], fmaxf(av[q >> 4][q & 15], 0.0f), part);
        part += __shfl_xor(part, 32);
        rgb[c] = 1.0f / (1.0f + expf(-(part + vec[V_BR + c])));
    }
    if (live && half == 0) *reinterpret_cast<f32x4*>(raw + p_raw * 4) = f32x4{rgb[0], rgb[1], rgb[2], sigma};
}


// ==========================================================================================================
// Split-bf16 variant ("bf16x6"): fp32-grade results from the bf16 matrix cores.
// Every fp32 operand is written as the sum of NS bf16 pieces (x = x0 + x1 + x2, each piece the bf16 rounding of what is left),
// and a product a*w is accumulated (in fp32, by the MFMA) from the piece products whose combined order is < NS:
//   NS = 3:  a0w0 + a0w1 + a1w0 + a1w1 + a0w2 + a2w0      (dropped terms are <= 2^-24 relative)   -> 6 MFMAs per k-step
//   NS = 2:  a0w0 + a0w1 + a1w0                            (<= 2^-16)                              -> 3 MFMAs
//   NS = 1:  plain bf16                                                                             -> 1 MFMA
// 6 bf16 MFMAs cost 6/16 of one fp32 MFMA k-step's time for the same inputs.  Weights are split at pack time (NS planes),
// activations in the layer epilogue (v_cvt_pk_bf16_f32 + subtract).  Weight planes stream through LDS in slabs of two
// k-steps (NS * 8 KB for a 128-wide layer), double-buffered by LDS-DMA, in exactly the order the kernel consumes them, so the
// stream is one running pointer.  Everything outside the GEMMs (positional encoding, bias modulation, ReLU, heads) is fp32.
// ==========================================================================================================
__host__ __device__ constexpr int sp_slab_elems(int ns, int nb) { return ns * 2 * nb * 64 * 8; }
__host__ __device__ inline int sp_slabs(int steps) { return (steps + 1) / 2; }
__host__ __device__ inline size_t sp_total_elems(int ns, int F)
{
    (void)F;
    return (size_t)(2 + 2 + 4 * 4 + 2 + 4 + 4) * sp_slab_elems(ns, 4) + (size_t)5 * sp_slab_elems(ns, 2);
}

__device__ inline void pack_split_segment(__bf16* __restrict__ dst, const float* __restrict__ W, int ld, int col_off, int kmap,
                                          int steps, int nb, int ns, int F, int tid, int nthreads)
{
    const int slabs = sp_slabs(steps);
    const int per_slab = sp_slab_elems(ns, nb);
    const int total = slabs * per_slab;
    for (int i = tid; i < total; i += nthreads) {
        const int sl = i / per_slab, r = i - sl * per_slab;
        const int j = r & 7, lane = (r >> 3) & 63, rest = r >> 9;                // rest = (plane*2 + s_local)*nb + b
        const int b = rest % nb, ps = rest / nb, s_local = ps & 1, plane = ps >> 1;
        const int st = 2 * sl + s_local;
        const int col = st < steps ? b_col(kmap, 8 * st + j, lane >> 5, F) : -1;
        float w = col < 0 ? 0.0f : W[(size_t)(b * 32 + (lane & 31)) * ld + col_off + col];
        __bf16 piece = (__bf16)w;
        for (int k = 0; k < plane; ++k) { w -= (float)piece; piece = (__bf16)w; }
        dst[i] = piece;
    }
}

__global__ __launch_bounds__(256) void mlp_pack_split_kernel(PackBArgs a, int ns, __bf16* __restrict__ packed)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const size_t s4 = sp_slab_elems(ns, 4);
    size_t o = 0;
    pack_split_segment(packed + o, a.w[6], a.F, 0, K_FEAT, 4, 4, ns, a.F, tid, nt);                         o += 2 * s4;   // always 2 slabs (zero-padded)
    pack_split_segment(packed + o, a.w[0], PE_DIM, 0, K_PE, B_PE_STEPS, 4, ns, a.F, tid, nt);               o += 2 * s4;
    for (int l = 1; l <= 4; ++l) { pack_split_segment(packed + o, a.w[l], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, ns, a.F, tid, nt); o += 4 * s4; }
    pack_split_segment(packed + o, a.w[5], WIDTH + PE_DIM, 0, K_PE, B_PE_STEPS, 4, ns, a.F, tid, nt);       o += 2 * s4;
    pack_split_segment(packed + o, a.w[5], WIDTH + PE_DIM, PE_DIM, K_ACT, B_ACT_STEPS, 4, ns, a.F, tid, nt); o += 4 * s4;
    pack_split_segment(packed + o, a.w[7], WIDTH, 0, K_ACT, B_ACT_STEPS, 4, ns, a.F, tid, nt);              o += 4 * s4;
    pack_split_segment(packed + o, a.w[9], WIDTH + 3, 0, K_VIEWS, B_VIEW_STEPS, 2, ns, a.F, tid, nt);
}

__host__ __device__ constexpr int split_nbuf(int ns) { return ns == 3 ? 2 : 3; }
constexpr int IL_NBUF = 4;           // interleaved schedule: slabs k+1 and k+2 in flight behind the one being consumed

template <int NS> struct Pieces { bf16x8 p[NS]; };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Split 8 fp32 values into NS bf16 pieces each.  For NS == 3 the pieces are taken by TRUNCATION (top 16 bits of the fp32 word):
// the remainder after each truncation has at most 16, then 8 significant bits, so x = p0 + p1 + p2 holds exactly (not just to
// rounding), and the work is full-rate bit operations (and / perm / packed subtract) instead of float->bf16 conversions.
// Fewer pieces keep round-to-nearest (v_cvt_pk_bf16_f32), where the rounding of the last piece matters.
template <int NS>
__device__ __forceinline__ Pieces<NS> split8(const float* v)
{
    Pieces<NS> r;
    float rem[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) rem[j] = v[j];
    if constexpr (NS == 3) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            u32x4 w;
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const unsigned lo = __builtin_bit_cast(unsigned, rem[2 * pr]) & 0xffff0000u;
                const unsigned hi = __builtin_bit_cast(unsigned, rem[2 * pr + 1]) & 0xffff0000u;
                w[pr] = hi | (lo >> 16);
                if (k + 1 < NS) {
                    rem[2 * pr] -= __builtin_bit_cast(float, lo);
                    rem[2 * pr + 1] -= __builtin_bit_cast(float, hi);
                }
            }
            r.p[k] = __builtin_bit_cast(bf16x8, w);
        }
    } else {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const __bf16 h = (__bf16)rem[j];
                r.p[k][j] = h;
                if (k + 1 < NS) rem[j] -= (float)h;
            }
        }
    }
    return r;
}

// Slab stream: slab k of the packed buffer lives in buf[k % 3]; while slab k is consumed, slabs k+1 and k+2 are in flight
// (a slab is only 2 k-steps = ~1500 MFMA cycles of work per wave, less than the L2 latency of its own DMA, so a prefetch
// distance of one slab would expose that latency 35 times per tile).
struct SlabStream {
    const char* g;          // next slab to fetch
    char* base; int stride; // buffer i = base + i*stride, i in 0..nbuf-1
    int cur;                // buffer `cur`: the slab about to be consumed
    int nbuf;               // number of LDS buffers = slabs in flight + 1
    int k, n4, n2;          // (interleaved schedule) stream position: slab k is in buffer `cur`; the stream is n4 slabs of `stride`
                            // bytes followed by n2 slabs of stride/2 bytes
    __device__ __forceinline__ int bytes_of(int idx) const { return idx < n4 ? stride : (idx < n4 + n2 ? stride >> 1 : 0); }
    __device__ __forceinline__ char* buf(int i) const { return base + i * stride; }
    __device__ __forceinline__ void prefetch(int bytes, int wave, int lane)
    {
        int into = cur + nbuf - 1; if (into >= nbuf) into -= nbuf;
        if (bytes > 0) slabb_dma(buf(into), reinterpret_cast<const __bf16*>(g), (size_t)bytes >> 1, wave, lane);
        g += bytes;
    }
};

// wait until at most `newer` of this wave's DMA instructions are outstanding (= the slab about to be used has landed), then
// make it visible to / release the oldest buffer from all four waves
template <int NEWER>
__device__ __forceinline__ void slab_wait()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NEWER) : "memory");
    __syncthreads();
}

// One segment = SLABS slabs of two k-steps each.  ahead1 / ahead2: byte sizes of the first and second slab AFTER this segment
// (0 = end of the stream); they decide what is prefetched and how many newer DMA instructions may stay in flight.
template <int NS, int SLABS, int NBLK, int AHEAD1, int AHEAD2, int VALU_PER_MFMA, typename BFN>
__device__ __forceinline__ void run_segment(SlabStream& st, f32x16 (&acc)[NBLK], int wave, int lane, BFN bfn)
{
    constexpr int SLAB_BYTES = sp_slab_elems(NS, NBLK) * 2;
    constexpr int NPROD = NS == 3 ? 6 : (NS == 2 ? 3 : 1);
    constexpr bool LEAN = NS == 3;
    constexpr int NBUF = split_nbuf(NS);
    // the B pieces of k-step s+1 are produced (VALU) while the MFMAs of k-step s issue: a wave issues in order, so the
    // split/encoding work only overlaps the matrix pipe if it sits BETWEEN the MFMAs in program order
    Pieces<NS> cur = bfn(0);
#pragma unroll
    for (int sl = 0; sl < SLABS; ++sl) {
        const int nxt1 = sl + 1 < SLABS ? SLAB_BYTES : AHEAD1;                       // slab k+1
        const int nxt2 = sl + 2 < SLABS ? SLAB_BYTES : (sl + 1 < SLABS ? AHEAD1 : AHEAD2);   // slab k+2
        if (NBUF == 2) {
            slab_wait<0>();                          // slab k has landed; k+1 is requested now
            st.prefetch(nxt1, wave, lane);
        } else {
            // slab k+1 was requested one step ago and may stay in flight: its DMA instructions per wave = bytes / 1 KB / 4 waves
            if (nxt1 == 0) slab_wait<0>();
            else if (nxt1 / 4096 == 2 * NS) slab_wait<2 * NS>();
            else slab_wait<NS>();
            st.prefetch(nxt2, wave, lane);
        }
        const char* w = st.buf(st.cur);
#pragma unroll
        for (int s_local = 0; s_local < 2; ++s_local) {
            const int s = 2 * sl + s_local;
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};      // piece products, smallest first
            if (LEAN) {
                // two waves per SIMD (the other wave's MFMAs cover this wave's VALU work): keep the register footprint minimal
                const Pieces<NS> b = s == 0 ? cur : bfn(s);
#pragma unroll
                for (int nb = 0; nb < NBLK; nb += 2) {           // two accumulators alternate: consecutive MFMAs are independent
                    bf16x8 a[NS][2];
#pragma unroll
                    for (int k = 0; k < NS; ++k)
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            a[k][e] = *reinterpret_cast<const bf16x8*>(w + ((((k * 2 + s_local) * NBLK) + nb + e) * 64 + lane) * 16);
#pragma unroll
                    for (int pr = 6 - NPROD; pr < 6; ++pr)
#pragma unroll
                        for (int e = 0; e < 2; ++e)
                            acc[nb + e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[pr] < NS ? PA[pr] : 0][e], b.p[PB[pr] < NS ? PB[pr] : 0], acc[nb + e], 0, 0, 0);
                }
            } else {
                Pieces<NS> nxt = cur;
                if (s + 1 < 2 * SLABS) nxt = bfn(s + 1);
                bf16x8 a[NS][NBLK];
#pragma unroll
                for (int k = 0; k < NS; ++k)
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb)
                        a[k][nb] = *reinterpret_cast<const bf16x8*>(w + ((((k * 2 + s_local) * NBLK) + nb) * 64 + lane) * 16);
                // the NBLK accumulators rotate so that consecutive MFMAs are independent
#pragma unroll
                for (int pr = 6 - NPROD; pr < 6; ++pr)
#pragma unroll
                    for (int nb = 0; nb < NBLK; ++nb)
                        acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[pr] < NS ? PA[pr] : 0][nb], cur.p[PB[pr] < NS ? PB[pr] : 0], acc[nb], 0, 0, 0);
                cur = nxt;
            }
        }
        st.cur = st.cur + 1 == NBUF ? 0 : st.cur + 1;
    }
}

// ---- interleaved schedule (one wave per SIMD, three buffers).  A wave issues in order and an MFMA blocks issue until the
// matrix pipe is free, so VALU work only overlaps the pipe if it sits BETWEEN MFMAs in program order.  Here the split of
// k-step s+1's operands (and, for the embedding layers, their sin/cos) is cut into 12 micro-ops (plane x value pair) that are
// placed after every other MFMA of k-step s, and the A fragments of the next output block are requested one block ahead.
// __builtin_amdgcn_sched_barrier(0) pins that order.  The readiness barrier of slab k+1 sits in the MIDDLE of slab k
// (prefetch distance 1.5 slabs) so that the first fragments of slab k+1 can be requested before slab k's last MFMAs.
template <int NS, int NBLK>
__device__ __forceinline__ void load_a(bf16x8 (&a)[NS], const char* w, int s_local, int nb, int lane)
{
#pragma unroll
    for (int k = 0; k < NS; ++k) a[k] = *reinterpret_cast<const bf16x8*>(w + ((((k * 2 + s_local) * NBLK) + nb) * 64 + lane) * 16);
}

template <int NS, int SLABS, int NBLK, int AHEAD1, int AHEAD2, typename VFN>
__device__ __forceinline__ void run_segment_il(SlabStream& st, f32x16 (&acc)[NBLK], int wave, int lane, VFN vfn)
{
    static_assert(NS == 3, "interleaved schedule is written for the 3-piece split");
    constexpr int SLAB_BYTES = sp_slab_elems(NS, NBLK) * 2;
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    constexpr int NSTEPS = 2 * SLABS;
    Pieces<NS> cur;
    {   // operands of k-step 0 (exposed: they depend on the previous layer's epilogue)
        float v8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v8[j] = vfn(0, j);
        cur = split8<NS>(v8);
    }
    // entry contract: slab `cur` of this segment has landed and is visible (the previous segment's mid-slab barrier, or the
    // kernel prologue, took care of it); slab cur+1 has been requested.
    // MFMA order within a k-step: the six piece products are the OUTER loop and the NBLK accumulators the inner one, so that
    // consecutive MFMAs never wait on each other's result:  (w2,b0) (w0,b2) (w1,b1) (w1,b0) (w0,b1) (w0,b0), each over all blocks.
    // Weight plane 2 is dead after the first group and plane 1 lives for two, so at most two planes (+ the next step's two
    // prefetched ones) are in registers.
    auto load_plane = [&](bf16x8 (&f)[NBLK], const char* w, int s_local, int plane) {
#pragma unroll
        for (int nb = 0; nb < NBLK; ++nb) f[nb] = *reinterpret_cast<const bf16x8*>(w + ((((plane * 2 + s_local) * NBLK) + nb) * 64 + lane) * 16);
    };
    bf16x8 f0[NBLK], f1[NBLK], f2[NBLK], n0[NBLK], n2[NBLK];
    load_plane(f2, st.buf(st.cur), 0, 2);
    load_plane(f0, st.buf(st.cur), 0, 0);
#pragma unroll
    for (int sl = 0; sl < SLABS; ++sl) {
        const char* w = st.buf(st.cur);
        const int ncur = st.cur + 1 == st.nbuf ? 0 : st.cur + 1;
#pragma unroll
        for (int s_local = 0; s_local < 2; ++s_local) {
            const int s = 2 * sl + s_local;
            const bool has_next = s + 1 < NSTEPS;
            Pieces<NS> nxt = cur;
            float rem[8];
            if (s_local == 1) {
                // mid-slab: slab k+1 must have landed before its fragments are requested below (slabs k+2 .. k+nbuf-2 may stay in
                // flight: DMA instructions per wave = bytes / 1 KB / 4 waves); then slab k+nbuf-1 may overwrite the buffer of slab
                // k-1 (every wave is past it)
                if (st.bytes_of(st.k + 1) != 0) {
                    const int newer = st.nbuf >= 4 ? st.bytes_of(st.k + 2) : 0;
                    if (newer == 0) slab_wait<0>();
                    else if (newer == st.stride) slab_wait<2 * NS>();
                    else slab_wait<NS>();
                    st.prefetch(st.bytes_of(st.k + st.nbuf - 1), wave, lane);
                }
            }
            load_plane(f1, w, s_local, 1);
            const bool more = s_local == 0 || (sl + 1 < SLABS);       // next k-step of this segment: same slab, or step 0 of the next
#pragma unroll
            for (int grp = 0; grp < 6; ++grp) {
                if (grp == 4 && more) {
                    if (s_local == 0) { load_plane(n2, w, 1, 2); load_plane(n0, w, 1, 0); }
                    else { load_plane(n2, st.buf(ncur), 0, 2); load_plane(n0, st.buf(ncur), 0, 0); }
                }
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) {
                    const bf16x8 wa = grp == 0 ? f2[nb] : (grp == 2 || grp == 3 ? f1[nb] : f0[nb]);
                    const bf16x8 bb = grp == 1 ? cur.p[2] : (grp == 2 || grp == 4 ? cur.p[1] : cur.p[0]);
                    acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa, bb, acc[nb], 0, 0, 0);
                    const int slot = grp * NBLK + nb;
                    if (has_next && (slot % (NBLK == 4 ? 2 : 1)) == 0 && slot / (NBLK == 4 ? 2 : 1) < 12) {
                        const int mm = slot / (NBLK == 4 ? 2 : 1), plane = mm / 4, pair = mm % 4;
                        if (plane == 0) { rem[2 * pair] = vfn(s + 1, 2 * pair); rem[2 * pair + 1] = vfn(s + 1, 2 * pair + 1); }
                        const __bf16 h0 = (__bf16)rem[2 * pair], h1 = (__bf16)rem[2 * pair + 1];
                        nxt.p[plane][2 * pair] = h0; nxt.p[plane][2 * pair + 1] = h1;
                        if (plane + 1 < NS) { rem[2 * pair] -= (float)h0; rem[2 * pair + 1] -= (float)h1; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (more) {
#pragma unroll
                for (int nb = 0; nb < NBLK; ++nb) { f2[nb] = n2[nb]; f0[nb] = n0[nb]; }
            }
            cur = nxt;
        }
        st.cur = ncur;
        st.k += 1;
    }
}

template <int NS, bool ALPHA_ONLY, int SCHED>     // SCHED 0: two waves per SIMD, lean registers; 1: one wave per SIMD, hand-interleaved
__global__ __launch_bounds__(256, SCHED == 1 ? 1 : 2) void mlp_fwd_split_kernel(
    const __bf16* __restrict__ wq, const float* __restrict__ packed_f32, int F, const float* __restrict__ ndc, int ndc_stride,
    const float* __restrict__ feat, int feat_stride, const float* __restrict__ dirs, int dirs_stride,
    int64_t P, int S, float* __restrict__ raw)
{
    constexpr int SB4 = sp_slab_elems(NS, 4) * 2, SB2 = sp_slab_elems(NS, 2) * 2;      // slab bytes for 4 / 2 output blocks
    extern __shared__ __attribute__((aligned(16))) char lds_s[];
    constexpr int NBUF = SCHED == 1 ? IL_NBUF : split_nbuf(NS);
    float* vec = reinterpret_cast<float*>(lds_s + NBUF * SB4);
    const Layout LF = layout(F);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int half = lane >> 5;
    const int64_t p_raw = ((int64_t)blockIdx.x * 4 + wave) * 32 + (lane & 31);
    const bool live = p_raw < P;
    const int64_t p = live ? p_raw : P - 1;

    // stream order (sizes are static: the pts_bias segment is always packed as 2 slabs): 30 slabs of SB4, then 5 of SB2
    SlabStream st{reinterpret_cast<const char*>(wq), lds_s, SB4, 1, NBUF, 0, ALPHA_ONLY ? 26 : 30, ALPHA_ONLY ? 0 : 5};
#pragma unroll
    for (int i = 0; i + 1 < NBUF; ++i) {          // slabs 0 .. NBUF-2 -> buffers 0 .. NBUF-2 (prefetch targets cur + NBUF - 1)
        st.cur = (i + 1) % NBUF;
        st.prefetch(SB4, wave, lane);
    }
    st.cur = 0;
    for (int i = tid; i < V_TOTAL; i += 256) vec[i] = packed_f32[LF.vec + i];
    const float px = ndc[p * ndc_stride + 0], py = ndc[p * ndc_stride + 1], pz = ndc[p * ndc_stride + 2];
    float fv[32];
    {
        const float* fp = feat + p * feat_stride + half * (F / 2);
#pragma unroll
        for (int i = 0; i < 32; ++i) fv[i] = i < F / 2 ? fp[i] : 0.0f;
    }
    if (SCHED == 1) slab_wait<0>();               // interleaved schedule: a segment starts with its first slab (and `vec`) visible
    auto pe_b = [&](int s) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = pe_op(8 * s + j, half, px, py, pz);
        return split8<NS>(t8);
    };
    float bias[64];
    // the activations stay fp32 in registers and are split into bf16 pieces per k-step inside the GEMM loop (each value is
    // split once per layer either way; holding NS planes instead would cost 32*NS registers and spill at two waves per SIMD)
    float h[64];
    auto hb = [&](int s) { return split8<NS>(h + 8 * (s & 7)); };
    auto hv = [&](int s, int j) { return h[8 * (s & 7) + j]; };
    auto pe_v = [&](int s, int j) { return pe_op(8 * s + j, half, px, py, pz); };
    auto finish = [&](f32x16 (&acc)[4], bool relu, bool mod) {
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            float v = acc[q >> 4][q & 15];
            if (mod) v *= bias[q];
            h[q] = relu ? fmaxf(v, 0.0f) : v;
        }
    };
    {   // bias = pts_bias(feat)
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_BIASG + half * 64);
        if constexpr (SCHED == 1) run_segment_il<NS, 2, 4, SB4, SB4>(st, acc, wave, lane, [&](int s, int j) { return fv[8 * (s & 3) + j]; });
        else run_segment<NS, 2, 4, SB4, SB4, 3>(st, acc, wave, lane, [&](int s) { return split8<NS>(fv + 8 * (s & 3)); });
#pragma unroll
        for (int q = 0; q < 64; ++q) bias[q] = acc[q >> 4][q & 15];
    }
    {   // layer 0
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + half * 64);
        if constexpr (SCHED == 1) run_segment_il<NS, 2, 4, SB4, SB4>(st, acc, wave, lane, pe_v);
        else run_segment<NS, 2, 4, SB4, SB4, 10>(st, acc, wave, lane, pe_b);
        finish(acc, true, true);
    }
#pragma unroll 1
    for (int layer = 1; layer <= 4; ++layer) {
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + 128 * layer + half * 64);
        if constexpr (SCHED == 1) run_segment_il<NS, 4, 4, SB4, SB4>(st, acc, wave, lane, hv);
        else run_segment<NS, 4, 4, SB4, SB4, 3>(st, acc, wave, lane, hb);
        finish(acc, true, true);
    }
    float sigma;
    {   // layer 5 on cat([pts, h4]); the positional encoding is recomputed instead of held in 16*NS registers since layer 0
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_L0 + 128 * 5 + half * 64);
        if constexpr (SCHED == 1) run_segment_il<NS, 2, 4, SB4, SB4>(st, acc, wave, lane, pe_v);
        else run_segment<NS, 2, 4, SB4, SB4, 10>(st, acc, wave, lane, pe_b);
        if constexpr (SCHED == 1) run_segment_il<NS, 4, 4, ALPHA_ONLY ? 0 : SB4, ALPHA_ONLY ? 0 : SB4>(st, acc, wave, lane, hv);
        else run_segment<NS, 4, 4, ALPHA_ONLY ? 0 : SB4, ALPHA_ONLY ? 0 : SB4, 3>(st, acc, wave, lane, hb);
        const float* wa = vec + V_WA + half * 64;
        float part = 0.0f;
#pragma unroll
        for (int q = 0; q < 64; ++q) part = fmaf(wa[q], fmaxf(acc[q >> 4][q & 15] * bias[q], 0.0f), part);
        part += __shfl_xor(part, 32);
        sigma = fmaxf(part + vec[V_BA], 0.0f);
        if (!ALPHA_ONLY) finish(acc, true, true);
    }
    if (ALPHA_ONLY) {
        if (live && half == 0) raw[p_raw] = sigma;
        return;
    }
    {   // feature_linear (no activation)
        f32x16 acc[4];
        init_acc_b<4>(acc, vec + V_FEAT + half * 64);
        if constexpr (SCHED == 1) run_segment_il<NS, 4, 4, SB2, SB2>(st, acc, wave, lane, hv);
        else run_segment<NS, 4, 4, SB2, SB2, 3>(st, acc, wave, lane, hb);
        finish(acc, false, false);
    }
    {   // views_linears[0] + rgb head
        const int64_t ray = ((uint64_t)p >> 32) == 0 ? (int64_t)((unsigned)p / (unsigned)S) : p / S;      // (a 64-bit division is ~10x a 32-bit one)
        float dl[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        dl[0] = half ? dirs[ray * dirs_stride + 1] : dirs[ray * dirs_stride + 0];
        dl[1] = half ? 0.0f : dirs[ray * dirs_stride + 2];
        const Pieces<NS> d8 = split8<NS>(dl);
        const float zero8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const Pieces<NS> z8 = split8<NS>(zero8);
        f32x16 acc[2];
        init_acc_b<2>(acc, vec + V_VIEWS + half * 32);
        if constexpr (SCHED == 1) run_segment_il<NS, 5, 2, 0, 0>(st, acc, wave, lane, [&](int s, int j) { return s < 8 ? h[8 * (s & 7) + j] : (s == 8 ? dl[j] : 0.0f); });
        else run_segment<NS, 5, 2, 0, 0, 6>(st, acc, wave, lane, [&](int s) { return s < 8 ? hb(s) : (s == 8 ? d8 : z8); });
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

template <int NS, int SCHED>
int launch_split(const __bf16* wq, const float* packed_f32, int F, const float* ndc, int ndc_stride, const float* feat, int feat_stride,
                 const float* dirs, int dirs_stride, int64_t P, int S, int alpha_only, float* raw, hipStream_t st)
{
    constexpr int LDS = (SCHED == 1 ? IL_NBUF : split_nbuf(NS)) * sp_slab_elems(NS, 4) * 2 + V_TOTAL * 4;
    static unsigned long long cap_a = 0, cap_b = 0;         // per-device bit masks (common.h)
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_split_kernel<NS, false, SCHED>), (int)(LDS), &cap_a)) return rc_;
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_split_kernel<NS, true, SCHED>), (int)(LDS), &cap_b)) return rc_;
    if (alpha_only)
        mlp_fwd_split_kernel<NS, true, SCHED><<<mvs_cdiv(P, 128), 256, LDS, st>>>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw);
    else
        mlp_fwd_split_kernel<NS, false, SCHED><<<mvs_cdiv(P, 128), 256, LDS, st>>>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw);
    return 0;
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
    static unsigned long long cap_a = 0, cap_b = 0;         // per-device bit masks (common.h)
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_bf16_pair_kernel<false>), (int)(B_LDS_BYTES), &cap_a)) return rc_;
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_bf16_pair_kernel<true>), (int)(B_LDS_BYTES), &cap_b)) return rc_;
    const __bf16* wq = reinterpret_cast<const __bf16*>(packed_bf16);
    const unsigned grid = mvs_cdiv(P, 32 * BP_WAVES);
    if (alpha_only)
        mlp_fwd_bf16_pair_kernel<true><<<grid, BP_THREADS, B_LDS_BYTES, st>>>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw);
    else
        mlp_fwd_bf16_pair_kernel<false><<<grid, BP_THREADS, B_LDS_BYTES, st>>>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// bf16 training forward: mvsnerf_mlp_fwd_bf16 + the activation store of mvsnerf_mlp_fwd_train (same slot format, fp32)
extern "C" int mvsnerf_mlp_fwd_bf16_train(const void* packed_bf16, const float* packed_f32, int F, const float* ndc, int ndc_stride,
                                          const float* feat, int feat_stride, const float* dirs, int dirs_stride,
                                          int64_t N, int S, float* raw, float* saved, void* stream)
{
    if (!packed_bf16 || !packed_f32 || !ndc || !feat || !dirs || !raw || !saved || N < 0 || S < 1 || feat_stride < F || ndc_stride < 3 || dirs_stride < 3) return MVSNERF_EINVAL;
    if (F < 2 || F > 32 || (F & 1)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed_bf16) || !mvs_aligned16(raw) || !mvs_aligned16(saved)) return MVSNERF_EALIGN;
    const int64_t P = N * S;
    if (P == 0) return MVSNERF_OK;
    static unsigned long long cap = 0;
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_fwd_bf16_kernel<false, true>), (int)(B_LDS_BYTES), &cap)) return rc_;
    mlp_fwd_bf16_kernel<false, true><<<mvs_cdiv(P, 128), 256, B_LDS_BYTES, (hipStream_t)stream>>>(
        reinterpret_cast<const __bf16*>(packed_bf16), packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, raw, saved);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}


extern "C" size_t mvsnerf_mlp_packed_split_elems(int F, int n_split)
{
    if (F < 2 || F > MAX_F || (F & 1)) return 0;
    if (n_split == MVSNERF_SPLIT_FP16) return mvs_mlp_f16x3_elems(F);
    if (n_split < 1 || n_split > 3) return 0;
    return sp_total_elems(n_split, F);
}

extern "C" int mvsnerf_mlp_pack_split(const float* const w[11], int F, int n_split, void* packed_split, void* stream)
{
    if (!w || !packed_split) return MVSNERF_EINVAL;
    if (F < 2 || F > MAX_F || (F & 1) || ((n_split < 1 || n_split > 3) && n_split != MVSNERF_SPLIT_FP16)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed_split)) return MVSNERF_EALIGN;
    if (n_split == MVSNERF_SPLIT_FP16) return mvs_mlp_f16x3_pack(w, F, packed_split, (hipStream_t)stream);
    PackBArgs a;
    for (int i = 0; i < 11; ++i) { if (!w[i]) return MVSNERF_EINVAL; a.w[i] = w[i]; }
    a.F = F;
    mlp_pack_split_kernel<<<64, 256, 0, (hipStream_t)stream>>>(a, n_split, reinterpret_cast<__bf16*>(packed_split));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_mlp_fwd_split(const void* packed_split, const float* packed_f32, int F, int n_split, const float* ndc, int ndc_stride,
                                     const float* feat, int feat_stride, const float* dirs, int dirs_stride,
                                     int64_t N, int S, int alpha_only, float* raw, void* stream)
{
    if (!packed_split || !packed_f32 || !ndc || !feat || !raw || N < 0 || S < 1 || feat_stride < F || ndc_stride < 3) return MVSNERF_EINVAL;
    if (!alpha_only && (!dirs || dirs_stride < 3)) return MVSNERF_EINVAL;
    if (F < 2 || F > MAX_F || (F & 1) || ((n_split < 1 || n_split > 3) && n_split != MVSNERF_SPLIT_FP16)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(packed_split) || !mvs_aligned16(raw)) return MVSNERF_EALIGN;
    const int64_t P = N * S;
    if (P == 0) return MVSNERF_OK;
    if (n_split == MVSNERF_SPLIT_FP16)
        return mvs_mlp_f16x3_fwd(packed_split, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, alpha_only, raw, (hipStream_t)stream);
    const __bf16* wq = reinterpret_cast<const __bf16*>(packed_split);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    switch (n_split) {
        case 1: rc = launch_split<1, 0>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, alpha_only, raw, st); break;
        case 2: rc = launch_split<2, 0>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, alpha_only, raw, st); break;
        default:
            rc = launch_split<3, 0>(wq, packed_f32, F, ndc, ndc_stride, feat, feat_stride, dirs, dirs_stride, P, S, alpha_only, raw, st);
            break;
    }
    if (rc) return rc;
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
