// Backward of the fused Renderer_ours MLP (models.py:145-222) for gfx950.
//
//   dgrad  one kernel, same structure as the forward: 32 points per wave, every W^T product on
//          v_mfma_f32_32x32x2_f32 with the gradient chained through registers (C/D fragment of one layer = B operand
//          of the next, see mlp_layout.h).  It reads the activations the training forward saved in slot format and
//          writes the per-layer pre-activation gradients in the same format.
//   wgrad  dW[n][k] = sum_points g[n][p] x[k][p]: a reduction over points, so points become the MFMA contraction
//          index.  A row of a saved tensor (one feature, 32 points, 128 contiguous bytes) is exactly the operand row
//          the MFMA wants: lane (i, half) loads points [16*half, 16*half+16) of row i with four 16-byte loads - no LDS,
//          no transposes.  Work-groups own contiguous tile ranges; partials are combined by a deterministic second
//          kernel (no float atomics), which also un-permutes fragment order back to nn.Linear's [out][in].
#include "common.h"
#include "lds_dma.h"
#include "act.h"
#include "mlp_layout.h"

using namespace mlp;

// ------------------------------------------------------------------------------------------ pack (W^T fragments)
struct PackBwdArgs { const float* w[11]; int F; };

__device__ inline void pack_t_segment(float* __restrict__ dst, const float* __restrict__ W, int ld, int col_off, int n_cols,
                                      int steps, int nb, int tid, int nthreads)
{
    const int total = steps * nb * 64;
    for (int i = tid; i < total; i += nthreads) {
        const int j = i & 3, lane = (i >> 2) & 63, rest = i >> 8;
        const int kb = rest % nb, t = (rest / nb) * 4 + j;
        const int n = act_n(t, lane >> 5);                  // contraction index (output feature of the forward layer)
        const int k = kb * 32 + (lane & 31);                // produced index (input feature)
        dst[i] = k < n_cols ? W[(size_t)n * ld + col_off + k] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void mlp_pack_bwd_kernel(PackBwdArgs a, float* __restrict__ packed)
{
    const LayoutBwd L = layout_bwd();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    pack_t_segment(packed + L.views, a.w[9], WIDTH + 3, 0, WIDTH, 32, 4, tid, nt);          // views_linears.0[:, :128]^T
    pack_t_segment(packed + L.feat, a.w[7], WIDTH, 0, WIDTH, ACT_STEPS, 4, tid, nt);
    pack_t_segment(packed + L.l5, a.w[5], WIDTH + PE_DIM, PE_DIM, WIDTH, ACT_STEPS, 4, tid, nt);   // h-part of cat([pts,h])
    pack_t_segment(packed + L.l4, a.w[4], WIDTH, 0, WIDTH, ACT_STEPS, 4, tid, nt);
    pack_t_segment(packed + L.l3, a.w[3], WIDTH, 0, WIDTH, ACT_STEPS, 4, tid, nt);
    pack_t_segment(packed + L.l2, a.w[2], WIDTH, 0, WIDTH, ACT_STEPS, 4, tid, nt);
    pack_t_segment(packed + L.l1, a.w[1], WIDTH, 0, WIDTH, ACT_STEPS, 4, tid, nt);
    pack_t_segment(packed + L.bias, a.w[6], a.F, 0, a.F, ACT_STEPS, 1, tid, nt);            // pts_bias^T (F <= 32 columns)
}

extern "C" size_t mvsnerf_mlp_packed_bwd_floats(void) { return layout_bwd().total; }

extern "C" int mvsnerf_mlp_pack_bwd(const float* const w[11], int F, float* packed_bwd, void* stream)
{
    if (!w || !packed_bwd) return MVSNERF_EINVAL;
    if (F < 2 || F > 32 || (F & 1)) return MVSNERF_EUNSUPPORTED;
    PackBwdArgs a;
    for (int i = 0; i < 11; ++i) { if (!w[i]) return MVSNERF_EINVAL; a.w[i] = w[i]; }
    a.F = F;
    mlp_pack_bwd_kernel<<<64, 256, 0, (hipStream_t)stream>>>(a, packed_bwd);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---- bf16 variant of the transposed segments (v_mfma_f32_32x32x16_bf16: a k-step spans 16 contraction indices, 8 per lane half;
// element j of lane (i, h) of step s, block kb = W[n(8s + j, h)][col_off + 32 kb + i]); offsets in bf16 elements
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
struct LayoutBwdB { size_t views, feat, l5, l4, l3, l2, l1, bias, total; };
__host__ __device__ inline size_t segb(int steps, int nb) { return (size_t)steps * nb * 64 * 8; }
__host__ __device__ inline LayoutBwdB layout_bwd_b()
{
    LayoutBwdB L;
    size_t o = 0;
    L.views = o; o += segb(4, 4);
    L.feat = o;  o += segb(8, 4);
    L.l5 = o;    o += segb(8, 4);
    L.l4 = o;    o += segb(8, 4);
    L.l3 = o;    o += segb(8, 4);
    L.l2 = o;    o += segb(8, 4);
    L.l1 = o;    o += segb(8, 4);
    L.bias = o;  o += segb(8, 1);
    L.total = o;
    return L;
}

__device__ inline void pack_tb_segment(__bf16* __restrict__ dst, const float* __restrict__ W, int ld, int col_off, int n_cols,
                                       int steps, int nb, int tid, int nthreads)
{
    const int total = steps * nb * 64 * 8;
    for (int i = tid; i < total; i += nthreads) {
        const int j = i & 7, lane = (i >> 3) & 63, rest = i >> 9;
        const int kb = rest % nb, s = rest / nb;
        const int n = act_n(8 * s + j, lane >> 5);
        const int k = kb * 32 + (lane & 31);
        dst[i] = (__bf16)(k < n_cols ? W[(size_t)n * ld + col_off + k] : 0.0f);
    }
}

__global__ __launch_bounds__(256) void mlp_pack_bwd_bf16_kernel(PackBwdArgs a, __bf16* __restrict__ packed)
{
    const LayoutBwdB L = layout_bwd_b();
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    pack_tb_segment(packed + L.views, a.w[9], WIDTH + 3, 0, WIDTH, 4, 4, tid, nt);
    pack_tb_segment(packed + L.feat, a.w[7], WIDTH, 0, WIDTH, 8, 4, tid, nt);
    pack_tb_segment(packed + L.l5, a.w[5], WIDTH + PE_DIM, PE_DIM, WIDTH, 8, 4, tid, nt);
    pack_tb_segment(packed + L.l4, a.w[4], WIDTH, 0, WIDTH, 8, 4, tid, nt);
    pack_tb_segment(packed + L.l3, a.w[3], WIDTH, 0, WIDTH, 8, 4, tid, nt);
    pack_tb_segment(packed + L.l2, a.w[2], WIDTH, 0, WIDTH, 8, 4, tid, nt);
    pack_tb_segment(packed + L.l1, a.w[1], WIDTH, 0, WIDTH, 8, 4, tid, nt);
    pack_tb_segment(packed + L.bias, a.w[6], a.F, 0, a.F, 8, 1, tid, nt);
}

extern "C" size_t mvsnerf_mlp_packed_bwd_bf16_elems(void) { return layout_bwd_b().total; }

extern "C" int mvsnerf_mlp_pack_bwd_bf16(const float* const w[11], int F, void* packed_bwd_bf16, void* stream)
{
    if (!w || !packed_bwd_bf16) return MVSNERF_EINVAL;
    if (F < 2 || F > 32 || (F & 1)) return MVSNERF_EUNSUPPORTED;
    PackBwdArgs a;
    for (int i = 0; i < 11; ++i) { if (!w[i]) return MVSNERF_EINVAL; a.w[i] = w[i]; }
    a.F = F;
    mlp_pack_bwd_bf16_kernel<<<64, 256, 0, (hipStream_t)stream>>>(a, reinterpret_cast<__bf16*>(packed_bwd_bf16));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" size_t mvsnerf_mlp_saved_floats(int64_t n_points) { return (size_t)((n_points + 127) / 128) * 4 * SLOTS_SAVED * 64; }
extern "C" size_t mvsnerf_mlp_gradslot_floats(int64_t n_points) { return (size_t)((n_points + 127) / 128) * 4 * SLOTS_GRAD * 64; }

// ------------------------------------------------------------------------------------------ dgrad
// One workgroup per CU (4 waves, one 32-point tile each).  The eight transposed weight segments stream through two 64 KB LDS
// buffers by LDS-DMA: segment g+1 is in flight while GEMM g runs (the first version staged each segment through registers
// between two barriers and stalled the matrix pipe for every one of them).  The saved post-ReLU activations a layer's
// epilogue needs are requested before the previous GEMM, and the gradient slots are stored after the buffer barrier, so that
// neither global loads nor stores sit between a barrier and the MFMAs that follow it.
constexpr int WBUF_FLOATS = 16384;
constexpr int LDS_FLOATS = 2 * WBUF_FLOATS + V_TOTAL;

__device__ __forceinline__ void wdma(float* __restrict__ dst, const float* __restrict__ src, int n_floats, int wave, int lane)
{
    lds_dma<4>(dst, src, n_floats >> 8, wave, lane);     // 1 KB per wave-instruction (lds_dma.h)
}

__device__ __forceinline__ void wsync()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's DMA pieces have landed
    __syncthreads();                                      // ... everybody's have, and everybody left the other buffer
}

template <int STEPS4, int NBLK, typename BFN>
__device__ __forceinline__ void gemm_t(const float* __restrict__ w, f32x16 (&acc)[NBLK], int lane, BFN bfn)
{
#pragma unroll
    for (int t4 = 0; t4 < STEPS4; ++t4) {
        f32x4 a[NBLK];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) a[b] = *reinterpret_cast<const f32x4*>(w + ((t4 * NBLK + b) * 64 + lane) * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float bv = bfn(t4 * 4 + j);
#pragma unroll
            for (int b = 0; b < NBLK; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[b][j], bv, acc[b], 0, 0, 0);
        }
    }
}

// bf16 form: STEPS k-steps of 16 (8 per lane half); B operand = this lane's gradient registers 8s..8s+7 rounded to bf16
template <int STEPS, int NBLK>
__device__ __forceinline__ void gemm_tb(const float* __restrict__ w, f32x16 (&acc)[NBLK], int lane, const float* __restrict__ g)
{
    const char* wb = reinterpret_cast<const char*>(w);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        bf16x8 bv;
#pragma unroll
        for (int j = 0; j < 8; ++j) bv[j] = (__bf16)g[8 * s + j];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(wb + ((s * NBLK + b) * 64 + lane) * 16);
            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bv, acc[b], 0, 0, 0);
        }
    }
}

template <int NBLK>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[NBLK])
{
#pragma unroll
    for (int b = 0; b < NBLK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
}

template <bool BF> struct SlotType { using type = float; };
template <> struct SlotType<true> { using type = __bf16; };

// BF: the transposed weights are bf16 fragments (mvsnerf_mlp_pack_bwd_bf16) and every W^T product runs on
// v_mfma_f32_32x32x16_bf16 with the gradient operand rounded to bf16 (fp32 accumulate); everything else is unchanged fp32.
template <bool BF>
__global__ __launch_bounds__(256, 1) void mlp_dgrad_kernel(
    const float* __restrict__ packed_fwd, int F, const float* __restrict__ packed_bwd,
    const float* __restrict__ raw, const float* __restrict__ d_raw, const float* __restrict__ saved,
    int64_t P, float* __restrict__ gslots, float* __restrict__ d_feat, int n_feat_out)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* buf0 = lds;
    float* buf1 = lds + WBUF_FLOATS;
    float* vec = lds + 2 * WBUF_FLOATS;
    const Layout LF = layout(F);
    // segment offsets and sizes in FLOAT units of the packed buffer (a bf16 segment of n elements occupies n/2 floats)
    struct Seg { size_t views, feat, l5, bias; int n_views, n_act, n_bias; } L;
    if (BF) {
        const LayoutBwdB B = layout_bwd_b();
        L = Seg{B.views / 2, B.feat / 2, B.l5 / 2, B.bias / 2, (int)(segb(4, 4) / 2), (int)(segb(8, 4) / 2), (int)(segb(8, 1) / 2)};
    } else {
        const LayoutBwd B = layout_bwd();
        L = Seg{B.views, B.feat, B.l5, B.bias, (int)seg_floats(32, 4), (int)seg_floats(ACT_STEPS, 4), (int)seg_floats(ACT_STEPS, 1)};
    }
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    const int64_t p_raw = tile * 32 + (lane & 31);
    const bool live = p_raw < P;
    // BF: both slot buffers hold bf16 elements (what mvsnerf_mlp_fwd_bf16_train stored; what the bf16 weight-gradient GEMMs read)
    using slot_t = typename SlotType<BF>::type;
    const slot_t* sv = reinterpret_cast<const slot_t*>(saved) + tile * (SLOTS_SAVED * 64) + lane;
    slot_t* gs = reinterpret_cast<slot_t*>(gslots) + tile * (SLOTS_GRAD * 64) + lane;

    wdma(buf0, packed_bwd + L.views, L.n_views, wave, lane);                         // segment 0 -> buf0
    for (int i = tid; i < V_TOTAL; i += 256) vec[i] = packed_fwd[LF.vec + i];

    // heads: rgb = sigmoid(z), sigma = relu(s)   (models.py:209,217)
    f32x4 o = {0, 0, 0, 0}, g = {0, 0, 0, 0};
    if (live) { o = *reinterpret_cast<const f32x4*>(raw + p_raw * 4); g = *reinterpret_cast<const f32x4*>(d_raw + p_raw * 4); }
    const float gz0 = g[0] * o[0] * (1.0f - o[0]), gz1 = g[1] * o[1] * (1.0f - o[1]), gz2 = g[2] * o[2] * (1.0f - o[2]);
    const float gsg = o[3] > 0.0f ? g[3] : 0.0f;
    float hv[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) hv[q] = (float)sv[(S_HV + q) * 64];
    wsync();                                                                           // vec + segment 0 visible
    wdma(buf1, packed_bwd + L.feat, L.n_act, wave, lane);                            // segment 1 -> buf1
    gs[(G_G4 + 0) * 64] = (slot_t)(half ? gz1 : gz0);
    gs[(G_G4 + 1) * 64] = (slot_t)(half ? gsg : gz2);

    // grad wrt views_linears[0] pre-activation: ghv = Wr^T gz, masked by relu
    float gh[64];
    {
        const float* wr = vec + V_WR + half * 32;
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const float ghv = fmaf(wr[128 + q], gz2, fmaf(wr[64 + q], gz1, wr[q] * gz0));
            gh[q] = hv[q] > 0.0f ? ghv : 0.0f;
            gs[(G_GPV + q) * 64] = (slot_t)gh[q];
        }
    }
    // gF = views_linears[0][:, :128]^T gpv   (no gradient to the view direction input)
    {
        f32x16 acc[4];
        zero_acc<4>(acc);
        if (BF) gemm_tb<4, 4>(buf0, acc, lane, gh);
        else gemm_t<8, 4>(buf0, acc, lane, [&](int t) { return gh[t]; });
#pragma unroll
        for (int q = 0; q < 64; ++q) gh[q] = acc[q >> 4][q & 15];
    }
    // gh5 = feature_linear^T gF + alpha_linear^T gsigma
    float hq[64];                          // saved post-ReLU activations of the layer whose epilogue comes next
    {
        f32x16 acc[4];
        wsync();                                                                       // segment 1 landed; buf0 free
        wdma(buf0, packed_bwd + L.l5, L.n_act, wave, lane);                          // segment 2 -> buf0
#pragma unroll
        for (int q = 0; q < 64; ++q) gs[(G_GF + q) * 64] = (slot_t)gh[q];
#pragma unroll
        for (int q = 0; q < 64; ++q) hq[q] = (float)sv[(S_H + 5 * 64 + q) * 64];
        zero_acc<4>(acc);
        if (BF) gemm_tb<8, 4>(buf1, acc, lane, gh);
        else gemm_t<16, 4>(buf1, acc, lane, [&](int t) { return gh[t]; });
        const float* wa = vec + V_WA + half * 64;
#pragma unroll
        for (int q = 0; q < 64; ++q) gh[q] = fmaf(wa[q], gsg, acc[q >> 4][q & 15]);
    }
    // pts_linears 5..0:  h_i = relu(p_i * b)  =>  gq = gh*[h_i>0], gp_i = gq*b, gb += gq*p_i = gq*h_i/b
    float bm[64], gbm[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) { bm[q] = (float)sv[(S_BM + q) * 64]; gbm[q] = 0.0f; }
#pragma unroll 1
    for (int layer = 5; layer >= 0; --layer) {
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            const bool on = hq[q] > 0.0f;
            const float gq = on ? gh[q] : 0.0f;
            gbm[q] += on ? gq * (hq[q] / bm[q]) : 0.0f;
            gh[q] = gq * bm[q];
        }
        // segment (7 - layer) for this layer's transposed GEMM sits in buf[(layer + 1) & 1]: l5 -> buf0, l4 -> buf1, ...
        float* cur = (layer & 1) ? buf0 : buf1;
        float* nxt = (layer & 1) ? buf1 : buf0;
        wsync();                                                                       // this layer's segment landed; the other buffer is free
        if (layer >= 2) wdma(nxt, packed_bwd + L.l5 + (size_t)(6 - layer) * L.n_act, L.n_act, wave, lane);
        else if (layer == 1) wdma(nxt, packed_bwd + L.bias, L.n_bias, wave, lane);
#pragma unroll
        for (int q = 0; q < 64; ++q) gs[(G_GP + layer * 64 + q) * 64] = (slot_t)gh[q];
        if (layer == 0) break;
#pragma unroll
        for (int q = 0; q < 64; ++q) hq[q] = (float)sv[(S_H + (layer - 1) * 64 + q) * 64];
        f32x16 acc[4];
        zero_acc<4>(acc);
        if (BF) gemm_tb<8, 4>(cur, acc, lane, gh);
        else gemm_t<16, 4>(cur, acc, lane, [&](int t) { return gh[t]; });
#pragma unroll
        for (int q = 0; q < 64; ++q) gh[q] = acc[q >> 4][q & 15];
    }
#pragma unroll
    for (int q = 0; q < 64; ++q) gs[(G_GBM + q) * 64] = (slot_t)gbm[q];
    // grad wrt the first 8 feature columns (the trilinear volume features): gf = pts_bias^T gb ; the bias segment was
    // requested during layer 1 into buf1 and made visible by the barrier of layer 0
    {
        f32x16 acc[1];
        zero_acc<1>(acc);
        if (BF) gemm_tb<8, 1>(buf1, acc, lane, gbm);
        else gemm_t<16, 1>(buf1, acc, lane, [&](int t) { return gbm[t]; });
        // C/D rows (r&3)+8*(r>>2)+4*half: registers 4q..4q+3 are feature columns 8q + 4*half + (0..3); the first n_feat_out
        // columns are stored (8: the trilinear volume features; F: every input feature, for a colour volume that is trained too)
        if (live) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (8 * q + 4 * half < n_feat_out)
                    *reinterpret_cast<f32x4*>(d_feat + p_raw * n_feat_out + 8 * q + 4 * half) =
                        f32x4{acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad
// G[ra][rb] = sum over tiles, points of A[tile][ra][m] * B[tile][rb][m];   rowsum[ra] = sum A[tile][ra][m]
struct WgradArgs {
    const float* A; int64_t a_tile_stride; int a_slot;           // A rows = slots a_slot.. (RA_BLOCKS*16 slots)
    const float* B; int64_t b_tile_stride; int b_slot0, b_nblk0, b_slot1;   // B = [segment0 (b_nblk0 blocks) | segment1]
    int64_t n_tiles;
    float* partial;      // [gridDim.x][RA][RB + 1]  (last column = row sums)
};

// BF: both slot buffers hold bf16 elements and the 32 points of a tile are contracted by two v_mfma_f32_32x32x16_bf16
// (lane (i, kkh) holds points [16 kkh, 16 kkh + 16): MFMA m takes its points 8m..8m+7 - the same assignment for A and B).
// Weight-gradient GEMMs of the SAME shape run as ONE launch (blockIdx.y = job): a GEMM alone is <= 256 workgroups of one wave per SIMD,
// i.e. latency-bound; two workgroups of different layers per CU cover each other's waits (the registers allow two).
constexpr int WG_MULTI = 5;
struct WgradJobs { WgradArgs j[WG_MULTI]; };

template <int RA_BLOCKS, int NBB, bool BF>
__global__ __launch_bounds__(64 * RA_BLOCKS) void mlp_wgrad_kernel(WgradJobs jobs)
{
    const WgradArgs& w = jobs.j[blockIdx.y];
    const int lane = threadIdx.x & 63, ablk = threadIdx.x >> 6;
    const int i = lane & 31, kkh = lane >> 5;
    constexpr int RB = NBB * 32;
    f32x16 acc[NBB];
    zero_acc<NBB>(acc);
    float rsum = 0.0f;
    const int64_t per = (w.n_tiles + gridDim.x - 1) / gridDim.x;
    const int64_t t0 = blockIdx.x * per, t1 = t0 + per < w.n_tiles ? t0 + per : w.n_tiles;
    const int arow = ablk * 32 + i;
    // Operands of tile t + 1 are requested before the MFMAs of tile t (two register sets, the loop walks them alternately).  The grid is
    // <= 256 workgroups of RA_BLOCKS waves - one wave per SIMD - so nothing else covers the ~2 us between a tile's request and its
    // arrival: without the prefetch every one of the 16 tiles of a workgroup paid that latency in full (50 us per 128 x 128 layer at
    // 1024 x 128 samples, 2.7 TB/s).
    if constexpr (BF) {
        // bf16 slots ([tile][slot][64] two-byte elements): a lane's 16 points of a row are 32 contiguous bytes = the two MFMA operands as stored
        const __bf16* A16 = reinterpret_cast<const __bf16*>(w.A);
        const __bf16* B16 = reinterpret_cast<const __bf16*>(w.B);
        auto request = [&](int64_t tile, bf16x8 (&a8)[2], bf16x8 (&b8)[NBB][2]) {
            const __bf16* ap = A16 + tile * w.a_tile_stride + (int64_t)(w.a_slot + (arow >> 1)) * 64 + (arow & 1) * 32 + kkh * 16;
            a8[0] = *reinterpret_cast<const bf16x8*>(ap); a8[1] = *reinterpret_cast<const bf16x8*>(ap + 8);
#pragma unroll
            for (int bb = 0; bb < NBB; ++bb) {
                const int slot = bb < w.b_nblk0 ? w.b_slot0 + bb * 16 : w.b_slot1 + (bb - w.b_nblk0) * 16;
                const __bf16* bp = B16 + tile * w.b_tile_stride + (int64_t)(slot + (i >> 1)) * 64 + (i & 1) * 32 + kkh * 16;
                b8[bb][0] = *reinterpret_cast<const bf16x8*>(bp); b8[bb][1] = *reinterpret_cast<const bf16x8*>(bp + 8);
            }
        };
        auto contract = [&](const bf16x8 (&a8)[2], const bf16x8 (&b8)[NBB][2]) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
#pragma unroll
                for (int j = 0; j < 8; ++j) rsum += (float)a8[m][j];                    // bias gradient: fp32 row sums
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb) acc[bb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[m], b8[bb][m], acc[bb], 0, 0, 0);
            }
        };
        bf16x8 a_e[2], b_e[NBB][2], a_o[2], b_o[NBB][2];                 // even / odd tiles of this workgroup's range
        if (t0 < t1) request(t0, a_e, b_e);
        for (int64_t tile = t0; tile < t1; tile += 2) {
            if (tile + 1 < t1) request(tile + 1, a_o, b_o);
            contract(a_e, b_e);
            if (tile + 2 < t1) request(tile + 2, a_e, b_e);
            if (tile + 1 < t1) contract(a_o, b_o);
        }
    } else {
        auto request = [&](int64_t tile, f32x4 (&a4)[4], f32x4 (&b4)[NBB][4]) {
            const float* ap = w.A + tile * w.a_tile_stride + (int64_t)(w.a_slot + (arow >> 1)) * 64 + (arow & 1) * 32 + kkh * 16;
#pragma unroll
            for (int k = 0; k < 4; ++k) a4[k] = *reinterpret_cast<const f32x4*>(ap + k * 4);
#pragma unroll
            for (int bb = 0; bb < NBB; ++bb) {
                const int slot = bb < w.b_nblk0 ? w.b_slot0 + bb * 16 : w.b_slot1 + (bb - w.b_nblk0) * 16;
                const float* bp = w.B + tile * w.b_tile_stride + (int64_t)(slot + (i >> 1)) * 64 + (i & 1) * 32 + kkh * 16;
#pragma unroll
                for (int k = 0; k < 4; ++k) b4[bb][k] = *reinterpret_cast<const f32x4*>(bp + k * 4);
            }
        };
        auto contract = [&](const f32x4 (&a4)[4], const f32x4 (&b4)[NBB][4]) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float av = a4[s >> 2][s & 3];
                rsum += av;
#pragma unroll
                for (int bb = 0; bb < NBB; ++bb) acc[bb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b4[bb][s >> 2][s & 3], acc[bb], 0, 0, 0);
            }
        };
        f32x4 a_e[4], b_e[NBB][4], a_o[4], b_o[NBB][4];                  // even / odd tiles of this workgroup's range
        if (t0 < t1) request(t0, a_e, b_e);
        for (int64_t tile = t0; tile < t1; tile += 2) {
            if (tile + 1 < t1) request(tile + 1, a_o, b_o);
            contract(a_e, b_e);
            if (tile + 2 < t1) request(tile + 2, a_e, b_e);
            if (tile + 1 < t1) contract(a_o, b_o);
        }
    }
    constexpr int RA = RA_BLOCKS * 32;
    float* out = w.partial + (int64_t)blockIdx.x * RA * (RB + 1);
#pragma unroll
    for (int bb = 0; bb < NBB; ++bb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ablk * 32 + (r & 3) + 8 * (r >> 2) + 4 * kkh;
            out[(int64_t)row * (RB + 1) + bb * 32 + i] = acc[bb][r];
        }
    rsum += __shfl_xor(rsum, 32);
    if (kkh == 0) out[(int64_t)arow * (RB + 1) + RB] = rsum;
}

// scatter (ra, rb) of the summed partials -> nn.Linear layout via the maps (-1 = drop), for all eleven weight/bias pairs in one launch
// (jobs in the kernel arguments); red[j]: the summed partials of GEMM j
constexpr int WG_JOBS = 12;
struct ScatterJobs {
    const float* red[WG_JOBS];
    const int* rowmap[WG_JOBS];
    const int* colmap[WG_JOBS];
    float* gw[WG_JOBS];
    float* gb[WG_JOBS];
    int RA[WG_JOBS], RB[WG_JOBS], ld[WG_JOBS], blk[WG_JOBS + 1];
    int n;
};

__global__ __launch_bounds__(256) void mlp_wgrad_scatter_multi_kernel(ScatterJobs J)
{
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.blk[j + 1]) ++j;
    const int idx = (blockIdx.x - J.blk[j]) * 256 + threadIdx.x;
    const int RA = J.RA[j], RB = J.RB[j];
    if (idx >= RA * (RB + 1)) return;
    const int ra = idx / (RB + 1), rb = idx - ra * (RB + 1);
    const int n = J.rowmap[j][ra];
    if (n < 0) return;
    const bool is_bias = rb == RB;
    const int k = is_bias ? 0 : J.colmap[j][rb];
    if (k < 0 || (is_bias && !J.gb[j])) return;
    const float v = J.red[j][idx];
    if (is_bias) J.gb[j][n] = v; else J.gw[j][(int64_t)n * J.ld[j] + k] = v;
}

extern "C" int mvsnerf_partial_sum_multi(int n_jobs, const float* const* partial, const int* n_part, const int64_t* n_out, float* const* dst,
                                         float* scratch, void* stream);

// ------------------------------------------------------------------------------------------ host orchestration
static int launch_wgrad(int ra_blocks, int nbb, const WgradJobs& w, int n_jobs, int grid, hipStream_t st, bool bf)
{
#define MVS_WG(RAB, NB) do { if (bf) mlp_wgrad_kernel<RAB, NB, true><<<dim3(grid, n_jobs), 64 * RAB, 0, st>>>(w); else mlp_wgrad_kernel<RAB, NB, false><<<dim3(grid, n_jobs), 64 * RAB, 0, st>>>(w); } while (0)
    switch (ra_blocks * 10 + nbb) {
        case 42: MVS_WG(4, 2); break;
        case 44: MVS_WG(4, 4); break;
        case 46: MVS_WG(4, 6); break;
        case 41: MVS_WG(4, 1); break;
        case 25: MVS_WG(2, 5); break;
        case 16: MVS_WG(1, 6); break;
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_WG
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ten weight-gradient GEMMs, each leaving `grid` <= 256 partial results of RA * (RB + 1) floats, reduced together at the end:
// sum of RA * (RB + 1) = 128*65 + 4*128*129 + 128*193 + 128*33 + 128*129 + 64*161 + 32*193
constexpr size_t WG_NOUT_SUM = 128 * 65 + 4 * 128 * 129 + 128 * 193 + 128 * 33 + 128 * 129 + 64 * 161 + 32 * 193;
extern "C" size_t mvsnerf_mlp_bwd_workspace_floats(void) { return (size_t)(256 + MVS_RED_SLICES + 1) * WG_NOUT_SUM + 1024; }

// maps: device int array of 8 consecutive tables (see mvsnerf_amd/ops.py:_mlp_bwd_maps):
//   [0] act128 (128): slot-row r=2q+h -> n(q,h)            [1] act64 (64): same for q<32
//   [2] pe (64): r=2t+h -> embedding column or -1          [3] pe_l5 (64): same (columns 0..62 of the 191-wide layer 5)
//   [4] feat (32): r=2t+h -> feature column or -1          [5] h_l5 (128): 63 + n(q,h)
//   [6] dir (32): r -> 128 + {0,1,2} or -1                 [7] g4_rgb (32): 0,1,2 -> rgb row, else -1   [8] g4_alpha (32): 3 -> 0
static int mlp_bwd_impl(bool bf, const float* packed_fwd, const float* packed_bwd, int F,
                        const float* raw, const float* d_raw, const float* saved, int64_t N, int S,
                        float* gslots, float* d_feat, int n_feat_out, float* const gw[11], float* const gb[11],
                        const int* maps, float* workspace, void* stream);

extern "C" int mvsnerf_mlp_bwd(const float* packed_fwd, const float* packed_bwd, int F,
                               const float* raw, const float* d_raw, const float* saved, int64_t N, int S,
                               float* gslots, float* d_feat, int n_feat_out, float* const gw[11], float* const gb[11],
                               const int* maps, float* workspace, void* stream)
{
    return mlp_bwd_impl(false, packed_fwd, packed_bwd, F, raw, d_raw, saved, N, S, gslots, d_feat, n_feat_out, gw, gb, maps, workspace, stream);
}

// bf16 backward (AMP-style): every GEMM of the backward pass - W^T products of the data gradient and the point contractions of
// the weight gradients - on v_mfma_f32_32x32x16_bf16 with operands rounded to bf16 and fp32 accumulation; activation-function
// derivatives, the multiplicative bias modulation, the bias gradients and the reductions stay fp32; the gradients come back fp32
// (fp32 master weights, fp32 all-reduce).  packed_bwd_bf16: mvsnerf_mlp_pack_bwd_bf16.  `saved`: the bf16 slots of mvsnerf_mlp_fwd_bf16_train.
extern "C" int mvsnerf_mlp_bwd_bf16(const float* packed_fwd, const void* packed_bwd_bf16, int F,
                                    const float* raw, const float* d_raw, const float* saved, int64_t N, int S,
                                    float* gslots, float* d_feat, int n_feat_out, float* const gw[11], float* const gb[11],
                                    const int* maps, float* workspace, void* stream)
{
    return mlp_bwd_impl(true, packed_fwd, reinterpret_cast<const float*>(packed_bwd_bf16), F, raw, d_raw, saved, N, S, gslots, d_feat, n_feat_out,
                        gw, gb, maps, workspace, stream);
}

static int mlp_bwd_impl(bool bf, const float* packed_fwd, const float* packed_bwd, int F,
                        const float* raw, const float* d_raw, const float* saved, int64_t N, int S,
                        float* gslots, float* d_feat, int n_feat_out, float* const gw[11], float* const gb[11],
                        const int* maps, float* workspace, void* stream)
{
    if (!packed_fwd || !packed_bwd || !raw || !d_raw || !saved || !gslots || !d_feat || !gw || !gb || !maps || !workspace) return MVSNERF_EINVAL;
    if (n_feat_out < 4 || n_feat_out > F || (n_feat_out & 3) || !mvs_aligned16(d_feat)) return MVSNERF_EINVAL;
    if (F < 2 || F > 32 || (F & 1) || N < 0 || S < 1) return MVSNERF_EUNSUPPORTED;
    const int64_t P = N * S;
    if (P == 0) return MVSNERF_OK;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nwg = mvs_cdiv(P, 128);
    const int64_t n_tiles = (int64_t)nwg * 4;
    const size_t lds_bytes = LDS_FLOATS * sizeof(float);
    static unsigned long long lds_cap_set = 0, lds_cap_set_b = 0;          // per-device bit masks (common.h)
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_dgrad_kernel<false>), (int)lds_bytes, &lds_cap_set)) return rc_;
    if (int rc_ = mvs_raise_lds_cap(reinterpret_cast<const void*>(mlp_dgrad_kernel<true>), (int)lds_bytes, &lds_cap_set_b)) return rc_;
    if (bf) mlp_dgrad_kernel<true><<<nwg, 256, lds_bytes, st>>>(packed_fwd, F, packed_bwd, raw, d_raw, saved, P, gslots, d_feat, n_feat_out);
    else mlp_dgrad_kernel<false><<<nwg, 256, lds_bytes, st>>>(packed_fwd, F, packed_bwd, raw, d_raw, saved, P, gslots, d_feat, n_feat_out);
    MVS_LAUNCH_CHECK();

    const int* M_ACT128 = maps, *M_ACT64 = maps + 128, *M_PE = maps + 192, *M_FEAT = maps + 320, *M_HL5 = maps + 352,
             *M_DIR = maps + 480, *M_G4RGB = maps + 512, *M_G4A = maps + 544;
    const int64_t ts_s = (int64_t)SLOTS_SAVED * 64, ts_g = (int64_t)SLOTS_GRAD * 64;
    const int grid = (int)(n_tiles < 256 ? n_tiles : 256);
    int rc;
    // every GEMM leaves its partials [grid][RA*(RB+1)] in its own piece of the workspace; ONE multi-job two-stage sum over the workgroups
    // (fixed order) and ONE scatter launch finish all of them (fragment order -> nn.Linear layout): 5 launches instead of 33
    float* part_next = workspace;
    float* red_all = workspace + (size_t)256 * WG_NOUT_SUM;
    float* red_scratch = red_all + WG_NOUT_SUM;
    float* red_next = red_all;
    const float* ps_part[WG_JOBS]; int ps_np[WG_JOBS]; int64_t ps_no[WG_JOBS]; float* ps_dst[WG_JOBS];
    int n_gemm = 0;
    ScatterJobs SJ;
    SJ.n = 0;
    int sblk = 0;
    // GEMMs are queued by shape (ra_blocks, nbb) and every shape is launched once, its jobs side by side (blockIdx.y)
    constexpr int N_SHAPES = 6;
    int shape_key[N_SHAPES], shape_n[N_SHAPES], n_shapes = 0;
    WgradJobs shape_jobs[N_SHAPES];
    auto gemm = [&](int a_slot, int ra_blocks, int b_slot0, int nblk0, int b_slot1, int nbb, int RA, int RB) -> int {
        WgradArgs w{gslots, ts_g, a_slot, saved, ts_s, b_slot0, nblk0, b_slot1, n_tiles, part_next};
        const int64_t n_out = (int64_t)RA * (RB + 1);
        ps_part[n_gemm] = part_next; ps_np[n_gemm] = grid; ps_no[n_gemm] = n_out; ps_dst[n_gemm] = red_next;
        ++n_gemm;
        part_next += (size_t)grid * n_out;
        red_next += n_out;
        const int key = ra_blocks * 10 + nbb;
        int k = 0;
        while (k < n_shapes && shape_key[k] != key) ++k;
        if (k == n_shapes) {
            if (n_shapes == N_SHAPES) return MVSNERF_EUNSUPPORTED;
            shape_key[k] = key; shape_n[k] = 0; ++n_shapes;
        }
        if (shape_n[k] == WG_MULTI) return MVSNERF_EUNSUPPORTED;
        shape_jobs[k].j[shape_n[k]++] = w;
        return MVSNERF_OK;
    };
    auto scatter = [&](int RA, int RB, const int* rowmap, const int* colmap, float* w_out, int ld, float* b_out) {     // of the LAST gemm's sums
        const int j = SJ.n++;
        SJ.red[j] = ps_dst[n_gemm - 1]; SJ.rowmap[j] = rowmap; SJ.colmap[j] = colmap; SJ.gw[j] = w_out; SJ.gb[j] = b_out;
        SJ.RA[j] = RA; SJ.RB[j] = RB; SJ.ld[j] = ld; SJ.blk[j] = sblk;
        sblk += (RA * (RB + 1) + 255) / 256;
    };
    // pts_linears.0: dW0 = GP0 x E^T
    if ((rc = gemm(G_GP, 4, S_E, 2, 0, 2, 128, 64))) return rc;
    scatter(128, 64, M_ACT128, M_PE, gw[0], PE_DIM, gb[0]);
    // pts_linears.1..4: dWi = GPi x H(i-1)^T
    for (int l = 1; l <= 4; ++l) {
        if ((rc = gemm(G_GP + l * 64, 4, S_H + (l - 1) * 64, 4, 0, 4, 128, 128))) return rc;
        scatter(128, 128, M_ACT128, M_ACT128, gw[l], WIDTH, gb[l]);
    }
    // pts_linears.5 on cat([pts, h4]): B = [E | H4]; the 192-entry column table [pe (64) | 63 + act (128)] follows the 9 tables
    if ((rc = gemm(G_GP + 5 * 64, 4, S_E, 2, S_H + 4 * 64, 6, 128, 192))) return rc;
    scatter(128, 192, M_ACT128, maps + 576, gw[5], WIDTH + PE_DIM, gb[5]);
    // pts_bias: dWb = GBM x Fv^T
    if ((rc = gemm(G_GBM, 4, S_FV, 1, 0, 1, 128, 32))) return rc;
    scatter(128, 32, M_ACT128, M_FEAT, gw[6], F, gb[6]);
    // feature_linear: dWf = GF x H5^T
    if ((rc = gemm(G_GF, 4, S_H + 5 * 64, 4, 0, 4, 128, 128))) return rc;
    scatter(128, 128, M_ACT128, M_ACT128, gw[7], WIDTH, gb[7]);
    // views_linears.0: dWv = GPV x [Fe | dir]^T; 160-entry column table [act128 | 128 + dir]
    if ((rc = gemm(G_GPV, 2, S_FE, 4, S_DR, 5, 64, 160))) return rc;
    scatter(64, 160, M_ACT64, maps + 768, gw[9], WIDTH + 3, gb[9]);
    // heads: rows (gz_r, gz_g, gz_b, gsigma) x [HV (64) | H5 (128)]; two scatters of the same sums
    if ((rc = gemm(G_G4, 1, S_HV, 2, S_H + 5 * 64, 6, 32, 192))) return rc;
    scatter(32, 192, M_G4RGB, maps + 928, gw[10], 64, gb[10]);          // 192 entries: [act64 | -1 x128]
    scatter(32, 192, M_G4A, maps + 1120, gw[8], WIDTH, gb[8]);          // 192 entries: [-1 x64 | act128]
    SJ.blk[SJ.n] = sblk;
    for (int k = 0; k < n_shapes; ++k)
        if ((rc = launch_wgrad(shape_key[k] / 10, shape_key[k] % 10, shape_jobs[k], shape_n[k], grid, st, bf))) return rc;
    if ((rc = mvsnerf_partial_sum_multi(n_gemm, ps_part, ps_np, ps_no, ps_dst, red_scratch, stream))) return rc;
    mlp_wgrad_scatter_multi_kernel<<<sblk, 256, 0, st>>>(SJ);
    MVS_LAUNCH_CHECK();
    (void)M_HL5; (void)M_DIR;
    return MVSNERF_OK;
}
