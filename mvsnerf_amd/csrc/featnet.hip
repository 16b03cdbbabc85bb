// FeatureNet (reference models.py:688-722) building blocks: 2-D convolutions over a batch of N images, channel-last
//   act[n][y][x][C]   C in {4 (rgb + pad), 8, 16, 32}
// with the same lazy InPlaceABN convention as the 3-D U-Net (encoder.hip): a conv writes its RAW output, the
// statistics kernel (mvsnerf_abn_stats over N*H*W pixels) turns batch statistics into (scale, shift), and every
// consumer applies leaky_relu(x*scale+shift) on load.  Kernels here:
//   conv2d_kernel        k in {1,3,5}, stride in {1,2}, padding k/2; also the data gradient of the stride-1 layers
//                        (mirrored taps, swapped channel roles) and the biased 1x1 `toplayer`
//   conv2d_dgrad_s2      data gradient of the k5 s2 p2 layers (a gather-form transposed convolution)
//   conv2d_wgrad         weight gradient, any k / stride
//   channel_sum          bias gradient of `toplayer`
// All fp32 FMA work (the north star reserves MFMA for the MLP).  Weights are re-laid as w[tap][ci][co] and, being
// wave-uniform, are fetched through the scalar cache.
#include "common.h"
#include "act.h"
#include "knobs.h"

__global__ void conv2d_pack_kernel(const float* __restrict__ w, int ci_real, int co_real, int cin_pad, int cout_pad,
                                   int s_ci, int s_co, int ntaps, int flip, float* __restrict__ packed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = ntaps * cin_pad * cout_pad;
    if (i >= total) return;
    const int co = i % cout_pad, ci = (i / cout_pad) % cin_pad;
    int tap = i / (cout_pad * cin_pad);
    if (flip) tap = ntaps - 1 - tap;
    packed[i] = (ci < ci_real && co < co_real) ? w[(int64_t)ci * s_ci + (int64_t)co * s_co + tap] : 0.f;
}

extern "C" int mvsnerf_conv2d_pack_weights(const float* w, int ci_real, int co_real, int cin_pad, int cout_pad,
                                           int s_ci, int s_co, int ksize, int flip, float* packed, void* stream)
{
    if (!w || !packed || ci_real < 1 || co_real < 1 || cin_pad < ci_real || cout_pad < co_real || ksize < 1) return MVSNERF_EINVAL;
    const int ntaps = ksize * ksize;
    conv2d_pack_kernel<<<mvs_cdiv((int64_t)ntaps * cin_pad * cout_pad, 256), 256, 0, (hipStream_t)stream>>>(
        w, ci_real, co_real, cin_pad, cout_pad, s_ci, s_co, ntaps, flip, packed);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// One thread = one output pixel x CT output channels; a workgroup owns a 16x16 pixel tile so that the K*K-fold input
// reuse is served by L1 (a pixel's channels are one contiguous vector, 16 lanes of a row read one contiguous span).
template <int CIN, int CT, int K, int S, int COUT>     // COUT a template constant: weight offsets are s_load immediates (see encoder.hip)
__global__ __launch_bounds__(256) void conv2d_kernel(ActSrc a, int ld, int Hi, int Wi, const float* __restrict__ wp,
                                                     const float* __restrict__ bias,
                                                     float* __restrict__ out, int Ho, int Wo, float* __restrict__ stats = nullptr)
{
    constexpr int P = K / 2;
    const int nbx = (Wo + 15) >> 4, nby = (Ho + 15) >> 4;
    const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);      // halo neighbours share an XCD's L2 (common.h)
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, n = tile_id / (nbx * nby);
    const int cg = blockIdx.y * CT;
    const int x = bx * 16 + (threadIdx.x & 15), y = by * 16 + (threadIdx.x >> 4);
    const bool live = x < Wo && y < Ho;
    float acc[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) acc[k] = bias ? bias[cg + k] : 0.f;
    const ActSrc none{nullptr, nullptr, nullptr};
    const int64_t img0 = (int64_t)n * Hi * Wi;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
        const int yi = y * S - P + ky;
        const bool yin = live && yi >= 0 && yi < Hi;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int xi = x * S - P + kx;
            const bool in = yin && xi >= 0 && xi < Wi;
            const int64_t pix = img0 + (in ? (int64_t)yi * Wi + xi : 0);
            const float* wt = wp + (int64_t)(ky * K + kx) * CIN * COUT + cg;
            // a REAL loop over the channel quads (not unrolled): unrolled, the scheduler requests every weight of the kernel row
            // (K*CIN*CT floats) at the top of the block, which does not fit the SGPR file - the weights were parked in VGPR lanes
            // and came back through two v_readlane per packed FMA (1 456 v_readlane around 816 v_pk_fma_f32 in the 32->32 layer)
#pragma unroll 1
            for (int c = 0; c < CIN; c += 4) {
                f32x4 v;
                load_act4<CIN>(a, none, pix, ld, c, v);
                if (!in) v = f32x4{0, 0, 0, 0};              // zero padding of the *activated* input
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int k = 0; k < CT; ++k) acc[k] = fmaf(v[k4], wt[(c + k4) * COUT + k], acc[k]);
            }
        }
    }
    if (live) {
        float* o = out + (((int64_t)n * Ho + y) * Wo + x) * COUT + cg;
#pragma unroll
        for (int k = 0; k < CT; k += 4) *reinterpret_cast<f32x4*>(o + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
    }
    if constexpr (CT == 8) {
        if (stats) {
            // InPlaceABN partial sums of this 16 x 16 tile's 8 channels (conv0.0 / conv0.1: the full-resolution output is not read again):
            // channel k gets the 32 lanes tid / 32 == k, each adds 8 pixel values and their squares, a shuffle tree finishes; fixed
            // order.  Slot = tile, gridDim.x slots (abn_part_at, common.h).
            __shared__ float red[8 * 256];
#pragma unroll
            for (int k = 0; k < 8; ++k) red[k * 256 + threadIdx.x] = live ? acc[k] : 0.0f;
            __syncthreads();
            const int k = threadIdx.x >> 5, part = threadIdx.x & 31;
            float ssum = 0.f, ssq = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float v = red[k * 256 + part * 8 + i]; ssum += v; ssq = fmaf(v, v, ssq); }
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { ssum += __shfl_xor(ssum, o); ssq += __shfl_xor(ssq, o); }
            if (part == 0) {
                stats[abn_part_at(0, cg + k, COUT, tile_id, gridDim.x)] = ssum;
                stats[abn_part_at(1, cg + k, COUT, tile_id, gridDim.x)] = ssq;
            }
        }
    }
}

// conv0.0 / conv0.1 (3(4) -> 8 and 8 -> 8 channels at FULL resolution, 3x3, stride 1) with the halo staged through LDS.  conv2d_kernel above re-reads every
// tap from L1 and applies the pending InPlaceABN of its input nine times per pixel (a third of its VALU instructions); here the 18 x 18 halo of a 16 x 16
// tile is activated ONCE on its way into LDS - from channel-last memory, or (NCHW3) straight from the caller's (N, 3, H, W) images, which removes the
// nchw_to_nhwc_pad pass of a no-grad encode - and a pixel's nine taps are ds_read_b128s.  Same tile numbering, same products in the same order (ky, kx, channel),
// same InPlaceABN partial sums as conv2d_kernel<CIN, 8, 3, 1, 8>: the results are its bits.
template <int CIN, bool NCHW3>
__global__ __launch_bounds__(256) void conv2d_c8_lds_kernel(ActSrc a, int ld, int Hi, int Wi, const float* __restrict__ wp, float* __restrict__ out,
                                                            float* __restrict__ stats)
{
    static_assert(CIN == 4 || CIN == 8, "4 or 8 input channels");
    constexpr int HW = 18, NH = HW * HW;
    __shared__ __attribute__((aligned(16))) float halo[NH * CIN];
    __shared__ float red[8 * 256];
    const int nbx = (Wi + 15) >> 4, nby = (Hi + 15) >> 4;
    const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);      // halo neighbours share an XCD's L2 (common.h)
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, n = tile_id / (nbx * nby);
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int x = bx * 16 + tx, y = by * 16 + ty;
    const bool live = x < Wi && y < Hi;
    const int64_t img0 = (int64_t)n * Hi * Wi;
    for (int i = threadIdx.x; i < NH; i += 256) {
        const int hy = i / HW, hx = i - hy * HW;
        const int yi = by * 16 - 1 + hy, xi = bx * 16 - 1 + hx;
        const bool in = yi >= 0 && yi < Hi && xi >= 0 && xi < Wi;
        const int64_t pix = in ? (int64_t)yi * Wi + xi : 0;
        if constexpr (NCHW3) {                                           // (N, 3, H, W) images: three coalesced plane reads, channel 3 = the zero pad
            const float* p0 = a.x + (int64_t)n * 3 * Hi * Wi + pix;
            f32x4 v = {p0[0], p0[(int64_t)Hi * Wi], p0[2 * (int64_t)Hi * Wi], 0.0f};
            if (!in) v = f32x4{0, 0, 0, 0};
            *reinterpret_cast<f32x4*>(halo + i * 4) = v;
        } else {
#pragma unroll
            for (int c = 0; c < CIN; c += 4) {
                f32x4 v;
                load_act4<CIN>(a, ActSrc{nullptr, nullptr, nullptr}, img0 + pix, ld, c, v);
                if (!in) v = f32x4{0, 0, 0, 0};                          // zero padding of the *activated* input
                *reinterpret_cast<f32x4*>(halo + i * CIN + c) = v;
            }
        }
    }
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const float* hp = halo + ((ty + ky) * HW + tx + kx) * CIN;
            const float* wt = wp + (int64_t)(ky * 3 + kx) * CIN * 8;
#pragma unroll 1
            for (int c = 0; c < CIN; c += 4) {                          // (a real loop, like conv2d_kernel's: the weights stay on the scalar path)
                const f32x4 v = *reinterpret_cast<const f32x4*>(hp + c);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int k = 0; k < 8; ++k) acc[k] = fmaf(v[k4], wt[(c + k4) * 8 + k], acc[k]);
            }
        }
    }
    if (live) {
        float* o = out + (img0 + (int64_t)y * Wi + x) * 8;
        *reinterpret_cast<f32x4*>(o) = f32x4{acc[0], acc[1], acc[2], acc[3]};
        *reinterpret_cast<f32x4*>(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
    if (stats) {                                                         // as conv2d_kernel: channel k gets the 32 lanes tid / 32 == k, fixed order
#pragma unroll
        for (int k = 0; k < 8; ++k) red[k * 256 + threadIdx.x] = live ? acc[k] : 0.0f;
        __syncthreads();
        const int k = threadIdx.x >> 5, part = threadIdx.x & 31;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float v = red[k * 256 + part * 8 + i]; ssum += v; ssq = fmaf(v, v, ssq); }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { ssum += __shfl_xor(ssum, o); ssq += __shfl_xor(ssq, o); }
        if (part == 0) {
            stats[abn_part_at(0, k, 8, tile_id, gridDim.x)] = ssum;
            stats[abn_part_at(1, k, 8, tile_id, gridDim.x)] = ssq;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The 16- and 32-output-channel layers (conv1.x, conv2.x and the data gradients of the stride-1 ones: 83 % of FeatureNet's multiply-adds)
// on the fp32 matrix cores.  (the VALU kernel above ran FeatureNet at 10.7 % of the fp32 peak.)  Implicit GEMM, one WAVE
// per M-tile of MT consecutive output pixels (linear index over n, y, x - a tile may wrap around a row end), no LDS:
//   COUT = 32: v_mfma_f32_32x32x2_f32, MT = 32, lane (m = lane & 31, kh = lane >> 5)
//   COUT = 16: v_mfma_f32_16x16x4_f32, MT = 16, lane (m = lane & 15, kh = lane >> 4)
// A operand: CPL consecutive input channels of pixel m at the tap (one 8- or 16-byte load, pending InPlaceABN applied on the fly, zero
// padding of the ACTIVATED input); MFMA j contracts the channels c0 + CPL kh + j of all kh.  B operand: w[tap][ci][co] (the VALU kernel's
// layout: a row of COUT consecutive floats per (tap, ci), lanes of one kh read one row).  The inputs stay in L1/L2 across the K x K taps.
// Optionally leaves the InPlaceABN partial sums of its M-tiles (abn_finalize_kernel's layout) so that no statistics pass re-reads the output.
template <int CIN, int COUT, int K, int S>
__global__ __launch_bounds__(256) void conv2d_mfma_kernel(ActSrc a, int ld, int N, int Hi, int Wi, const float* __restrict__ wp,
                                                          float* __restrict__ out, int Ho, int Wo, float* __restrict__ stats)
{
    constexpr int MT = COUT == 32 ? 32 : 16, KL = 64 / MT, P = K / 2;
    constexpr int CPL = CIN / KL >= 4 ? 4 : CIN / KL;               // channels per lane and load
    constexpr int NG = CIN / (KL * CPL);                            // load groups per tap
    static_assert((COUT == 32 || COUT == 16) && CPL >= 2 && NG * KL * CPL == CIN, "channel split");
    typedef float fcpl __attribute__((ext_vector_type(CPL)));
    typedef float facc __attribute__((ext_vector_type(COUT == 32 ? 16 : 4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & (MT - 1), kh = lane / MT;
    const int64_t npix = (int64_t)N * Ho * Wo;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile * MT >= npix) return;
    const int64_t pix = tile * MT + m;
    const bool live = pix < npix;
    const int64_t pc = live ? pix : npix - 1;
    int x, y, n;
    mvs_unflatten3(pc, Wo, Ho, x, y, n);
    const bool lazy = a.scale != nullptr;
    facc acc;
#pragma unroll
    for (int r = 0; r < (COUT == 32 ? 16 : 4); ++r) acc[r] = 0.0f;
    const float* xin = a.x + (int64_t)n * Hi * Wi * ld + kh * CPL;
    float sc[NG][CPL], sh[NG][CPL];                                  // this lane's channels: pending InPlaceABN of the producer
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            sc[g][j] = lazy ? a.scale[g * KL * CPL + kh * CPL + j] : 1.0f;
            sh[g][j] = lazy ? a.shift[g * KL * CPL + kh * CPL + j] : 0.0f;
        }
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int yi = y * S - P + ky;
        const bool yin = live && yi >= 0 && yi < Hi;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const int xi = x * S - P + kx;
            const bool in = yin && xi >= 0 && xi < Wi;
            const float* src = xin + (in ? ((int64_t)yi * Wi + xi) * ld : 0);
            const float* wt = wp + (int64_t)((ky * K + kx) * CIN + kh * CPL) * COUT + m;
            fcpl av[NG];
            float bw[NG][CPL];
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                av[g] = *reinterpret_cast<const fcpl*>(src + g * KL * CPL);
#pragma unroll
                for (int j = 0; j < CPL; ++j) bw[g][j] = wt[(g * KL * CPL + j) * COUT];
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                fcpl v = av[g];
                if (lazy) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) v[j] = act_apply(v[j], sc[g][j], sh[g][j]);
                }
                if (!in) {
#pragma unroll
                    for (int j = 0; j < CPL; ++j) v[j] = 0.0f;
                }
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    if constexpr (COUT == 32) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], bw[g][j], acc, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j], bw[g][j], acc, 0, 0, 0);
                }
            }
        }
    }
    // D: COUT = 32: register r of lane (col m, half kh) = tile pixel (r & 3) + 8 (r >> 2) + 4 kh; COUT = 16: register r = tile pixel 4 kh + r
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int r = 0; r < (COUT == 32 ? 16 : 4); ++r) {
        const int row = COUT == 32 ? (r & 3) + 8 * (r >> 2) + 4 * kh : 4 * kh + r;
        const int64_t op = tile * MT + row;
        if (op < npix) { out[op * COUT + m] = acc[r]; ssum += acc[r]; ssq = fmaf(acc[r], acc[r], ssq); }
    }
    if (stats) {
        ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);
        if (COUT == 16) { ssum += __shfl_xor(ssum, 16); ssq += __shfl_xor(ssq, 16); }
        const int64_t ntiles = (npix + MT - 1) / MT;
        if (kh == 0) { stats[abn_part_at(0, m, COUT, tile, ntiles)] = ssum; stats[abn_part_at(1, m, COUT, tile, ntiles)] = ssq; }
    }
}

static bool conv2d_mfma_shape(int Cin, int Cout, int ksize, int stride)
{
    const int key = ((Cin * 100 + Cout) * 10 + ksize) * 10 + stride;
    return g_conv_mfma && (key == ((8 * 100 + 16) * 10 + 5) * 10 + 2 || key == ((16 * 100 + 16) * 10 + 3) * 10 + 1 ||
                           key == ((16 * 100 + 32) * 10 + 5) * 10 + 2 || key == ((32 * 100 + 32) * 10 + 3) * 10 + 1);
}

// number of M-tiles (= rows of InPlaceABN partial sums) mvsnerf_conv2d_fwd_stats leaves for this layer; 0: the layer has no matrix-core kernel
static bool conv2d_valu_stats_shape(int Cin, int Cout, int ksize, int stride) { return (Cin == 4 || Cin == 8) && Cout == 8 && ksize == 3 && stride == 1; }

extern "C" int mvsnerf_conv2d_mfma_tiles(int Cin, int Cout, int N, int H, int W, int ksize, int stride)
{
    if (N < 1 || H < 1 || W < 1) return 0;
    if (conv2d_valu_stats_shape(Cin, Cout, ksize, stride)) return ((W + 15) / 16) * ((H + 15) / 16) * N;   // conv0.0 / conv0.1: the VALU kernel's 16 x 16 tiles
    if (!conv2d_mfma_shape(Cin, Cout, ksize, stride)) return 0;
    const int P = ksize / 2, Ho = (H + 2 * P - ksize) / stride + 1, Wo = (W + 2 * P - ksize) / stride + 1;
    const int64_t npix = (int64_t)N * Ho * Wo, MT = Cout == 32 ? 32 : 16;
    return (int)((npix + MT - 1) / MT);
}

static int conv2d_mfma_launch(const ActSrc& a, int Cin, int cin_ld, int N, int H, int W, const float* wpacked, int Cout, int ksize, int stride,
                              float* out, float* stats, hipStream_t st)
{
    const int P = ksize / 2, Ho = (H + 2 * P - ksize) / stride + 1, Wo = (W + 2 * P - ksize) / stride + 1;
    const unsigned tiles = (unsigned)mvsnerf_conv2d_mfma_tiles(Cin, Cout, N, H, W, ksize, stride);
    const unsigned nwg = (tiles + 3) / 4;
#define MVS_C2M(CIN, COUT, K, S) conv2d_mfma_kernel<CIN, COUT, K, S><<<nwg, 256, 0, st>>>(a, cin_ld, N, H, W, wpacked, out, Ho, Wo, stats)
    if (Cin == 8) MVS_C2M(8, 16, 5, 2);
    else if (Cin == 16 && Cout == 16) MVS_C2M(16, 16, 3, 1);
    else if (Cin == 16) MVS_C2M(16, 32, 5, 2);
    else MVS_C2M(32, 32, 3, 1);
#undef MVS_C2M
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// mvsnerf_conv2d_fwd of a layer with a matrix-core kernel (mvsnerf_conv2d_mfma_tiles > 0; no bias) that also leaves the InPlaceABN partial
// sums of its raw output: stats_part[2][Cout][tiles] for mvsnerf_abn_finalize
extern "C" int mvsnerf_conv2d_fwd_stats(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int N, int H, int W,
                                        const float* wpacked, int Cout, int ksize, int stride, float* out, float* stats_part, void* stream)
{
    if (!x || !wpacked || !out || !stats_part || ((scale == nullptr) != (shift == nullptr)) || N < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if ((cin_ld & 3) || cin_ld < Cin || !mvs_aligned16(x) || !mvs_aligned16(out)) return MVSNERF_EALIGN;
    if (conv2d_valu_stats_shape(Cin, Cout, ksize, stride)) {
        const ActSrc a{x, scale, shift};
        const unsigned tiles = (unsigned)mvsnerf_conv2d_mfma_tiles(Cin, Cout, N, H, W, ksize, stride);
        if (Cin == 4) conv2d_c8_lds_kernel<4, false><<<tiles, 256, 0, (hipStream_t)stream>>>(a, cin_ld, H, W, wpacked, out, stats_part);
        else conv2d_c8_lds_kernel<8, false><<<tiles, 256, 0, (hipStream_t)stream>>>(a, cin_ld, H, W, wpacked, out, stats_part);
        MVS_LAUNCH_CHECK();
        return MVSNERF_OK;
    }
    if (!conv2d_mfma_shape(Cin, Cout, ksize, stride)) return MVSNERF_EUNSUPPORTED;
    return conv2d_mfma_launch(ActSrc{x, scale, shift}, Cin, cin_ld, N, H, W, wpacked, Cout, ksize, stride, out, stats_part, (hipStream_t)stream);
}

// FeatureNet's first layer (models.py:693: 3 -> 8 channels, 3x3) straight from the caller's (N, 3, H, W) images: the layer stages its halo through LDS anyway,
// so the channel-last, zero-padded copy that mvsnerf_nchw_to_nhwc would make (one more pass over the images per encode) is not needed when nothing else
// reads it (a no-grad encode; the weight gradient of a training step reads the copy).  wpacked: the layer's packed weights with cin_pad = 4; results = the bits
// of mvsnerf_conv2d_fwd_stats on the padded copy.
extern "C" int mvsnerf_conv2d_c3_nchw_fwd_stats(const float* imgs_nchw, int N, int H, int W, const float* wpacked, float* out, float* stats_part, void* stream)
{
    if (!imgs_nchw || !wpacked || !out || !stats_part || N < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(out)) return MVSNERF_EALIGN;
    const unsigned tiles = (unsigned)mvsnerf_conv2d_mfma_tiles(4, 8, N, H, W, 3, 1);
    conv2d_c8_lds_kernel<4, true><<<tiles, 256, 0, (hipStream_t)stream>>>(ActSrc{imgs_nchw, nullptr, nullptr}, 4, H, W, wpacked, out, stats_part);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_conv2d_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld,
                                  int N, int H, int W, const float* wpacked, const float* bias, int Cout,
                                  int ksize, int stride, float* out, void* stream)
{
    if (!x || !wpacked || !out || ((scale == nullptr) != (shift == nullptr)) || N < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if ((cin_ld & 3) || cin_ld < Cin || !mvs_aligned16(x) || !mvs_aligned16(out)) return MVSNERF_EALIGN;
    const int P = ksize / 2;
    const int Ho = (H + 2 * P - ksize) / stride + 1, Wo = (W + 2 * P - ksize) / stride + 1;
    const ActSrc a{x, scale, shift};
    hipStream_t st = (hipStream_t)stream;
    if (!bias && conv2d_mfma_shape(Cin, Cout, ksize, stride))
        return conv2d_mfma_launch(a, Cin, cin_ld, N, H, W, wpacked, Cout, ksize, stride, out, nullptr, st);
    const unsigned tiles = (unsigned)(((Wo + 15) / 16) * ((Ho + 15) / 16) * N);
#define MVS_C2D(CIN, CT, K, S, COUT) conv2d_kernel<CIN, CT, K, S, COUT><<<dim3(tiles, COUT / CT), 256, 0, st>>>(a, cin_ld, H, W, wpacked, bias, out, Ho, Wo)
    const int key = ((Cin * 100 + Cout) * 10 + ksize) * 10 + stride;
    switch (key) {
        case ((4 * 100 + 8) * 10 + 3) * 10 + 1:   MVS_C2D(4, 8, 3, 1, 8); break;      // conv0.0  (rgb + pad -> 8)
        case ((8 * 100 + 8) * 10 + 3) * 10 + 1:   MVS_C2D(8, 8, 3, 1, 8); break;      // conv0.1 and its data gradient
        case ((8 * 100 + 16) * 10 + 5) * 10 + 2:  MVS_C2D(8, 16, 5, 2, 16); break;     // conv1.0
        case ((16 * 100 + 16) * 10 + 3) * 10 + 1: MVS_C2D(16, 16, 3, 1, 16); break;    // conv1.1, conv1.2 (+ data gradients)
        case ((16 * 100 + 32) * 10 + 5) * 10 + 2: MVS_C2D(16, 16, 5, 2, 32); break;    // conv2.0
        case ((32 * 100 + 32) * 10 + 3) * 10 + 1: MVS_C2D(32, 16, 3, 1, 32); break;    // conv2.1, conv2.2 (+ data gradients)
        case ((32 * 100 + 32) * 10 + 1) * 10 + 1: MVS_C2D(32, 16, 1, 1, 32); break;    // toplayer (+ data gradient)
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_C2D
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// Data gradient of a k5 s2 p2 convolution: gx[yi][xi][co] = sum_{ky,kx,ci} g[(yi+2-ky)/2][(xi+2-kx)/2][ci] w[ky][kx][ci][co]
// over the taps whose (yi+2-ky), (xi+2-kx) are even and land inside the (Ho,Wo) grid.  Thread per input-grid pixel.
template <int CIN, int CT>
__global__ __launch_bounds__(256) void conv2d_dgrad_k5s2_kernel(const float* __restrict__ g, int Ho, int Wo, const float* __restrict__ wp,
                                                                int Cout, float* __restrict__ out, int Hi, int Wi)
{
    // blockIdx.z = pixel parity class (y&1, x&1): the contributing taps - hence the weight addresses - are then
    // uniform over the workgroup and stay on the scalar path
    const int py = blockIdx.z >> 1, px = blockIdx.z & 1;
    const int Wh = (Wi + 1) >> 1, Hh = (Hi + 1) >> 1;
    const int nbx = (Wh + 15) >> 4, nby = (Hh + 15) >> 4;
    const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, n = blockIdx.x / (nbx * nby);
    const int cg = blockIdx.y * CT;
    const int x = (bx * 16 + (threadIdx.x & 15)) * 2 + px, y = (by * 16 + (threadIdx.x >> 4)) * 2 + py;
    const bool live = x < Wi && y < Hi;
    float acc[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) acc[k] = 0.f;
    const float* gi = g + (int64_t)n * Ho * Wo * CIN;
    for (int ky = py; ky < 5; ky += 2) {                      // y + 2 - ky even  <=>  ky has y's parity
        const int yo = (y + 2 - ky) >> 1;
        const bool yin = live && yo >= 0 && yo < Ho;
        for (int kx = px; kx < 5; kx += 2) {
            const int xo = (x + 2 - kx) >> 1;
            const bool in = yin && xo >= 0 && xo < Wo;
            const float* gp = gi + (in ? ((int64_t)yo * Wo + xo) * CIN : 0);
            const float* wt = wp + (int64_t)(ky * 5 + kx) * CIN * Cout + cg;
#pragma unroll
            for (int c = 0; c < CIN; c += 4) {
                f32x4 v = *reinterpret_cast<const f32x4*>(gp + c);
                if (!in) v = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int k = 0; k < CT; ++k) acc[k] = fmaf(v[k4], wt[(int64_t)(c + k4) * Cout + k], acc[k]);
            }
        }
    }
    if (live) {
        float* o = out + (((int64_t)n * Hi + y) * Wi + x) * Cout + cg;
#pragma unroll
        for (int k = 0; k < CT; k += 4) *reinterpret_cast<f32x4*>(o + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
    }
}

extern "C" int mvsnerf_conv2d_dgrad_k5s2(const float* g, int Cin, int N, int Ho, int Wo, const float* wpacked, int Cout,
                                         int Hi, int Wi, float* out, void* stream)
{
    if (!g || !wpacked || !out || N < 1 || Ho < 1 || Wo < 1 || Hi < 1 || Wi < 1) return MVSNERF_EINVAL;
    if ((Hi - 1) / 2 + 1 != Ho || (Wi - 1) / 2 + 1 != Wo) return MVSNERF_EINVAL;
    if (!mvs_aligned16(g) || !mvs_aligned16(out)) return MVSNERF_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const unsigned tiles = (unsigned)((((Wi + 1) / 2 + 15) / 16) * (((Hi + 1) / 2 + 15) / 16) * N);
    switch (Cin * 100 + Cout) {
        case 16 * 100 + 8:  conv2d_dgrad_k5s2_kernel<16, 8><<<dim3(tiles, 1, 4), 256, 0, st>>>(g, Ho, Wo, wpacked, Cout, out, Hi, Wi); break;    // conv1.0
        case 32 * 100 + 16: conv2d_dgrad_k5s2_kernel<32, 16><<<dim3(tiles, 1, 4), 256, 0, st>>>(g, Ho, Wo, wpacked, Cout, out, Hi, Wi); break;   // conv2.0
        default: return MVSNERF_EUNSUPPORTED;
    }
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// Weight gradient: gW[a][b][tap] = sum_o G[o][a] * X[o*S - P + tap][b]   (G raw output gradient on the (Ho,Wo) grid,
// X the lazily-activated input).  Same organisation as the 3-D kernel: a thread owns NP (tap,b) pairs x 8 channels of
// `a`, a workgroup walks a contiguous pixel range (G is wave-uniform => scalar loads), partials are reduced afterwards.
template <int NP, int S, int K>
__global__ __launch_bounds__(256) void conv2d_wgrad_kernel(const float* __restrict__ g, int A, ActSrc x1, int B, int ldx,
                                                           int N, int Ho, int Wo, int Hi, int Wi, float* __restrict__ partial)
{
    constexpr int A_T = 8, P = K / 2, U = 4;
    const int a0 = blockIdx.y * A_T;
    const int npix = N * Ho * Wo;
    const int per = (npix + gridDim.x - 1) / gridDim.x;
    const int n0 = blockIdx.x * per, n1 = n0 + per < npix ? n0 + per : npix;
    int dy[NP], dx[NP], bch[NP];
    bool valid[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int p = threadIdx.x + j * blockDim.x;
        valid[j] = p < K * K * B;
        const int tap = valid[j] ? p / B : 0;
        bch[j] = valid[j] ? p - tap * B : 0;
        dy[j] = tap / K - P; dx[j] = tap % K - P;
    }
    float acc[NP][A_T];
#pragma unroll
    for (int j = 0; j < NP; ++j)
#pragma unroll
        for (int a = 0; a < A_T; ++a) acc[j][a] = 0.f;
    int cx = 0, cy = 0, cn = 0;
    if (n0 < n1) { cx = n0 % Wo; cy = (n0 / Wo) % Ho; cn = n0 / (Wo * Ho); }
    for (int ob = n0; ob < n1; ob += U) {
        float gv[U][A_T], xv[U][NP];
#pragma unroll
        for (int u_ = 0; u_ < U; ++u_) {
            const bool live = ob + u_ < n1;
            const int o = live ? ob + u_ : n1 - 1;
            const int ox = cx, oy = cy, on = cn;
            if (live) { if (++cx == Wo) { cx = 0; if (++cy == Ho) { cy = 0; ++cn; } } }
#pragma unroll
            for (int a = 0; a < A_T; ++a) gv[u_][a] = live ? g[(int64_t)o * A + a0 + a] : 0.f;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int yi = oy * S + dy[j], xi = ox * S + dx[j];
                const bool in = live && valid[j] && (unsigned)yi < (unsigned)Hi && (unsigned)xi < (unsigned)Wi;      // (not live: `on` may be N)
                const int64_t idx = in ? (((int64_t)on * Hi + yi) * Wi + xi) * ldx + bch[j] : 0;
                const float t = act1(x1, idx, bch[j]);
                xv[u_][j] = in ? t : 0.f;
            }
        }
#pragma unroll
        for (int u_ = 0; u_ < U; ++u_)
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int a = 0; a < A_T; ++a) acc[j][a] = fmaf(gv[u_][a], xv[u_][j], acc[j][a]);
    }
#pragma unroll
    for (int j = 0; j < NP; ++j)
        if (valid[j]) {
            const int tap = (dy[j] + P) * K + (dx[j] + P);
#pragma unroll
            for (int a = 0; a < A_T; ++a)
                partial[((int64_t)blockIdx.x * A + a0 + a) * B * (K * K) + (int64_t)bch[j] * (K * K) + tap] = acc[j][a];
        }
}

int mvs_conv2d_wgrad_mfma4_parts(int A, int B, int N, int Ho, int Wo, int ksize, int stride, int cap_parts);
int mvs_conv2d_wgrad_mfma4(const float* g, int A, const ActSrc& x1, int B, int ldx, int N, int Ho, int Wo, int Hi, int Wi, int ksize, int stride,
                           float* partial, int cap_parts, hipStream_t st);

// >= 64 pixels per workgroup, <= 4096 workgroups over all channel groups (bounds the partials to 32768*B*k*k floats)
static int wgrad2d_nwg(int64_t npix, int A)
{
    const int64_t cap = 4096 / (A / 8 > 0 ? A / 8 : 1), want = npix / 64;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

extern "C" size_t mvsnerf_conv2d_wgrad_workspace_floats(int A, int B, int ksize)
{
    return (size_t)(4096 / (A / 8 > 0 ? A / 8 : 1) + MVS_RED_SLICES) * A * B * ksize * ksize;
}

extern "C" int mvsnerf_conv2d_wgrad(const float* g, int A, const float* x, const float* x_scale, const float* x_shift, int B, int ldx,
                                    int N, int Ho, int Wo, int Hi, int Wi, int ksize, int stride,
                                    float* gw, float* workspace, void* stream)
{
    if (!g || !x || !workspace || A < 8 || (A & 7) || B < 1 || ldx < B || N < 1) return MVSNERF_EINVAL;
    if (((x_scale == nullptr) != (x_shift == nullptr))) return MVSNERF_EINVAL;
    const ActSrc X1{x, x_scale, x_shift};
    const int64_t npix = (int64_t)N * Ho * Wo;
    if (npix > 0x7fffffff) return MVSNERF_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (g_conv_mfma) {                                     // matrix cores (wgrad_mfma.hip) for FeatureNet's eight k3 / k5 layers
        const int cap = 4096 / (A / 8);
        const int rc = mvs_conv2d_wgrad_mfma4(g, A, X1, B, ldx, N, Ho, Wo, Hi, Wi, ksize, stride, workspace, cap, st);
        if (rc != MVSNERF_EUNSUPPORTED) {
            if (rc != MVSNERF_OK || !gw) return rc;
            const int64_t n_out = (int64_t)A * B * ksize * ksize;
            mvs_partial_sum(workspace, mvs_conv2d_wgrad_mfma4_parts(A, B, N, Ho, Wo, ksize, stride, cap), n_out, workspace + (size_t)cap * n_out, gw, st);
            MVS_LAUNCH_CHECK();
            return MVSNERF_OK;
        }
    }
    const int nwg = wgrad2d_nwg(npix, A);
    const int pairs = ksize * ksize * B;
    const int threads = pairs >= 256 ? 256 : (pairs + 63) / 64 * 64;      // waves without any (tap,b) pair are not launched
    const int np = (pairs + threads - 1) / threads;
    const dim3 grid(nwg, A / 8);
#define MVS_WG2(NP_, S_, K_) conv2d_wgrad_kernel<NP_, S_, K_><<<grid, threads, 0, st>>>(g, A, X1, B, ldx, N, Ho, Wo, Hi, Wi, workspace)
    switch ((np * 10 + stride) * 10 + ksize) {
        case (1 * 10 + 1) * 10 + 3: MVS_WG2(1, 1, 3); break;     // 9B <= 256: B = 3, 8, 16
        case (2 * 10 + 1) * 10 + 3: MVS_WG2(2, 1, 3); break;     // B = 32
        case (1 * 10 + 2) * 10 + 5: MVS_WG2(1, 2, 5); break;     // conv1.0 (25*8 = 200)
        case (2 * 10 + 2) * 10 + 5: MVS_WG2(2, 2, 5); break;     // conv2.0 (25*16 = 400)
        case (1 * 10 + 1) * 10 + 1: MVS_WG2(1, 1, 1); break;     // toplayer
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_WG2
    MVS_LAUNCH_CHECK();
    if (!gw) return MVSNERF_OK;                            // partials left in `workspace` for mvsnerf_partial_sum_multi
    const int64_t n_out = (int64_t)A * B * ksize * ksize;
    mvs_partial_sum(workspace, nwg, n_out, workspace + (size_t)(4096 / (A / 8)) * n_out, gw, st);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// number of partial results mvsnerf_conv2d_wgrad leaves at the start of its workspace (rows of A*B*k*k floats)
extern "C" int mvsnerf_conv2d_wgrad_parts(int A, int B, int N, int Ho, int Wo, int ksize, int stride)
{
    if (A < 8) return 0;
    if (g_conv_mfma) {
        const int n = mvs_conv2d_wgrad_mfma4_parts(A, B, N, Ho, Wo, ksize, stride, 4096 / (A / 8));
        if (n > 0) return n;
    }
    return wgrad2d_nwg((int64_t)N * Ho * Wo, A);
}

// per-channel sum over n rows of a dense [n][C] tensor (C <= 64, 256 % C == 0): the bias gradient of `toplayer`
__global__ __launch_bounds__(256) void channel_sum_partial_kernel(const float* __restrict__ g, int64_t n, int C, float* __restrict__ partial)
{
    __shared__ float red[256];
    const int c = threadIdx.x % C, r = threadIdx.x / C, rows = 256 / C;
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * rows + r; i < n; i += (int64_t)gridDim.x * rows) s += g[i * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < C) {
        float t = 0.f;
        for (int k = 0; k < rows; ++k) t += red[k * C + threadIdx.x];
        partial[(int64_t)blockIdx.x * C + threadIdx.x] = t;
    }
}

extern "C" size_t mvsnerf_channel_sum_workspace_floats(int C) { return (size_t)(256 + MVS_RED_SLICES) * C; }

extern "C" int mvsnerf_channel_sum(const float* g, int64_t n, int C, float* out, float* workspace, void* stream)
{
    if (!g || !out || !workspace || n < 1 || C < 1 || C > 64 || (256 % C)) return MVSNERF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int rows = 256 / C;
    int nb = (int)((n + rows - 1) / rows);
    if (nb > 256) nb = 256;
    channel_sum_partial_kernel<<<nb, 256, 0, st>>>(g, n, C, workspace);
    MVS_LAUNCH_CHECK();
    mvs_partial_sum(workspace, nb, C, workspace + (size_t)256 * C, out, st);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
