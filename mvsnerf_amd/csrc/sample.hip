// Gather kernels of the ray march: trilinear volume lookup, per-view colour lookup, view-direction
// feature, and the NCDHW<->NDHWC boundary transposes.  All HBM/L2-bound; no MFMA.
//
// Layouts (include/mvsnerf_hip.h): MVSNERF_VOL_DHWC vol[d][y][x][8] - one trilinear corner = one 32-byte sector, the two x-neighbours of
// a corner pair one 64-byte read - and MVSNERF_VOL_HWDC vol[y][x][d][8], depth fastest: what the encoder emits, because the samples of a ray
// walk depth (the `_zfast` kernels below; sample_dev.h).  Same arithmetic, same bits.
#include "common.h"
#include "sample_dev.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------
// Trilinear lookup (reference: F.grid_sample 5-D, zeros padding, align_corners=True;
// utils.py:381-382).  4 lanes cooperate on one sample.  Lane q owns float4 number q of the 64-byte
// x-pair row  vol[z][y][x0..x0+1][0..7]  (q>>1 = x corner, q&1 = channel half) and walks the four
// (z,y) rows, so every wave-level load instruction reads 16 samples x 64 CONTIGUOUS bytes (one
// segment per quad instead of four); one __shfl_xor(.,2) step folds the two x corners.
// Measured alternatives (rocprofv3, 1024x128 samples, 128x176x208 volume): this mapping 8.1 us; 2 lanes per
// sample 11.9 us; 2 or 4 consecutive samples per quad 11.6 / 9.0 us; 1024-thread blocks 8.4 us.
// ---------------------------------------------------------------------------------------------
template <int SPQ, bool SMALL>   // samples per lane quad; SMALL: 32-bit voxel offsets (vox_off8)
__global__ __launch_bounds__(256) void volume_sample_c8_kernel(
    const float* __restrict__ vol, int D, int H, int W,
    const float* __restrict__ ndc, int64_t P, float* __restrict__ out, int out_stride)
{
#pragma clang fp contract(off)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(tid & 3);
    const int xc = q >> 1, ch = (q & 1) * 4;
    const int64_t p0 = (tid >> 2) * SPQ;
    f32x4 acc[SPQ];
#pragma unroll
    for (int j = 0; j < SPQ; ++j) {
        const int64_t p = p0 + j;
        const int64_t pc = p < P ? p : (P - 1);
        // same op order as the reference: grid = ndc*2-1 (utils.py:381); unnormalise ((g+1)/2)*(size-1)
        typedef float f32x3 __attribute__((ext_vector_type(3)));
        const f32x3 nd = *reinterpret_cast<const f32x3*>(ndc + pc * 3);            // one 12-byte load (4-byte aligned is enough for dwordx3)
        const float gx = nd[0] * 2.0f - 1.0f;
        const float gy = nd[1] * 2.0f - 1.0f;
        const float gz = nd[2] * 2.0f - 1.0f;
        const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
        const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
        const float iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
        const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
        // weights as ATen forms them: (x1-ix) for the low corner, (ix-x0) for the high one
        const float wx = xc ? (ix - fx) : ((fx + 1.0f) - ix);
        const float cxf = fx + (float)xc;
        // NaN / huge coordinates: the float compares reject them before any int conversion is used
        const bool x_in = (cxf >= 0.0f) && (cxf <= (float)(W - 1));
        f32x4 vv[4];
        float vw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int zc = k >> 1, yc = k & 1;
            const float cyf = fy + (float)yc, czf = fz + (float)zc;
            const bool in = x_in && (cyf >= 0.0f) && (cyf <= (float)(H - 1)) && (czf >= 0.0f) && (czf <= (float)(D - 1));
            vw[k] = (wx * (yc ? (iy - fy) : ((fy + 1.0f) - iy))) * (zc ? (iz - fz) : ((fz + 1.0f) - iz));
            const float* src = in ? vol + vox_off8<SMALL>((int)czf, (int)cyf, (int)cxf, H, W) + ch : reinterpret_cast<const float*>(&g_zero_tap);
            vv[k] = ldg16(src);
        }
        acc[j] = trilinear_fold_x0_lane(vv, vw);                                  // ATen's term order and roundings (sample_dev.h); valid in the x0 lanes
    }
#pragma unroll
    for (int j = 0; j < SPQ; ++j) {
        if (p0 + j < P && xc == 0)                                                // lanes q=0,1 store channels 0-3 / 4-7
            *reinterpret_cast<f32x4*>(out + (p0 + j) * out_stride + ch) = acc[j];
    }
}

// ---------------------------------------------------------------------------------------------
// The same lookup on a depth-fastest volume (MVSNERF_VOL_HWDC, sample_dev.h): four lanes per sample, lane = (row y, channel half).
// ---------------------------------------------------------------------------------------------
template <bool SMALL>
__global__ __launch_bounds__(256) void volume_sample_c8_zfast_kernel(
    const float* __restrict__ vol, int D, int H, int W,
    const float* __restrict__ ndc, int64_t P, float* __restrict__ out, int out_stride)
{
#pragma clang fp contract(off)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(tid & 3);
    const int64_t p = tid >> 2;
    const int64_t pc = p < P ? p : (P - 1);
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    const f32x3 nd = *reinterpret_cast<const f32x3*>(ndc + pc * 3);
    const ZfastTaps t = zfast_taps<SMALL>(vol, D, H, W, nd[0], nd[1], nd[2], q);
    const f32x4 acc = zfast_fold_y0_lane(t);                                      // ATen's term order and roundings; valid in the y0 lanes
    if (p < P && q < 2) *reinterpret_cast<f32x4*>(out + p * out_stride + q * 4) = acc;
}

// Generic-C fallback of the same op (C != 8, e.g. colour volumes): one thread per (sample, channel); either layout.
__global__ __launch_bounds__(256) void volume_sample_generic_kernel(
    const float* __restrict__ vol, int D, int H, int W, int C,
    const float* __restrict__ ndc, int64_t P, float* __restrict__ out, int out_stride, int zfast)
{
#pragma clang fp contract(off)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= P * C) return;
    const int64_t p = tid / C;
    const int c = (int)(tid - p * C);
    const float ix = ((ndc[p * 3 + 0] * 2.0f - 1.0f + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((ndc[p * 3 + 1] * 2.0f - 1.0f + 1.0f) / 2.0f) * (float)(H - 1);
    const float iz = ((ndc[p * 3 + 2] * 2.0f - 1.0f + 1.0f) / 2.0f) * (float)(D - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int zc = k >> 2, yc = (k >> 1) & 1, xc = k & 1;
        const float cx = fx + xc, cy = fy + yc, cz = fz + zc;
        const float w = ((xc ? ix - fx : fx + 1.0f - ix) * (yc ? iy - fy : fy + 1.0f - iy)) * (zc ? iz - fz : fz + 1.0f - iz);
        if (cx >= 0.0f && cx <= (float)(W - 1) && cy >= 0.0f && cy <= (float)(H - 1) && cz >= 0.0f && cz <= (float)(D - 1))
            acc += vol[(zfast ? (((int64_t)cy * W + (int)cx) * D + (int)cz) : (((int64_t)cz * H + (int)cy) * W + (int)cx)) * C + c] * w;
    }
    out[p * out_stride + c] = acc;
}

extern "C" int mvsnerf_volume_sample_fwd(const float* vol, int D, int H, int W, int C,
                                         const float* ndc, int64_t P, float* out, int out_stride, int vol_layout, void* stream)
{
    if (!vol || !ndc || !out || D < 1 || H < 1 || W < 1 || C < 1 || P < 0 || out_stride < C) return MVSNERF_EINVAL;
    if (vol_layout != MVSNERF_VOL_DHWC && vol_layout != MVSNERF_VOL_HWDC) return MVSNERF_EINVAL;
    if (P == 0) return MVSNERF_OK;
    hipStream_t st = (hipStream_t)stream;
    if (C == 8 && !mvs_aligned16(vol)) return MVSNERF_EALIGN;
    if (C == 8 && !(out_stride & 3) && mvs_aligned16(out)) {      // 16-byte row stores; rows of another stride take the per-channel kernel (same bits)
        const bool small = (int64_t)D * H < (1 << 24) && W < (1 << 24) && (int64_t)H * W < (1 << 24) && D < (1 << 24) && (int64_t)D * H * W * 8 < ((int64_t)1 << 31);
        const unsigned grid = mvs_cdiv(P * 4, 256);
        if (vol_layout == MVSNERF_VOL_HWDC) {
            if (small) volume_sample_c8_zfast_kernel<true><<<grid, 256, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride);
            else volume_sample_c8_zfast_kernel<false><<<grid, 256, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride);
        } else {
            if (small) volume_sample_c8_kernel<1, true><<<grid, 256, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride);
            else volume_sample_c8_kernel<1, false><<<grid, 256, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride);
        }
    } else {
        volume_sample_generic_kernel<<<mvs_cdiv(P * C, 256), 256, 0, st>>>(vol, D, H, W, C, ndc, P, out, out_stride, vol_layout == MVSNERF_VOL_HWDC);
    }
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// Shared arithmetic of the colour lookup and the direction feature.  Written with explicit fmaf/mul/add and fp
// contraction OFF so that the stand-alone kernels and the fused gather kernel produce the same bits whatever the
// surrounding code looks like to the optimiser.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Per-view colour lookup (utils.py:300-332).  One thread per (sample, view).  Camera matrices are
// wave-uniform per view only if V divides the wave, so they are read through the vector path from a
// 21-float-per-view table (L1/L2 resident).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void color_sample_kernel(
    const float* __restrict__ imgs, int V, int H, int W,
    const float* __restrict__ w2c, const float* __restrict__ Kmat,
    const float* __restrict__ pts, int64_t P, int with_mask, float* __restrict__ out, int out_stride)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= P * V) return;
    const int64_t p = tid / V;
    const int v = (int)(tid - p * V);
    const ColorTap t = color_project(pts[p * 3 + 0], pts[p * 3 + 1], pts[p * 3 + 2], w2c + v * 16, Kmat + v * 9, W, H);
    const int Cv = 3 + (with_mask ? 1 : 0);
    float* o = out + p * out_stride + v * Cv;
    const float* img = imgs + (int64_t)v * 3 * H * W;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* pl = img + (int64_t)c * H * W + (int64_t)t.y0 * W + t.x0;
        o[c] = color_blend(t, pl[0], t.x1in ? pl[1] : 0.f, t.y1in ? pl[W] : 0.f, (t.x1in && t.y1in) ? pl[W + 1] : 0.f);
    }
    if (with_mask) o[3] = color_mask(t);
}

extern "C" int mvsnerf_color_sample_fwd(const float* imgs, int V, int H, int W, const float* w2c, const float* K,
                                        const float* pts, int64_t P, int with_mask, float* out, int out_stride, void* stream)
{
    if (!imgs || !w2c || !K || !pts || !out || V < 1 || H < 2 || W < 2 || P < 0) return MVSNERF_EINVAL;
    if (out_stride < V * (3 + (with_mask ? 1 : 0))) return MVSNERF_EINVAL;
    if (P == 0) return MVSNERF_OK;
    color_sample_kernel<<<mvs_cdiv(P * V, 256), 256, 0, (hipStream_t)stream>>>(imgs, V, H, W, w2c, K, pts, P, with_mask, out, out_stride);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// build_color_volume(img_feat=...) (utils.py:300-332, the `img_feat is not None` branch): per view [r, g, b | Cf feature channels | mask].
// Colours: border padding (utils.py:320); the per-view feature maps (their own resolution Hf x Wf) are read at the SAME normalised grid
// with ZEROS padding (:322).  One thread per (sample, view); feat maps NCHW [V][Cf][Hf][Wf].  Off the hot path (training_step passes None).
__global__ __launch_bounds__(256) void color_feat_sample_kernel(
    const float* __restrict__ imgs, int V, int H, int W, const float* __restrict__ feats, int Cf, int Hf, int Wf,
    const float* __restrict__ w2c, const float* __restrict__ Kmat,
    const float* __restrict__ pts, int64_t P, int with_mask, float* __restrict__ out, int out_stride)
{
#pragma clang fp contract(off)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= P * V) return;
    const int64_t p = tid / V;
    const int v = (int)(tid - p * V);
    const ColorTap t = color_project(pts[p * 3 + 0], pts[p * 3 + 1], pts[p * 3 + 2], w2c + v * 16, Kmat + v * 9, W, H);
    const int Cv = 3 + Cf + (with_mask ? 1 : 0);
    float* o = out + p * out_stride + v * Cv;
    const float* img = imgs + (int64_t)v * 3 * H * W;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* pl = img + (int64_t)c * H * W + (int64_t)t.y0 * W + t.x0;
        o[c] = color_blend(t, pl[0], t.x1in ? pl[1] : 0.f, t.y1in ? pl[W] : 0.f, (t.x1in && t.y1in) ? pl[W + 1] : 0.f);
    }
    // zeros padding, align_corners=True, the feature map's own size (ATen grid_sampler_unnormalize + the nw,ne,sw,se fma chain)
    const float ix = ((t.gx + 1.0f) / 2.0f) * (float)(Wf - 1), iy = ((t.gy + 1.0f) / 2.0f) * (float)(Hf - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
    const bool x0in = fx >= 0.f && fx <= (float)(Wf - 1), x1in = fx + 1.f >= 0.f && fx + 1.f <= (float)(Wf - 1);
    const bool y0in = fy >= 0.f && fy <= (float)(Hf - 1), y1in = fy + 1.f >= 0.f && fy + 1.f <= (float)(Hf - 1);
    const bool finite = (ix == ix) && (iy == iy) && fabsf(ix) < 1e9f && fabsf(iy) < 1e9f;
    const int x0 = finite ? (int)fx : 0, y0 = finite ? (int)fy : 0;
    const float* fb = feats + (int64_t)v * Cf * Hf * Wf;
    for (int c = 0; c < Cf; ++c) {
        const float* pl = fb + (int64_t)c * Hf * Wf;
        const float nw = (finite && x0in && y0in) ? pl[(int64_t)y0 * Wf + x0] : 0.f, ne = (finite && x1in && y0in) ? pl[(int64_t)y0 * Wf + x0 + 1] : 0.f;
        const float sw = (finite && x0in && y1in) ? pl[(int64_t)(y0 + 1) * Wf + x0] : 0.f, se = (finite && x1in && y1in) ? pl[(int64_t)(y0 + 1) * Wf + x0 + 1] : 0.f;
        o[3 + c] = fmaf(se, wx1 * wy1, fmaf(sw, wx0 * wy1, fmaf(ne, wx1 * wy0, nw * (wx0 * wy0))));
    }
    if (with_mask) o[3 + Cf] = color_mask(t);
}

extern "C" int mvsnerf_color_feat_sample_fwd(const float* imgs, int V, int H, int W, const float* img_feat, int Cf, int Hf, int Wf,
                                             const float* w2c, const float* K, const float* pts, int64_t P, int with_mask,
                                             float* out, int out_stride, void* stream)
{
    if (!imgs || !img_feat || !w2c || !K || !pts || !out || V < 1 || H < 2 || W < 2 || Cf < 1 || Hf < 2 || Wf < 2 || P < 0) return MVSNERF_EINVAL;
    if (out_stride < V * (3 + Cf + (with_mask ? 1 : 0))) return MVSNERF_EINVAL;
    if (P == 0) return MVSNERF_OK;
    color_feat_sample_kernel<<<mvs_cdiv(P * V, 256), 256, 0, (hipStream_t)stream>>>(imgs, V, H, W, img_feat, Cf, Hf, Wf, w2c, K, pts, P, with_mask, out, out_stride);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// dirs = normalise(rays_dir) @ R_ref^T     (renderer.py:142-147, 111-122)
// ---------------------------------------------------------------------------------------------
__global__ void dir_feature_kernel(const float* __restrict__ rays_dir, const float* __restrict__ w2c, int64_t N, int normalize, float* __restrict__ out)
{
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    dir_feature_of(rays_dir + n * 3, w2c, normalize, out + n * 3);
}

extern "C" int mvsnerf_dir_feature_fwd(const float* rays_dir, const float* w2c_ref, int64_t N, int normalize, float* dirs_out, void* stream)
{
    if (!rays_dir || !dirs_out || N < 0) return MVSNERF_EINVAL;
    if (N == 0) return MVSNERF_OK;
    dir_feature_kernel<<<mvs_cdiv(N, 256), 256, 0, (hipStream_t)stream>>>(rays_dir, w2c_ref, N, normalize, dirs_out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// gen_pts_feats + gen_dir_feature in ONE launch (renderer.py:111-136): the three gathers above are each a few
// microseconds of exposed memory latency, so one kernel that has all of a sample's loads in flight at once costs
// about as much as the slowest of them.  Same lane quad per sample as the trilinear kernel: lane q folds its share of
// the 8 volume corners, lanes q < V additionally project the sample into source view q (q, q+4 for V > 4) and fetch
// the 4 bilinear taps as 16-byte pixels from channel-last images img[v][y][x][4], and the quad of a ray's first
// sample writes the ray's view-direction feature.  Arithmetic (operation order included) is that of the separate
// kernels, so results are bit-identical to them.
// ---------------------------------------------------------------------------------------------
struct GatherArgs {
    const float* vol; int D, H, W;
    const float* img; int V, IH, IW;            // [V][IH][IW][4]
    const float* w2c; const float* Kmat;        // [V][4][4], [V][3][3]
    const float* pts; const float* ndc; int64_t P; int64_t N;
    const float* rays_dir;                      // [P/S][3]
    float* feat; int feat_stride; float* dirs_out;
};

template <bool SMALL, bool ZFAST>      // ZFAST: the volume is depth-fastest (MVSNERF_VOL_HWDC): lane = (row y, channel half), see sample_dev.h
__global__ __launch_bounds__(256) void gather_fused_kernel(GatherArgs a)
{
#pragma clang fp contract(off)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(tid & 3);
    const int64_t p_raw = tid >> 2;
    const bool live = p_raw < a.P;
    using idx_t = typename std::conditional<SMALL, unsigned, int64_t>::type;      // SMALL: sample offsets fit 32 bits (checked by the launcher)
    const idx_t p = (idx_t)(live ? p_raw : a.P - 1);
    const int D = a.D, H = a.H, W = a.W;
    // ---- trilinear volume lookup (identical to volume_sample_c8_kernel<1> / volume_sample_c8_zfast_kernel)
    const int xc = q >> 1, ch = (q & 1) * 4;
    f32x4 vv[4];
    float vw[4];
    ZfastTaps zt;
    if constexpr (ZFAST) {
        zt = zfast_taps<SMALL>(a.vol, D, H, W, a.ndc[p * 3 + 0], a.ndc[p * 3 + 1], a.ndc[p * 3 + 2], q);
    } else {
        const float gx = a.ndc[p * 3 + 0] * 2.0f - 1.0f;
        const float gy = a.ndc[p * 3 + 1] * 2.0f - 1.0f;
        const float gz = a.ndc[p * 3 + 2] * 2.0f - 1.0f;
        const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
        const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
        const float iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
        const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
        const float wx = xc ? (ix - fx) : ((fx + 1.0f) - ix);
        const float cxf = fx + (float)xc;
        const bool x_in = (cxf >= 0.0f) && (cxf <= (float)(W - 1));
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int zc = k >> 1, yc = k & 1;
            const float cyf = fy + (float)yc, czf = fz + (float)zc;
            const bool in = x_in && (cyf >= 0.0f) && (cyf <= (float)(H - 1)) && (czf >= 0.0f) && (czf <= (float)(D - 1));
            vw[k] = (wx * (yc ? (iy - fy) : ((fy + 1.0f) - iy))) * (zc ? (iz - fz) : ((fz + 1.0f) - iz));
            const float* src = in ? a.vol + vox_off8<SMALL>((int)czf, (int)cyf, (int)cxf, H, W) + ch : reinterpret_cast<const float*>(&g_zero_tap);
            vv[k] = ldg16(src);
        }
    }
    // ---- colour lookup of view q (+4): issue its taps before the volume taps are consumed
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    const f32x3 pp = *reinterpret_cast<const f32x3*>(a.pts + p * 3);              // one 12-byte load
    const float px = pp[0], py = pp[1], pz = pp[2];
    float* frow = a.feat + p * (idx_t)a.feat_stride;
    for (int v = q; v < a.V; v += 4) {
        const int IW = a.IW, IH = a.IH;
        const ColorTap t = color_project(px, py, pz, a.w2c + v * 16, a.Kmat + v * 9, IW, IH);
        const float* pl = a.img + (SMALL ? (int64_t)((__umul24(v * IH + t.y0, IW) + t.x0) << 2) : (((int64_t)v * IH + t.y0) * IW + t.x0) * 4);
        const float* zt = reinterpret_cast<const float*>(&g_zero_tap);
        const f32x4 t_nw = ldg16(pl);
        const f32x4 t_ne = ldg16(t.x1in ? pl + 4 : zt);
        const f32x4 t_sw = ldg16(t.y1in ? pl + (int64_t)IW * 4 : zt);
        const f32x4 t_se = ldg16((t.x1in && t.y1in) ? pl + (int64_t)IW * 4 + 4 : zt);
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 3; ++c) o[c] = color_blend(t, t_nw[c], t_ne[c], t_sw[c], t_se[c]);
        o[3] = color_mask(t);
        if (live) *reinterpret_cast<f32x4*>(frow + 8 + 4 * v) = o;
    }
    // ---- fold the volume taps in ATen's term order and roundings (sample_dev.h)
    if constexpr (ZFAST) {
        const f32x4 acc = zfast_fold_y0_lane(zt);                                 // valid in the y0 lanes
        if (live && q < 2) *reinterpret_cast<f32x4*>(frow + q * 4) = acc;
    } else {
        const f32x4 acc = trilinear_fold_x0_lane(vv, vw);                         // valid in the x0 lanes
        if (live && xc == 0) *reinterpret_cast<f32x4*>(frow + ch) = acc;
    }
    // ---- view-direction feature: lane 3 of quad number n < N handles ray n (no p / S: a 64-bit division costs every lane of
    // the wave dozens of instruction slots)
    if (q == 3 && a.dirs_out && p_raw < a.N) dir_feature_of(a.rays_dir + p_raw * 3, a.w2c, 1, a.dirs_out + p_raw * 3);   // reference view = view 0
}

extern "C" int mvsnerf_gather_fwd(const float* vol, int D, int H, int W, const float* imgs_nhwc4, int V, int IH, int IW,
                                  const float* w2c, const float* K, const float* pts, const float* ndc, int64_t N, int S,
                                  const float* rays_dir, float* feat, int feat_stride, float* dirs_out, int vol_layout, void* stream)
{
    if (vol_layout != MVSNERF_VOL_DHWC && vol_layout != MVSNERF_VOL_HWDC) return MVSNERF_EINVAL;
    if (!vol || !imgs_nhwc4 || !w2c || !K || !pts || !ndc || !feat || D < 1 || H < 1 || W < 1 || V < 1 || IH < 2 || IW < 2 || N < 0 || S < 1)
        return MVSNERF_EINVAL;
    if (dirs_out && !rays_dir) return MVSNERF_EINVAL;
    if (feat_stride < 8 + 4 * V) return MVSNERF_EINVAL;
    if ((feat_stride & 3) || !mvs_aligned16(feat) || !mvs_aligned16(vol) || !mvs_aligned16(imgs_nhwc4)) return MVSNERF_EALIGN;
    if (N == 0) return MVSNERF_OK;
    const int64_t P = N * S;
    const GatherArgs a{vol, D, H, W, imgs_nhwc4, V, IH, IW, w2c, K, pts, ndc, P, N, rays_dir, feat, feat_stride, dirs_out};
    const bool small = (int64_t)D * H < (1 << 24) && W < (1 << 24) && (int64_t)H * W < (1 << 24) && D < (1 << 24) && (int64_t)D * H * W * 8 < ((int64_t)1 << 31) &&
                       (int64_t)V * IH < (1 << 24) && IW < (1 << 24) && (int64_t)V * IH * IW * 4 < ((int64_t)1 << 31) &&
                       P * (int64_t)(feat_stride > 3 ? feat_stride : 3) < ((int64_t)1 << 31);
    const unsigned grid = mvs_cdiv(P * 4, 256);
    hipStream_t st = (hipStream_t)stream;
    if (vol_layout == MVSNERF_VOL_HWDC) {
        if (small) gather_fused_kernel<true, true><<<grid, 256, 0, st>>>(a);
        else gather_fused_kernel<false, true><<<grid, 256, 0, st>>>(a);
    } else {
        if (small) gather_fused_kernel<true, false><<<grid, 256, 0, st>>>(a);
        else gather_fused_kernel<false, false><<<grid, 256, 0, st>>>(a);
    }
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// Boundary transposes (C small): one thread per voxel, C strided reads / one contiguous C-vector write.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ncdhw_to_ndhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int64_t n_vox)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_vox) return;
    for (int c = 0; c < C; ++c) dst[i * C + c] = src[(int64_t)c * n_vox + i];
}
__global__ __launch_bounds__(256) void ndhwc_to_ncdhw_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int64_t n_vox)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_vox) return;
    for (int c = 0; c < C; ++c) dst[(int64_t)c * n_vox + i] = src[i * C + c];
}

extern "C" int mvsnerf_ncdhw_to_ndhwc(const float* src, float* dst, int C, int D, int H, int W, void* stream)
{
    if (!src || !dst || C < 1 || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    const int64_t n = (int64_t)D * H * W;
    ncdhw_to_ndhwc_kernel<<<mvs_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(src, dst, C, n);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
extern "C" int mvsnerf_ndhwc_to_ncdhw(const float* src, float* dst, int C, int D, int H, int W, void* stream)
{
    if (!src || !dst || C < 1 || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    const int64_t n = (int64_t)D * H * W;
    ndhwc_to_ncdhw_kernel<<<mvs_cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(src, dst, C, n);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// Stand-alone positional encoding (Embedder.embed models.py:47-51) for callers that use embed_fn
// directly; the MLP kernel embeds internally and never materialises this.
// out[p] = [x(d) | sin(x_c 2^f) f-major (d*L) | cos(...) (d*L)]
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ x, int64_t P, int d, int L, float* __restrict__ out)
{
    const int width = d * (1 + 2 * L);
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= P * width) return;
    const int64_t p = tid / width;
    const int k = (int)(tid - p * width);
    float v;
    if (k < d) v = x[p * d + k];
    else {
        const int j = (k - d) % (d * L), f = j / d, c = j - f * d;
        const float a = x[p * d + c] * (float)(1u << f);
        v = (k - d) < d * L ? sinf(a) : cosf(a);
    }
    out[tid] = v;
}

extern "C" int mvsnerf_posenc_fwd(const float* x, int64_t P, int d, int L, float* out, void* stream)
{
    if (!x || !out || P < 0 || d < 1 || L < 0 || L > 30) return MVSNERF_EINVAL;
    if (P == 0) return MVSNERF_OK;
    posenc_kernel<<<mvs_cdiv(P * d * (1 + 2 * L), 256), 256, 0, (hipStream_t)stream>>>(x, P, d, L, out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the trilinear lookup w.r.t. the volume: scatter-add of w_corner * g[p][c] into the channel-last
// gradient volume (zeros padding => out-of-range corners receive nothing).  Same quad mapping as the forward;
// float atomics => summation order (and the last bits) vary run to run - documented in DESIGN.md.
// ---------------------------------------------------------------------------------------------
// Consecutive samples of a ray (P is [ray][sample]: neighbours in p are neighbours along the ray) step ~1 depth plane inside the same (x, y)
// cell, so the z1 corners of sample s are the z0 corners of sample s + 1: the quad of s hands its z1 contributions to the quad of s + 1 (one DPP
// row shift by four lanes - inside a row of 16 lanes = 4 samples, so three of four neighbour pairs) which adds them to its own before the
// atomics; s then skips its z1 atomics.  A third of the 8.4 M atomics of a 1024 x 128 batch go away.
__global__ __launch_bounds__(256) void volume_sample_c8_bwd_kernel(
    int D, int H, int W, const float* __restrict__ ndc, int64_t P, const float* __restrict__ g, int g_stride, float* __restrict__ gvol)
{
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(tid & 3);
    const int xc = q >> 1, ch = (q & 1) * 4;
    const int64_t p_raw = tid >> 2;
    const bool live = p_raw < P;
    const int64_t p = live ? p_raw : P - 1;
    const float gx = ndc[p * 3 + 0] * 2.0f - 1.0f, gy = ndc[p * 3 + 1] * 2.0f - 1.0f, gz = ndc[p * 3 + 2] * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1), iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float wx = xc ? (ix - fx) : ((fx + 1.0f) - ix);
    const float cxf = fx + (float)xc;
    const bool x_ok = live && (cxf >= 0.0f) && (cxf <= (float)(W - 1));
    // the cell as integers (clamped far outside the volume: such cells never match a valid one and never pass the range checks below)
    const int cxi = (int)fminf(fmaxf(cxf, -4.0f), (float)W + 4.0f), fyi = (int)fminf(fmaxf(fy, -4.0f), (float)H + 4.0f), fzi = (int)fminf(fmaxf(fz, -4.0f), (float)D + 4.0f);
    f32x4 gv{0.f, 0.f, 0.f, 0.f};
    if (x_ok) gv = *reinterpret_cast<const f32x4*>(g + p * g_stride + ch);
    const float wy[2] = {(fy + 1.0f) - iy, iy - fy}, wz[2] = {(fz + 1.0f) - iz, iz - fz};
    f32x4 c[2][2];                                                // [zc][yc]
#pragma unroll
    for (int zc = 0; zc < 2; ++zc)
#pragma unroll
        for (int yc = 0; yc < 2; ++yc) c[zc][yc] = gv * (wx * wy[yc] * wz[zc]);
    // neighbours along the ray: lane - 4 (previous sample, same (xc, channel half)) and lane + 4, inside the 16-lane row
    constexpr int NONE = -(1 << 30);
    const int my_cell = x_ok ? (fyi * 4096 + cxi) : NONE + 1;     // W, H < 4096 (the volumes are a few hundred voxels wide)
    const int prev_cell = __builtin_amdgcn_update_dpp(NONE, my_cell, 0x114, 0xf, 0xf, false);      // row_shr:4
    const int prev_fz = __builtin_amdgcn_update_dpp(NONE, fzi, 0x114, 0xf, 0xf, false);
    const int next_cell = __builtin_amdgcn_update_dpp(NONE, my_cell, 0x104, 0xf, 0xf, false);      // row_shl:4
    const int next_fz = __builtin_amdgcn_update_dpp(NONE, fzi, 0x104, 0xf, 0xf, false);
    const bool take_prev = x_ok && prev_cell == my_cell && prev_fz + 1 == fzi;
    const bool give_next = x_ok && next_cell == my_cell && next_fz == fzi + 1;
#pragma unroll
    for (int yc = 0; yc < 2; ++yc)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mine = c[1][yc][k];                         // a float of its own: __builtin_bit_cast of a vector-element lvalue read element 0 for every k (hipcc 7.2)
            const float from_prev = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, mine), 0x114, 0xf, 0xf, false));
            if (take_prev) c[0][yc][k] += from_prev;
        }
    if (!x_ok) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int zc = k >> 1, yc = k & 1;
        if (zc == 1 && give_next) continue;                       // the next sample's quad carries these
        const float cyf = fy + (float)yc, czf = fz + (float)zc;
        if (!((cyf >= 0.0f) && (cyf <= (float)(H - 1)) && (czf >= 0.0f) && (czf <= (float)(D - 1)))) continue;
        float* dst = gvol + (((((int64_t)czf * H + (int)cyf) * W + (int)cxf) << 3) + ch);
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) atomicAdd(dst + cc, c[zc][yc][cc]);
    }
}

// Any channel count that is a multiple of 4 (the 8 + 4V-channel colour volume of --use_color_volume fine-tuning,
// train_mvs_nerf_finetuning_pl.py:72-82): one thread per (point, channel quad), eight corners each.
__global__ __launch_bounds__(256) void volume_sample_bwd_kernel(
    int D, int H, int W, int C, const float* __restrict__ ndc, int64_t P, const float* __restrict__ g, int g_stride, float* __restrict__ gvol)
{
    const int Q = C >> 2;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = tid / Q;
    const int ch = (int)(tid - p * Q) * 4;
    if (p >= P) return;
    const float gx = ndc[p * 3 + 0] * 2.0f - 1.0f, gy = ndc[p * 3 + 1] * 2.0f - 1.0f, gz = ndc[p * 3 + 2] * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1), iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + p * g_stride + ch);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int zc = k >> 2, yc = (k >> 1) & 1, xc = k & 1;
        const float cxf = fx + (float)xc, cyf = fy + (float)yc, czf = fz + (float)zc;
        if (!((cxf >= 0.0f) && (cxf <= (float)(W - 1)) && (cyf >= 0.0f) && (cyf <= (float)(H - 1)) && (czf >= 0.0f) && (czf <= (float)(D - 1)))) continue;
        const float w = (xc ? (ix - fx) : ((fx + 1.0f) - ix)) * (yc ? (iy - fy) : ((fy + 1.0f) - iy)) * (zc ? (iz - fz) : ((fz + 1.0f) - iz));
        float* dst = gvol + ((((int64_t)czf * H + (int)cyf) * W + (int)cxf) * C + ch);
#pragma unroll
        for (int c = 0; c < 4; ++c) atomicAdd(dst + c, gv[c] * w);
    }
}

extern "C" int mvsnerf_volume_sample_bwd(int D, int H, int W, int C, const float* ndc, int64_t P,
                                         const float* g, int g_stride, float* gvol, void* stream)
{
    if (!ndc || !g || !gvol || D < 1 || H < 1 || W < 1 || P < 0 || g_stride < C) return MVSNERF_EINVAL;
    if (C < 4 || (C & 3)) return MVSNERF_EUNSUPPORTED;
    if ((g_stride & 3) || !mvs_aligned16(g)) return MVSNERF_EALIGN;
    if (P == 0) return MVSNERF_OK;
    if (C == 8) volume_sample_c8_bwd_kernel<<<mvs_cdiv(P * 4, 256), 256, 0, (hipStream_t)stream>>>(D, H, W, ndc, P, g, g_stride, gvol);
    else volume_sample_bwd_kernel<<<mvs_cdiv(P * (C >> 2), 256), 256, 0, (hipStream_t)stream>>>(D, H, W, C, ndc, P, g, g_stride, gvol);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// The same scatter with an ORDER-INDEPENDENT reduction (round 6; the plane sweep has had one since round 4): float atomics make the volume gradient depend
// on the order in which samples arrive, and under use_amp one flipped last bit there grows 3-5x per layer through the bf16 encoder backward.  Here every
// contribution is rounded ONCE to 64-bit fixed point - scale 2^(40 - e) with max |g| < 2^e found on the device, so a contribution resolves 2^-40 of the largest
// one and 4 M of them fit a word - integer atomics commute, and a last pass adds the sums to gvol.  Two runs, and N ranks against one, give identical bits.
// workspace (int64 words, zeroed by the caller): [0] = bit pattern of max |g|, [8 ..] = one accumulator per volume element.
__global__ __launch_bounds__(256) void absmax_bits_kernel(const float* __restrict__ g, int64_t P, int C, int g_stride, unsigned* __restrict__ out)
{
    unsigned m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < P * C; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i / C;
        m = max(m, __float_as_uint(fabsf(g[p * g_stride + (i - p * C)])) & 0x7fffffffu);
    }
    for (int o = 32; o; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

__device__ __forceinline__ int det_exponent(unsigned max_bits)        // e with max |g| < 2^e (finite, non-zero input)
{
    int e;
    (void)frexpf(__uint_as_float(max_bits), &e);
    return e;
}

__global__ __launch_bounds__(256) void volume_sample_bwd_det_scatter_kernel(
    int D, int H, int W, int C, const float* __restrict__ ndc, int64_t P, const float* __restrict__ g, int g_stride,
    const unsigned* __restrict__ max_bits, unsigned long long* __restrict__ acc)
{
    const unsigned mb = *max_bits;
    if (mb == 0 || mb >= 0x7f800000u) return;                       // all-zero gradient; a non-finite one is left to the float path's semantics (finish reports nothing)
    const double scale = ldexp(1.0, 40 - det_exponent(mb));
    const int Q = C >> 2;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t p = tid / Q;
    const int ch = (int)(tid - p * Q) * 4;
    if (p >= P) return;
    const float gx = ndc[p * 3 + 0] * 2.0f - 1.0f, gy = ndc[p * 3 + 1] * 2.0f - 1.0f, gz = ndc[p * 3 + 2] * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1), iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + p * g_stride + ch);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int zc = k >> 2, yc = (k >> 1) & 1, xc = k & 1;
        const float cxf = fx + (float)xc, cyf = fy + (float)yc, czf = fz + (float)zc;
        if (!((cxf >= 0.0f) && (cxf <= (float)(W - 1)) && (cyf >= 0.0f) && (cyf <= (float)(H - 1)) && (czf >= 0.0f) && (czf <= (float)(D - 1)))) continue;
        const float w = (xc ? (ix - fx) : ((fx + 1.0f) - ix)) * (yc ? (iy - fy) : ((fy + 1.0f) - iy)) * (zc ? (iz - fz) : ((fz + 1.0f) - iz));
        unsigned long long* dst = acc + ((((int64_t)czf * H + (int)cyf) * W + (int)cxf) * C + ch);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const long long q = __double2ll_rn((double)(gv[c] * w) * scale);        // the float product of the atomic path, rounded once more to the fixed grid
            if (q) atomicAdd(dst + c, (unsigned long long)q);                        // (two's complement: unsigned addition is the signed one)
        }
    }
}

__global__ __launch_bounds__(256) void volume_sample_bwd_det_finish_kernel(const long long* __restrict__ acc, int64_t n, const unsigned* __restrict__ max_bits,
                                                                           float* __restrict__ gvol)
{
    const unsigned mb = *max_bits;
    if (mb == 0 || mb >= 0x7f800000u) return;
    const double inv = ldexp(1.0, det_exponent(mb) - 40);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const long long q = acc[i];
        if (q) gvol[i] += (float)((double)q * inv);
    }
}

extern "C" size_t mvsnerf_volume_sample_bwd_det_workspace_words(int D, int H, int W, int C)
{
    if (D < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return 0;
    return (size_t)8 + (size_t)D * H * W * C;
}

extern "C" int mvsnerf_volume_sample_bwd_det(int D, int H, int W, int C, const float* ndc, int64_t P, const float* g, int g_stride, float* gvol,
                                             void* workspace_zeroed, void* stream)
{
    if (!ndc || !g || !gvol || !workspace_zeroed || D < 1 || H < 1 || W < 1 || P < 0 || g_stride < C) return MVSNERF_EINVAL;
    if (C < 4 || (C & 3)) return MVSNERF_EUNSUPPORTED;
    if ((g_stride & 3) || !mvs_aligned16(g) || (reinterpret_cast<uintptr_t>(workspace_zeroed) & 7)) return MVSNERF_EALIGN;
    if (P == 0) return MVSNERF_OK;
    hipStream_t st = (hipStream_t)stream;
    unsigned* mb = reinterpret_cast<unsigned*>(workspace_zeroed);
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(workspace_zeroed) + 8;
    const int64_t n = (int64_t)D * H * W * C;
    absmax_bits_kernel<<<1024, 256, 0, st>>>(g, P, C, g_stride, mb);
    MVS_LAUNCH_CHECK();
    volume_sample_bwd_det_scatter_kernel<<<mvs_cdiv(P * (C >> 2), 256), 256, 0, st>>>(D, H, W, C, ndc, P, g, g_stride, mb, acc);
    MVS_LAUNCH_CHECK();
    volume_sample_bwd_det_finish_kernel<<<4096, 256, 0, st>>>(reinterpret_cast<const long long*>(acc), n, mb, gvol);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// Ray generation: the arithmetic of build_rays / build_rays_test (utils.py:86-108, 148-297) after the RNG draws.
// Pixel ids (CPU RNG) and the stratified jitter (device RNG) stay with the caller, so ray indices are bit-exact with
// the reference; everything downstream of them is one kernel instead of ~25 ATen launches per batch:
//   dirs   = [(x-cx)/fx, (y-cy)/fy, 1] @ c2w[:3,:3]^T                      (utils.py:101-104)
//   z_s    = near*(1-t_s) + far*t_s, t = linspace(0,1,S) [+ stratified jitter]  (:211-221 / :279-282)
//   pts    = o + z*d                                                        (:223)
//   ndc    = get_ndc_coordinate(w2c_ref, K_ref, pts, (W-1,H-1), near_ref, far_ref, pad)   (:112-146)
// One thread per sample.
// ---------------------------------------------------------------------------------------------
struct RayGenArgs {
    const float* xs; const float* ys;      // [N] pixel ids as floats, or null: row-major ids first_pixel + n
    int64_t first_pixel; int W_img, H_img;   // target view (row-major pixel ids)
    int W_ref, H_ref;                        // size of the view the NDC coordinates refer to (the reference assumes == target)
    const float* Kt; const float* c2w;     // target camera: intrinsics [3][3], c2w [4][4]   (device, wave-uniform loads)
    const float* Kr; const float* w2c;     // reference camera: intrinsics [3][3], w2c [4][4]
    const float* nf_tgt; const float* nf_ref;   // [2] = (near, far) of the target / reference view
    int pad, lindisp;
    const float* t_rand;                    // [N][S] or null
    int64_t N; int S;
    float* rays_pts; float* rays_dir; float* rays_ndc; float* z_vals; float* pix;   // pix: [2][N] = (ys, xs), may be null
    // build_rays' per-pixel extras (mvsnerf_raygen_train_fwd; all null / 0 for the plain entry)
    const float* tgt_img;                   // [3][H_img][W_img] target view: colours gathered at the pixel ids (utils.py:190-192)
    const float* depth_map;                 // [H_img][W_img] ground-truth depth of the target view or null (utils.py:194-196)
    const float* z_map;                     // depth_mode 2: [H_img][W_img] map the single depth candidate is read from (utils.py:199-200)
    int depth_mode;                         // 0: near/far of the view; 1: importanceSampling (depth -+ 0.1, :202-204); 2: with_depth (S == 1)
    float* colors; float* rays_depth;       // [N][3], [N] (null without depth_map)
};

__device__ __forceinline__ float linspace01(int i, int steps)
{
    // torch.linspace(0,1,steps): symmetric evaluation (start + i*step below the midpoint, end - (steps-1-i)*step above)
    if (steps == 1) return 0.0f;
    const float step = 1.0f / (float)(steps - 1);
    return i < steps / 2 ? step * (float)i : 1.0f - step * (float)(steps - 1 - i);
}

__global__ __launch_bounds__(256) void raygen_kernel(RayGenArgs a)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.N * a.S) return;
    // (three 64-bit divisions per SAMPLE were most of this kernel's instructions: sample and pixel ids below 2^32 - every frame - take 32-bit ones)
    int64_t n;
    if (((uint64_t)t >> 32) == 0) n = (int64_t)((unsigned)t / (unsigned)a.S); else n = t / a.S;
    const int s = (int)(t - n * a.S);
    float x, y;
    if (a.xs) { x = a.xs[n]; y = a.ys[n]; }
    else {
        const int64_t p = a.first_pixel + n;
        if (((uint64_t)p >> 32) == 0) { const unsigned q = (unsigned)p / (unsigned)a.W_img; y = (float)q; x = (float)((unsigned)p - q * (unsigned)a.W_img); }
        else { y = (float)(p / a.W_img); x = (float)(p % a.W_img); }
    }
    float near = a.nf_tgt[0], far = a.nf_tgt[1];
    const float near_ref = a.nf_ref[0], far_ref = a.nf_ref[1];
    const int64_t pix_off = (int64_t)y * a.W_img + (int64_t)x;            // pixel_coordinates.long() (:189): ids are non-negative integers
    if (a.depth_mode == 1) { const float d = a.depth_map[pix_off]; near = d - 0.1f; far = d + 0.1f; }      // utils.py:203
    const float cxd = (x - a.Kt[2]) / a.Kt[0], cyd = (y - a.Kt[5]) / a.Kt[4];
    const float dx = fmaf(1.0f, a.c2w[2], fmaf(cyd, a.c2w[1], cxd * a.c2w[0]));
    const float dy = fmaf(1.0f, a.c2w[6], fmaf(cyd, a.c2w[5], cxd * a.c2w[4]));
    const float dz = fmaf(1.0f, a.c2w[10], fmaf(cyd, a.c2w[9], cxd * a.c2w[8]));
    auto zplain = [&](int i) {
        const float tv = linspace01(i, a.S);
        return a.lindisp ? 1.0f / (1.0f / near * (1.0f - tv) + 1.0f / far * tv) : near * (1.0f - tv) + far * tv;
    };
    float z = a.depth_mode == 2 ? a.z_map[pix_off] : zplain(s);           // :200: one candidate per ray, no stratification
    if (a.t_rand && a.depth_mode != 2) {
        const float lo = s == 0 ? z : 0.5f * (z + zplain(s - 1));
        const float up = s == a.S - 1 ? z : 0.5f * (zplain(s + 1) + z);
        z = lo + (up - lo) * a.t_rand[t];
    }
    const float ox = a.c2w[3], oy = a.c2w[7], oz = a.c2w[11];
    const float px = ox + z * dx, py = oy + z * dy, pz = oz + z * dz;
    // reference camera: R x + t, K p, /z, /(W-1,H-1), depth normalisation, pad re-scale
    const float cx = fmaf(pz, a.w2c[2], fmaf(py, a.w2c[1], px * a.w2c[0])) + a.w2c[3];
    const float cy = fmaf(pz, a.w2c[6], fmaf(py, a.w2c[5], px * a.w2c[4])) + a.w2c[7];
    const float cz = fmaf(pz, a.w2c[10], fmaf(py, a.w2c[9], px * a.w2c[8])) + a.w2c[11];
    const float qx = fmaf(cz, a.Kr[2], fmaf(cy, a.Kr[1], cx * a.Kr[0]));
    const float qy = fmaf(cz, a.Kr[5], fmaf(cy, a.Kr[4], cx * a.Kr[3]));
    const float qz = fmaf(cz, a.Kr[8], fmaf(cy, a.Kr[7], cx * a.Kr[6]));
    float nx = (qx / qz + 0.0f) / (float)(a.W_ref - 1);
    float ny = (qy / qz + 0.0f) / (float)(a.H_ref - 1);
    const float nz = a.lindisp ? (1.0f / qz - 1.0f / near_ref) / (1.0f / far_ref - 1.0f / near_ref)
                               : (qz - near_ref) / (far_ref - near_ref);
    if (a.pad > 0) {
        const float Wf = (float)a.W_ref / 4.0f, Hf = (float)a.H_ref / 4.0f;          // (inv_scale+1)/4
        ny = ny * Hf / (Hf + (float)(a.pad * 2)) + (float)a.pad / (Hf + (float)(a.pad * 2));
        nx = nx * Wf / (Wf + (float)(a.pad * 2)) + (float)a.pad / (Wf + (float)(a.pad * 2));
    }
    a.rays_pts[t * 3 + 0] = px; a.rays_pts[t * 3 + 1] = py; a.rays_pts[t * 3 + 2] = pz;
    a.rays_ndc[t * 3 + 0] = nx; a.rays_ndc[t * 3 + 1] = ny; a.rays_ndc[t * 3 + 2] = nz;
    a.z_vals[t] = z;
    if (s == 0) {
        a.rays_dir[n * 3 + 0] = dx; a.rays_dir[n * 3 + 1] = dy; a.rays_dir[n * 3 + 2] = dz;
        if (a.pix) { a.pix[n] = y; a.pix[a.N + n] = x; }
        if (a.colors) {
            const int64_t plane = (int64_t)a.H_img * a.W_img;
            a.colors[n * 3 + 0] = a.tgt_img[pix_off]; a.colors[n * 3 + 1] = a.tgt_img[plane + pix_off]; a.colors[n * 3 + 2] = a.tgt_img[2 * plane + pix_off];
        }
        if (a.rays_depth) a.rays_depth[n] = a.depth_map[pix_off];
    }
}

static int raygen_launch(const float* xs, const float* ys, int64_t first_pixel, int W_img, int H_img, int W_ref, int H_ref,
                         const float* K_tgt, const float* c2w_tgt, const float* K_ref, const float* w2c_ref,
                         const float* near_far_tgt, const float* near_far_ref, int pad, int lindisp,
                         const float* t_rand, int64_t N, int S,
                         float* rays_pts, float* rays_dir, float* rays_ndc, float* z_vals, float* pix,
                         const float* tgt_img, const float* depth_map, const float* z_map, int depth_mode, float* colors, float* rays_depth, void* stream);

extern "C" int mvsnerf_raygen_fwd(const float* xs, const float* ys, int64_t first_pixel, int W_img, int H_img, int W_ref, int H_ref,
                                  const float* K_tgt, const float* c2w_tgt, const float* K_ref, const float* w2c_ref,
                                  const float* near_far_tgt, const float* near_far_ref, int pad, int lindisp,
                                  const float* t_rand, int64_t N, int S,
                                  float* rays_pts, float* rays_dir, float* rays_ndc, float* z_vals, float* pix, void* stream)
{
    return raygen_launch(xs, ys, first_pixel, W_img, H_img, W_ref, H_ref, K_tgt, c2w_tgt, K_ref, w2c_ref, near_far_tgt, near_far_ref, pad, lindisp,
                         t_rand, N, S, rays_pts, rays_dir, rays_ndc, z_vals, pix, nullptr, nullptr, nullptr, 0, nullptr, nullptr, stream);
}

// build_rays of one training step in ONE launch (utils.py:148-241 downstream of the RNG draws): the plain ray generation plus the
// per-pixel gathers (target colours :190-192, ground-truth depth :194-196) and the two per-pixel depth ranges
// (depth_mode 1 = importanceSampling :202-204, 2 = with_depth :199-200 with S == 1).
extern "C" int mvsnerf_raygen_train_fwd(const float* xs, const float* ys, int W_img, int H_img, int W_ref, int H_ref,
                                        const float* K_tgt, const float* c2w_tgt, const float* K_ref, const float* w2c_ref,
                                        const float* near_far_tgt, const float* near_far_ref, int pad, int lindisp,
                                        const float* t_rand, int64_t N, int S,
                                        const float* tgt_img, const float* depth_map, const float* z_map, int depth_mode,
                                        float* rays_pts, float* rays_dir, float* rays_ndc, float* z_vals, float* pix, float* colors, float* rays_depth,
                                        void* stream)
{
    if (!xs || !ys || !tgt_img || !colors || depth_mode < 0 || depth_mode > 2) return MVSNERF_EINVAL;
    if ((depth_mode == 1 && !depth_map) || (depth_mode == 2 && (!z_map || S != 1)) || (rays_depth && !depth_map)) return MVSNERF_EINVAL;
    return raygen_launch(xs, ys, 0, W_img, H_img, W_ref, H_ref, K_tgt, c2w_tgt, K_ref, w2c_ref, near_far_tgt, near_far_ref, pad, lindisp,
                         t_rand, N, S, rays_pts, rays_dir, rays_ndc, z_vals, pix, tgt_img, depth_map, z_map, depth_mode, colors, rays_depth, stream);
}

static int raygen_launch(const float* xs, const float* ys, int64_t first_pixel, int W_img, int H_img, int W_ref, int H_ref,
                         const float* K_tgt, const float* c2w_tgt, const float* K_ref, const float* w2c_ref,
                         const float* near_far_tgt, const float* near_far_ref, int pad, int lindisp,
                         const float* t_rand, int64_t N, int S,
                         float* rays_pts, float* rays_dir, float* rays_ndc, float* z_vals, float* pix,
                         const float* tgt_img, const float* depth_map, const float* z_map, int depth_mode, float* colors, float* rays_depth, void* stream)
{
    if (!K_tgt || !c2w_tgt || !K_ref || !w2c_ref || !near_far_tgt || !near_far_ref || !rays_pts || !rays_dir || !rays_ndc || !z_vals || N < 0 || S < 1) return MVSNERF_EINVAL;
    if ((xs == nullptr) != (ys == nullptr) || W_img < 2 || H_img < 2) return MVSNERF_EINVAL;
    if (W_ref <= 0) W_ref = W_img;
    if (H_ref <= 0) H_ref = H_img;
    if (W_ref < 2 || H_ref < 2) return MVSNERF_EINVAL;
    if (N == 0) return MVSNERF_OK;
    RayGenArgs a;
    a.xs = xs; a.ys = ys; a.first_pixel = first_pixel; a.W_img = W_img; a.H_img = H_img; a.W_ref = W_ref; a.H_ref = H_ref;
    a.Kt = K_tgt; a.c2w = c2w_tgt; a.Kr = K_ref; a.w2c = w2c_ref; a.nf_tgt = near_far_tgt; a.nf_ref = near_far_ref;
    a.pad = pad; a.lindisp = lindisp; a.t_rand = t_rand; a.N = N; a.S = S;
    a.rays_pts = rays_pts; a.rays_dir = rays_dir; a.rays_ndc = rays_ndc; a.z_vals = z_vals; a.pix = pix;
    a.tgt_img = tgt_img; a.depth_map = depth_map; a.z_map = z_map; a.depth_mode = depth_mode; a.colors = colors; a.rays_depth = rays_depth;
    raygen_kernel<<<mvs_cdiv(N * S, 256), 256, 0, (hipStream_t)stream>>>(a);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
