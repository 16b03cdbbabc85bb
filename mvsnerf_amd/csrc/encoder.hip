// Scene-encode kernels (L1a): plane-sweep homography warp + variance cost volume, and the CostRegNet
// 3-D U-Net building blocks (k3 conv, k3 s2 transposed conv, train-mode InPlaceABN).
//
// Layouts (all fp32, channel-last so that a voxel's channels are one contiguous vector):
//   source features   feat[v][y][x][32]          (NCHW from the 2-D FeatureNet is transposed once)
//   source thumbnails img[v][y][x][4]            (rgb + pad)
//   cost volume       cost[d][y][x][CP]          CP = round_up4(3V+32): [ref rgb | src rgb.. | variance(32) | 0-pad]
//   activations       act[d][y][x][C]            C in {8,16,32,64}
// InPlaceABN is applied lazily: a conv writes its RAW output; a stats kernel turns the batch statistics
// into per-channel (scale, shift); every consumer applies leaky_relu(x*scale+shift) while loading.
// Skip additions (models.py:762-766) are the sum of two such lazily-activated tensors.
#include "common.h"

// =============================================================================================
// small layout / resize helpers
// =============================================================================================
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               int N, int C, int64_t HW, int Cpad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over N*HW*Cpad
    if (i >= (int64_t)N * HW * Cpad) return;
    const int c = (int)(i % Cpad);
    const int64_t r = i / Cpad;
    const int64_t n = r / HW, p = r - n * HW;
    dst[i] = c < C ? src[(n * C + c) * HW + p] : 0.0f;
}

extern "C" int mvsnerf_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int Cpad, void* stream)
{
    if (!src || !dst || N < 1 || C < 1 || H < 1 || W < 1 || Cpad < C) return MVSNERF_EINVAL;
    const int64_t tot = (int64_t)N * H * W * Cpad;
    nchw_to_nhwc_pad_kernel<<<mvs_cdiv(tot, 256), 256, 0, (hipStream_t)stream>>>(src, dst, N, C, (int64_t)H * W, Cpad);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// F.interpolate(mode='bilinear', align_corners=False) (models.py:859), planes [NC][Hi][Wi] -> [NC][Ho][Wo]
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              int NC, int Hi, int Wi, int Ho, int Wo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)NC * Ho * Wo) return;
    const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
    const int64_t n = i / ((int64_t)Wo * Ho);
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;     // area_pixel_compute_scale
    float fy = sh * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = sw * ((float)x + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    const float* p = src + n * Hi * Wi;
    dst[i] = hy * (hx * p[(int64_t)y0 * Wi + x0] + lx * p[(int64_t)y0 * Wi + x1]) +
             ly * (hx * p[(int64_t)y1 * Wi + x0] + lx * p[(int64_t)y1 * Wi + x1]);
}

extern "C" int mvsnerf_resize_bilinear(const float* src, float* dst, int NC, int Hi, int Wi, int Ho, int Wo, void* stream)
{
    if (!src || !dst || NC < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return MVSNERF_EINVAL;
    resize_bilinear_kernel<<<mvs_cdiv((int64_t)NC * Ho * Wo, 256), 256, 0, (hipStream_t)stream>>>(src, dst, NC, Hi, Wi, Ho, Wo);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// =============================================================================================
// plane sweep: homo_warp (utils.py:580-630) + build_volume_costvar[_img] (models.py:787-893) in ONE pass.
// One thread per voxel (d,y,x), x fastest.  Reads ~1 KB of L2-resident source features per voxel,
// writes the voxel's CP-channel vector once (the reference moves ~10 GB for the same result).
// =============================================================================================
struct SweepGeom { float R[9]; float T[3]; };    // one source view: proj_mat[:, :3], proj_mat[:, 3]

template <int C>   // feature channels (32)
__global__ __launch_bounds__(256) void planesweep_kernel(
    const float* __restrict__ feat,   // [V][H][W][C]
    const float* __restrict__ img,    // [V][H][W][4] or null
    const float* __restrict__ proj,   // [V][3][4]
    const float* __restrict__ depth,  // [D]
    int V, int H, int W, int D, int pad,
    float* __restrict__ cost, int CP,   // [D][Hp][Wp][CP]
    float* __restrict__ masks,          // with img: [V][D][Hp][Wp] per-view; else [D][Hp][Wp] count
    int with_img)
{
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int64_t nvox = (int64_t)D * Hp * Wp;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    const int x = (int)(i % Wp), y = (int)((i / Wp) % Hp), d = (int)(i / ((int64_t)Wp * Hp));
    const float u = (float)(x - pad), v = (float)(y - pad);     // utils.py:603-605
    const float dep = depth[d];
    const bool interior = x >= pad && x < W + pad && y >= pad && y < H + pad;

    float s[C], s2[C];
    float* o = cost + i * CP;
    const int c_var = with_img ? 3 * V : 0;
    if (interior) {                                              // ref volume: zero-padded ref feature (models.py:856,862)
        const f32x4* r = reinterpret_cast<const f32x4*>(feat + ((int64_t)(y - pad) * W + (x - pad)) * C);
#pragma unroll
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const f32x4 t = r[c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[c4 * 4 + k] = t[k]; s2[c4 * 4 + k] = t[k] * t[k]; }
        }
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) { s[c] = 0.f; s2[c] = 0.f; }
    }
    if (with_img) {                                              // channels 0:3 = ref thumbnail, border := 0 (models.py:858-860)
        const float* ri = img + ((int64_t)(y - pad) * W + (x - pad)) * 4;
        o[0] = interior ? ri[0] : 0.f; o[1] = interior ? ri[1] : 0.f; o[2] = interior ? ri[2] : 0.f;
        masks[i] = 1.0f;                                         // view 0 mask (models.py:869)
    }
    float cnt = 1.0f;
    for (int vv = 1; vv < V; ++vv) {
        const float* P = proj + vv * 12;
        // utils.py:612  R @ (u,v,1) + T/depth   (k-ordered fma chain like the reference's bmm)
        const float p0 = fmaf(P[2], 1.0f, fmaf(P[1], v, P[0] * u)) + P[3] / dep;
        const float p1 = fmaf(P[6], 1.0f, fmaf(P[5], v, P[4] * u)) + P[7] / dep;
        const float p2 = fmaf(P[10], 1.0f, fmaf(P[9], v, P[8] * u)) + P[11] / dep;
        const float gx = (p0 / p2) / ((float)(W - 1) / 2.0f) - 1.0f;          // :617-620 (un-padded W,H)
        const float gy = (p1 / p2) / ((float)(H - 1) / 2.0f) - 1.0f;
        const float m = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;   // models.py:875-876
        cnt += m;
        if (with_img) masks[(int64_t)vv * nvox + i] = m;
        // F.grid_sample bilinear, zeros padding, align_corners=True (utils.py:625)
        const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
        const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
        const float fx = floorf(ix), fy = floorf(iy);
        const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
        const bool x0in = fx >= 0.f && fx <= (float)(W - 1), x1in = fx + 1.f >= 0.f && fx + 1.f <= (float)(W - 1);
        const bool y0in = fy >= 0.f && fy <= (float)(H - 1), y1in = fy + 1.f >= 0.f && fy + 1.f <= (float)(H - 1);
        const float w_nw = (x0in && y0in) ? wx0 * wy0 : 0.f, w_ne = (x1in && y0in) ? wx1 * wy0 : 0.f;
        const float w_sw = (x0in && y1in) ? wx0 * wy1 : 0.f, w_se = (x1in && y1in) ? wx1 * wy1 : 0.f;
        // clamp the tap addresses (weights are already zero where a tap is outside)
        const bool any = (x0in || x1in) && (y0in || y1in);
        const int xa = any ? min(max((int)fx, 0), W - 1) : 0, xb = any ? min(max((int)fx + 1, 0), W - 1) : 0;
        const int ya = any ? min(max((int)fy, 0), H - 1) : 0, yb = any ? min(max((int)fy + 1, 0), H - 1) : 0;
        const float* fb = feat + (int64_t)vv * H * W * C;
        const f32x4* t_nw = reinterpret_cast<const f32x4*>(fb + ((int64_t)ya * W + xa) * C);
        const f32x4* t_ne = reinterpret_cast<const f32x4*>(fb + ((int64_t)ya * W + xb) * C);
        const f32x4* t_sw = reinterpret_cast<const f32x4*>(fb + ((int64_t)yb * W + xa) * C);
        const f32x4* t_se = reinterpret_cast<const f32x4*>(fb + ((int64_t)yb * W + xb) * C);
#pragma unroll
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const f32x4 a = t_nw[c4], b = t_ne[c4], c_ = t_sw[c4], e = t_se[c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float wv = ((a[k] * w_nw + b[k] * w_ne) + c_[k] * w_sw) + e[k] * w_se;   // ATen's nw,ne,sw,se order
                s[c4 * 4 + k] += wv;                             // models.py:880
                s2[c4 * 4 + k] += wv * wv;                       // :881
            }
        }
        if (with_img) {                                          // warped thumbnail with the same grid (models.py:872)
            const float* ib = img + (int64_t)vv * H * W * 4;
            const f32x4 a = *reinterpret_cast<const f32x4*>(ib + ((int64_t)ya * W + xa) * 4);
            const f32x4 b = *reinterpret_cast<const f32x4*>(ib + ((int64_t)ya * W + xb) * 4);
            const f32x4 c_ = *reinterpret_cast<const f32x4*>(ib + ((int64_t)yb * W + xa) * 4);
            const f32x4 e = *reinterpret_cast<const f32x4*>(ib + ((int64_t)yb * W + xb) * 4);
#pragma unroll
            for (int k = 0; k < 3; ++k) o[3 * vv + k] = ((a[k] * w_nw + b[k] * w_ne) + c_[k] * w_sw) + e[k] * w_se;
        }
    }
    if (!with_img) masks[i] = cnt;                               // build_volume_costvar returns the count (models.py:821)
    const float inv = 1.0f / cnt;                                // models.py:889
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float mean = s[c] * inv;
        o[c_var + c] = s2[c] * inv - mean * mean;                // :890
    }
    for (int c = c_var + C; c < CP; ++c) o[c] = 0.0f;
}

extern "C" int mvsnerf_planesweep_costvar_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                              int V, int C, int H, int W, int D, int pad, float* cost, int CP, float* masks,
                                              int with_img, void* stream)
{
    if (!feats_cl || !proj || !depth || !cost || !masks || V < 1 || H < 2 || W < 2 || D < 1 || pad < 0) return MVSNERF_EINVAL;
    if (with_img && !imgs_cl) return MVSNERF_EINVAL;
    if (C != 32) return MVSNERF_EUNSUPPORTED;
    if (CP < (with_img ? 3 * V : 0) + C) return MVSNERF_EINVAL;
    if (!mvs_aligned16(feats_cl) || (imgs_cl && !mvs_aligned16(imgs_cl))) return MVSNERF_EALIGN;
    const int64_t nvox = (int64_t)D * (H + 2 * pad) * (W + 2 * pad);
    planesweep_kernel<32><<<mvs_cdiv(nvox, 256), 256, 0, (hipStream_t)stream>>>(feats_cl, imgs_cl, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// stand-alone homo_warp (utils.py:580-630): one source view, NCHW in, (C,D,Hp,Wp) out + grid (D,Hp*Wp,2)
__global__ __launch_bounds__(256) void homo_warp_kernel(const float* __restrict__ src, const float* __restrict__ P, const float* __restrict__ depth,
                                                        const float* __restrict__ grid_in, int C, int H, int W, int D, int pad,
                                                        float* __restrict__ out, float* __restrict__ grid_out)
{
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int64_t nvox = (int64_t)D * Hp * Wp;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nvox) return;
    float gx, gy;
    if (grid_in) { gx = grid_in[i * 2]; gy = grid_in[i * 2 + 1]; }
    else {
        const int x = (int)(i % Wp), y = (int)((i / Wp) % Hp), d = (int)(i / ((int64_t)Wp * Hp));
        const float u = (float)(x - pad), v = (float)(y - pad), dep = depth[d];
        const float p0 = fmaf(P[2], 1.0f, fmaf(P[1], v, P[0] * u)) + P[3] / dep;
        const float p1 = fmaf(P[6], 1.0f, fmaf(P[5], v, P[4] * u)) + P[7] / dep;
        const float p2 = fmaf(P[10], 1.0f, fmaf(P[9], v, P[8] * u)) + P[11] / dep;
        gx = (p0 / p2) / ((float)(W - 1) / 2.0f) - 1.0f;
        gy = (p1 / p2) / ((float)(H - 1) / 2.0f) - 1.0f;
    }
    if (grid_out) { grid_out[i * 2] = gx; grid_out[i * 2 + 1] = gy; }
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
    const bool x0in = fx >= 0.f && fx <= (float)(W - 1), x1in = fx + 1.f >= 0.f && fx + 1.f <= (float)(W - 1);
    const bool y0in = fy >= 0.f && fy <= (float)(H - 1), y1in = fy + 1.f >= 0.f && fy + 1.f <= (float)(H - 1);
    const int x0 = (int)fx, y0 = (int)fy;
    for (int c = 0; c < C; ++c) {
        const float* pl = src + (int64_t)c * H * W;
        float acc = 0.f;
        if (x0in && y0in) acc += pl[(int64_t)y0 * W + x0] * (wx0 * wy0);
        if (x1in && y0in) acc += pl[(int64_t)y0 * W + x0 + 1] * (wx1 * wy0);
        if (x0in && y1in) acc += pl[(int64_t)(y0 + 1) * W + x0] * (wx0 * wy1);
        if (x1in && y1in) acc += pl[(int64_t)(y0 + 1) * W + x0 + 1] * (wx1 * wy1);
        out[(int64_t)c * nvox + i] = acc;
    }
}

extern "C" int mvsnerf_homo_warp_fwd(const float* src_nchw, const float* proj, const float* depth, const float* grid_in,
                                     int C, int H, int W, int D, int pad, float* warped, float* grid_out, void* stream)
{
    if (!src_nchw || !warped || C < 1 || H < 2 || W < 2 || D < 1 || pad < 0) return MVSNERF_EINVAL;
    if (!grid_in && (!proj || !depth)) return MVSNERF_EINVAL;
    const int64_t nvox = (int64_t)D * (H + 2 * pad) * (W + 2 * pad);
    homo_warp_kernel<<<mvs_cdiv(nvox, 256), 256, 0, (hipStream_t)stream>>>(src_nchw, proj, depth, grid_in, C, H, W, D, pad, warped, grid_out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// =============================================================================================
// CostRegNet building blocks (models.py:674-685, 725-769)
// =============================================================================================
// lazily-activated operand: value = leaky(x*scale[c]+shift[c]) (scale == null: identity, no activation),
// optionally + a second such tensor (the U-Net skip sums).
struct ActSrc { const float* x; const float* scale; const float* shift; };

__device__ __forceinline__ float act_apply(float x, float sc, float sh) { const float y = fmaf(x, sc, sh); return y > 0.f ? y : 0.01f * y; }

template <int CIN>
__device__ __forceinline__ void load_act4(const ActSrc& a, const ActSrc& b, int64_t vox, int ld, int c, f32x4& out)
{
    out = *reinterpret_cast<const f32x4*>(a.x + vox * ld + c);
    if (a.scale) {
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] = act_apply(out[k], a.scale[c + k], a.shift[c + k]);
    }
    if (b.x) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(b.x + vox * ld + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) out[k] += act_apply(t[k], b.scale[c + k], b.shift[c + k]);
    }
}

// weights re-laid as w[tap][ci][co] (co fastest) so that a thread's CT output channels are contiguous and
// wave-uniform => the compiler fetches them through the scalar cache (s_load) and feeds v_fma from SGPRs.
__global__ void conv3d_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int cin_pad, int transposed, float* __restrict__ packed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 27 * cin_pad * Cout;
    if (i >= total) return;
    const int co = i % Cout, ci = (i / Cout) % cin_pad, tap = i / (Cout * cin_pad);
    float v = 0.f;
    if (ci < Cin) v = transposed ? w[((int64_t)ci * Cout + co) * 27 + tap]      // ConvTranspose3d weight (Cin,Cout,3,3,3)
                                 : w[((int64_t)co * Cin + ci) * 27 + tap];      // Conv3d weight (Cout,Cin,3,3,3)
    packed[i] = v;
}

extern "C" int mvsnerf_conv3d_pack_weights(const float* w, int Cout, int Cin, int cin_pad, int transposed, float* packed, void* stream)
{
    if (!w || !packed || Cout < 1 || Cin < 1 || cin_pad < Cin) return MVSNERF_EINVAL;
    conv3d_pack_kernel<<<mvs_cdiv(27 * cin_pad * Cout, 256), 256, 0, (hipStream_t)stream>>>(w, Cout, Cin, cin_pad, transposed, packed);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// Direct 3x3x3 convolution, padding 1, stride S.  One thread = one output voxel x CT output channels.
// Neighbouring threads re-read each other's input voxels through L1/L2 (27-fold reuse).
template <int CIN, int CT, int S>
__global__ __launch_bounds__(256) void conv3d_k3_kernel(ActSrc a, ActSrc b, int ld, int Di, int Hi, int Wi,
                                                       const float* __restrict__ wp, int Cout,
                                                       float* __restrict__ out, int Do, int Ho, int Wo)
{
    const int64_t nvox = (int64_t)Do * Ho * Wo;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = blockIdx.y * CT;                       // first output channel of this thread
    if (i >= nvox) return;
    const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), z = (int)(i / ((int64_t)Wo * Ho));
    float acc[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) acc[k] = 0.f;
    for (int tap = 0; tap < 27; ++tap) {
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        const int zi = z * S - 1 + dz, yi = y * S - 1 + dy, xi = x * S - 1 + dx;
        const bool in = zi >= 0 && zi < Di && yi >= 0 && yi < Hi && xi >= 0 && xi < Wi;
        const int64_t vox = in ? ((int64_t)zi * Hi + yi) * Wi + xi : 0;
        const float* wt = wp + (int64_t)tap * CIN * Cout + cg;
#pragma unroll 2
        for (int c = 0; c < CIN; c += 4) {
            f32x4 v;
            load_act4<CIN>(a, b, vox, ld, c, v);
            if (!in) v = f32x4{0, 0, 0, 0};              // zero padding of the *activated* input
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int k = 0; k < CT; ++k) acc[k] = fmaf(v[k4], wt[(int64_t)(c + k4) * Cout + k], acc[k]);
        }
    }
    float* o = out + i * Cout + cg;
#pragma unroll
    for (int k = 0; k < CT; k += 4) *reinterpret_cast<f32x4*>(o + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
}

// LDS-tiled variant for the stride-1 layers (conv0 = 74.5 % of CostRegNet's FLOPs).  A workgroup owns a 4x8x8
// block of output voxels; the (6x10x10)-voxel input halo is staged through LDS in chunks of CK=12 channels with
// fully coalesced loads (a voxel's channels are contiguous; the generic kernel's per-tap gathers at a 176-B lane
// stride touch one cache line per lane).  The pending InPlaceABN of the producer is applied once per staged
// element instead of once per tap.  Voxel stride 12 floats => conflict-free ds_read_b128 across x-neighbours.
template <int CIN, int CT>
__global__ __launch_bounds__(256) void conv3d_k3s1_tiled_kernel(ActSrc a, ActSrc b, int ld, int D, int H, int W,
                                                               const float* __restrict__ wp, int Cout, float* __restrict__ out)
{
    constexpr int CK = 12, IY = 10, IX = 10, NV = 6 * IY * IX;
    __shared__ __attribute__((aligned(16))) float tile[NV * CK];
    const int nbx = (W + 7) / 8, nby = (H + 7) / 8;
    const int bx = blockIdx.x % nbx, by = (blockIdx.x / nbx) % nby, bz = blockIdx.x / (nbx * nby);
    const int cg = blockIdx.y * CT;
    const int tid = threadIdx.x, tx = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
    const int x0 = bx * 8 - 1, y0 = by * 8 - 1, z0 = bz * 4 - 1;
    float acc[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) acc[k] = 0.f;
#pragma unroll 1
    for (int c0 = 0; c0 < CIN; c0 += CK) {
        const int ck4 = (CIN - c0 < CK ? CIN - c0 : CK) / 4;          // float4s per voxel in this chunk
        __syncthreads();
        for (int idx = tid; idx < NV * ck4; idx += 256) {
            const int v = idx / ck4, c4 = idx - v * ck4;
            const int vx = v % IX, vy = (v / IX) % IY, vz = v / (IX * IY);
            const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
            f32x4 val = {0, 0, 0, 0};
            if (gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D)
                load_act4<CIN>(a, b, ((int64_t)gz * H + gy) * W + gx, ld, c0 + c4 * 4, val);
            *reinterpret_cast<f32x4*>(tile + v * CK + c4 * 4) = val;
        }
        __syncthreads();
        for (int tap = 0; tap < 27; ++tap) {
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const float* tv = tile + (((tz + dz) * IY + ty + dy) * IX + tx + dx) * CK;
            const float* wt = wp + ((int64_t)tap * CIN + c0) * Cout + cg;
            for (int c4 = 0; c4 < ck4; ++c4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(tv + c4 * 4);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                    for (int k = 0; k < CT; ++k) acc[k] = fmaf(v[k4], wt[(int64_t)(c4 * 4 + k4) * Cout + k], acc[k]);
            }
        }
    }
    const int ox = bx * 8 + tx, oy = by * 8 + ty, oz = bz * 4 + tz;
    if (ox < W && oy < H && oz < D) {
        float* o = out + (((int64_t)oz * H + oy) * W + ox) * Cout + cg;
#pragma unroll
        for (int k = 0; k < CT; k += 4) *reinterpret_cast<f32x4*>(o + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
    }
}

// Transposed 3x3x3 convolution, stride 2, padding 1, output_padding 1 (models.py:739-752): output size = 2x input.
// out[o] = sum over taps k with o = 2*i - 1 + k.  One thread = one output voxel x CT channels; per dimension an even
// output coordinate has one tap (k=1), an odd one two (k=0 from i=(o+1)/2, k=2 from i=(o-1)/2).
template <int CIN, int CT>
__global__ __launch_bounds__(256) void convT3d_k3s2_kernel(ActSrc a, ActSrc b, int Di, int Hi, int Wi,
                                                          const float* __restrict__ wp, int Cout, float* __restrict__ out)
{
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const int64_t nvox = (int64_t)Do * Ho * Wo;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int cg = blockIdx.y * CT;
    if (i >= nvox) return;
    const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), z = (int)(i / ((int64_t)Wo * Ho));
    float acc[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) acc[k] = 0.f;
    for (int tap = 0; tap < 27; ++tap) {
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        const int tz = z + 1 - kz, ty = y + 1 - ky, tx = x + 1 - kx;      // = 2*i
        if ((tz | ty | tx) & 1) continue;
        if (tz < 0 || ty < 0 || tx < 0) continue;
        const int zi = tz >> 1, yi = ty >> 1, xi = tx >> 1;
        if (zi >= Di || yi >= Hi || xi >= Wi) continue;
        const int64_t vox = ((int64_t)zi * Hi + yi) * Wi + xi;
        const float* wt = wp + (int64_t)tap * CIN * Cout + cg;
        for (int c = 0; c < CIN; c += 4) {
            f32x4 v;
            load_act4<CIN>(a, b, vox, CIN, c, v);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int k = 0; k < CT; ++k) acc[k] = fmaf(v[k4], wt[(int64_t)(c + k4) * Cout + k], acc[k]);
        }
    }
    float* o = out + i * Cout + cg;
#pragma unroll
    for (int k = 0; k < CT; k += 4) *reinterpret_cast<f32x4*>(o + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
}

int g_conv_tiled = 1;   // A/B knob (mvsnerf_tune "conv_tiled")
static bool act_ok(const float* x, const float* sc, const float* sh) { return x && ((sc == nullptr) == (sh == nullptr)) && mvs_aligned16(x); }

extern "C" int mvsnerf_conv3d_fwd(const float* x1, const float* scale1, const float* shift1,
                                  const float* x2, const float* scale2, const float* shift2,
                                  int Cin, int cin_ld, int D, int H, int W, const float* wpacked, int Cout, int stride,
                                  float* out, void* stream)
{
    if (!act_ok(x1, scale1, shift1) || !wpacked || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (x2 && (!act_ok(x2, scale2, shift2) || !scale2)) return MVSNERF_EINVAL;
    if (stride != 1 && stride != 2) return MVSNERF_EUNSUPPORTED;
    if ((cin_ld & 3) || cin_ld < Cin || !mvs_aligned16(out)) return MVSNERF_EALIGN;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;   // k3 p1
    const int64_t nvox = (int64_t)Do * Ho * Wo;
    hipStream_t st = (hipStream_t)stream;
#define MVS_CONV(CIN, CT, S)                                                                          \
    conv3d_k3_kernel<CIN, CT, S><<<dim3(mvs_cdiv(nvox, 256), Cout / CT), 256, 0, st>>>(a, b, cin_ld, D, H, W, wpacked, Cout, out, Do, Ho, Wo)
#define MVS_CONV_TILED(CIN, CT)                                                                       \
    conv3d_k3s1_tiled_kernel<CIN, CT><<<dim3(((W + 7) / 8) * ((H + 7) / 8) * ((D + 3) / 4), Cout / CT), 256, 0, st>>>(a, b, cin_ld, D, H, W, wpacked, Cout, out)
    // (Cin rounded up to a multiple of 4 by the caller's channel padding; Cout in {8,16,32,64})
    const int key = Cin * 1000 + Cout * 10 + stride;
    switch (key) {
        case 44 * 1000 + 8 * 10 + 1:  if (g_conv_tiled) MVS_CONV_TILED(44, 8); else MVS_CONV(44, 8, 1); break;     // conv0 (41 real channels + 3 zero pad)
        case 8 * 1000 + 16 * 10 + 2:  MVS_CONV(8, 16, 2); break;     // conv1
        case 16 * 1000 + 16 * 10 + 1: if (g_conv_tiled) MVS_CONV_TILED(16, 16); else MVS_CONV(16, 16, 1); break;    // conv2
        case 16 * 1000 + 32 * 10 + 2: MVS_CONV(16, 16, 2); break;    // conv3
        case 32 * 1000 + 32 * 10 + 1: if (g_conv_tiled) MVS_CONV_TILED(32, 16); else MVS_CONV(32, 16, 1); break;    // conv4
        case 32 * 1000 + 64 * 10 + 2: MVS_CONV(32, 16, 2); break;    // conv5
        case 64 * 1000 + 64 * 10 + 1: if (g_conv_tiled) MVS_CONV_TILED(64, 16); else MVS_CONV(64, 16, 1); break;    // conv6
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_CONV
#undef MVS_CONV_TILED
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_conv_transpose3d_fwd(const float* x1, const float* scale1, const float* shift1,
                                            const float* x2, const float* scale2, const float* shift2,
                                            int Cin, int D, int H, int W, const float* wpacked, int Cout, float* out, void* stream)
{
    if (!act_ok(x1, scale1, shift1) || !wpacked || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (x2 && (!act_ok(x2, scale2, shift2) || !scale2)) return MVSNERF_EINVAL;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    const int64_t nvox = (int64_t)8 * D * H * W;
    hipStream_t st = (hipStream_t)stream;
#define MVS_CONVT(CIN, CT) convT3d_k3s2_kernel<CIN, CT><<<dim3(mvs_cdiv(nvox, 256), Cout / CT), 256, 0, st>>>(a, b, D, H, W, wpacked, Cout, out)
    switch (Cin * 100 + Cout) {
        case 64 * 100 + 32: MVS_CONVT(64, 16); break;   // conv7
        case 32 * 100 + 16: MVS_CONVT(32, 16); break;   // conv9
        case 16 * 100 + 8:  MVS_CONVT(16, 8); break;    // conv11
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_CONVT
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// InPlaceABN in train mode: batch statistics over all voxels of a channel-last tensor x[n][C].
// Stage 1: per-block partial (sum, sum of squares) per channel, fp32, deterministic order.
// Stage 2: one block combines the partials in fp64 and emits
//     scale = (|w|+eps) / sqrt(var_biased + eps),  shift = b - mean*scale
// and updates running_mean / running_var (momentum, unbiased variance) like F.batch_norm does.
// ---------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(256) void abn_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part)
{
    // thread t handles channel group (t % (C/4)) of voxels t / (C/4) + k*stride  -> float4 loads, fully coalesced
    constexpr int G = C / 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)gridDim.x * blockDim.x;
    const int g = (int)(t % G);
    f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    for (int64_t v = t / G; v < n; v += total / G) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + v * C + g * 4);
        s += a; q += a * a;
    }
    // block reduce: threads with equal g (stride G in threadIdx since 256 % G == 0)
    __shared__ float sh[256 * 8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sh[threadIdx.x * 8 + k] = s[k]; sh[threadIdx.x * 8 + 4 + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < C * 2) {
        const int c = threadIdx.x % C, which = threadIdx.x / C;     // which: 0 sum, 1 sumsq
        const int gg = c / 4, k = c % 4;
        float acc = 0.f;
        for (int j = gg; j < 256; j += G) acc += sh[j * 8 + which * 4 + k];
        part[((int64_t)blockIdx.x * 2 + which) * C + c] = acc;
    }
}

__global__ __launch_bounds__(64) void abn_finalize_kernel(const float* __restrict__ part, int nblocks, int C, int64_t n,
                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                    float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift)
{
    // one wavefront per channel: lanes stride over the per-block partials (fixed order => deterministic), fp64 combine
    const int c = blockIdx.x, lane = threadIdx.x;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nblocks; b += 64) { s += (double)part[((int64_t)b * 2) * C + c]; q += (double)part[((int64_t)b * 2 + 1) * C + c]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
    if (lane != 0) return;
    const double mean = s / (double)n;
    double var = q / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gamma = fabsf(weight[c]) + eps;                    // InPlaceABN: |weight| + eps
    const float sc = gamma * invstd;
    scale[c] = sc;
    shift[c] = bias[c] - (float)mean * sc;
    if (running_mean) {
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        const double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

extern "C" size_t mvsnerf_abn_workspace_floats(int C) { return (size_t)1024 * 2 * C; }

extern "C" int mvsnerf_abn_stats(const float* x, int64_t n_vox, int C, const float* weight, const float* bias,
                                 float* running_mean, float* running_var, float momentum, float eps,
                                 float* scale, float* shift, float* workspace, void* stream)
{
    if (!x || !weight || !bias || !scale || !shift || !workspace || n_vox < 1) return MVSNERF_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x)) return MVSNERF_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)((n_vox * (C / 4) + 255) / 256);
    if (nb > 1024) nb = 1024;
    switch (C) {
        case 8:  abn_partial_kernel<8><<<nb, 256, 0, st>>>(x, n_vox, workspace); break;
        case 16: abn_partial_kernel<16><<<nb, 256, 0, st>>>(x, n_vox, workspace); break;
        case 32: abn_partial_kernel<32><<<nb, 256, 0, st>>>(x, n_vox, workspace); break;
        case 64: abn_partial_kernel<64><<<nb, 256, 0, st>>>(x, n_vox, workspace); break;
        default: return MVSNERF_EUNSUPPORTED;
    }
    MVS_LAUNCH_CHECK();
    abn_finalize_kernel<<<C, 64, 0, st>>>(workspace, nb, C, n_vox, weight, bias, running_mean, running_var, momentum, eps, scale, shift);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// out = leaky(x1*scale1+shift1) [+ leaky(x2*scale2+shift2)]  (materialises an activated tensor, e.g. the final
// 8-channel neural volume  conv0 + conv11(x), models.py:766)
__global__ __launch_bounds__(256) void abn_apply_add_kernel(ActSrc a, ActSrc b, int64_t n4, int C, float* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    f32x4 v = *reinterpret_cast<const f32x4*>(a.x + i * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = act_apply(v[k], a.scale[c + k], a.shift[c + k]);
    if (b.x) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(b.x + i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += act_apply(t[k], b.scale[c + k], b.shift[c + k]);
    }
    *reinterpret_cast<f32x4*>(out + i * 4) = v;
}

extern "C" int mvsnerf_abn_apply_add(const float* x1, const float* scale1, const float* shift1,
                                     const float* x2, const float* scale2, const float* shift2,
                                     int64_t n_vox, int C, float* out, void* stream)
{
    if (!x1 || !scale1 || !shift1 || !out || n_vox < 1 || (C & 3)) return MVSNERF_EINVAL;
    if (x2 && (!scale2 || !shift2)) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x1) || !mvs_aligned16(out) || (x2 && !mvs_aligned16(x2))) return MVSNERF_EALIGN;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    const int64_t n4 = n_vox * C / 4;
    abn_apply_add_kernel<<<mvs_cdiv(n4, 256), 256, 0, (hipStream_t)stream>>>(a, b, n4, C, out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
