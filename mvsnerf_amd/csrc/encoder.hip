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
#include "conv3d_bf16_layout.h"
#include "act.h"
#include "knobs.h"

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

// the same interpolation of N three-channel images written channel-last with a zero fourth channel ([N][Ho][Wo][4]: what the plane sweep reads): one launch
// instead of resize + mvsnerf_nchw_to_nhwc, the same operations per value
__global__ __launch_bounds__(256) void resize_bilinear_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int Hi, int Wi, int Ho, int Wo)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * Ho * Wo) return;
    int x, y, n;
    mvs_unflatten3(i, Wo, Ho, x, y, n);
    const float sh = (float)Hi / (float)Ho, sw = (float)Wi / (float)Wo;     // area_pixel_compute_scale
    float fy = sh * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
    float fx = sw * ((float)x + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < Hi - 1 ? 1 : 0), x1 = x0 + (x0 < Wi - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = src + ((int64_t)n * 3 + c) * Hi * Wi;
        o[c] = hy * (hx * p[(int64_t)y0 * Wi + x0] + lx * p[(int64_t)y0 * Wi + x1]) +
               ly * (hx * p[(int64_t)y1 * Wi + x0] + lx * p[(int64_t)y1 * Wi + x1]);
    }
    *reinterpret_cast<f32x4*>(dst + i * 4) = o;
}

extern "C" int mvsnerf_resize_bilinear_nhwc4(const float* src, float* dst, int N, int Hi, int Wi, int Ho, int Wo, void* stream)
{
    if (!src || !dst || N < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(dst)) return MVSNERF_EALIGN;
    resize_bilinear_nhwc4_kernel<<<mvs_cdiv((int64_t)N * Ho * Wo, 256), 256, 0, (hipStream_t)stream>>>(src, dst, N, Hi, Wi, Ho, Wo);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_resize_bilinear(const float* src, float* dst, int NC, int Hi, int Wi, int Ho, int Wo, void* stream)
{
    if (!src || !dst || NC < 1 || Hi < 1 || Wi < 1 || Ho < 1 || Wo < 1) return MVSNERF_EINVAL;
    resize_bilinear_kernel<<<mvs_cdiv((int64_t)NC * Ho * Wo, 256), 256, 0, (hipStream_t)stream>>>(src, dst, NC, Hi, Wi, Ho, Wo);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// depth hypotheses of MVSNet.forward (models.py:903-906, linear in depth): depth[i] = near * (1 - t[i]) + far * t[i], every operation rounded to fp32 on its own like the
// four ATen kernels it replaces (t = torch.linspace(0, 1, D), cached by the caller; near_far = the two floats on the device: no host synchronisation)
__global__ __launch_bounds__(256) void depth_values_kernel(const float* __restrict__ t, const float* __restrict__ near_far, int D, float* __restrict__ out)
{
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= D) return;
    const float near = near_far[0], far = near_far[1];
    const float a = 1.0f - t[i];
    const float b = near * a;
    const float c = far * t[i];
    out[i] = b + c;
}

extern "C" int mvsnerf_depth_values(const float* t, const float* near_far, int D, float* out, void* stream)
{
    if (!t || !near_far || !out || D < 1) return MVSNERF_EINVAL;
    depth_values_kernel<<<mvs_cdiv(D, 256), 256, 0, (hipStream_t)stream>>>(t, near_far, D, out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// =============================================================================================
// plane sweep forward (homo_warp + build_volume_costvar[_img] in one pass): csrc/planesweep.hip - its own translation unit, compiled
// WITHOUT packed fp32 instructions (see the header there); the guarded encode head below reaches it through mvs_planesweep_launch.
// =============================================================================================
int mvs_planesweep_launch(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                          int V, int C, int H, int W, int D, int pad, float* cost, int CP, float* masks,
                          int with_img, int blocked, void* stream, int* guard, const int* run_if);

// stand-alone homo_warp (utils.py:580-630): one source view, NCHW in, (C,D,Hp,Wp) out + grid (D,Hp*Wp,2)
__global__ __launch_bounds__(256) void homo_warp_kernel(const float* __restrict__ src, const float* __restrict__ P, const float* __restrict__ depth,
                                                        const float* __restrict__ grid_in, int C, int H, int W, int D, int pad,
                                                        float* __restrict__ out, float* __restrict__ grid_out)
{
#pragma clang fp contract(off)
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
        // the CPU reference's chain fma(se, w_se, fma(sw, w_sw, fma(ne, w_ne, nw * w_nw))) with 0 for a tap outside the image
        const float nw = (x0in && y0in) ? pl[(int64_t)y0 * W + x0] : 0.f, ne = (x1in && y0in) ? pl[(int64_t)y0 * W + x0 + 1] : 0.f;
        const float sw = (x0in && y1in) ? pl[(int64_t)(y0 + 1) * W + x0] : 0.f, se = (x1in && y1in) ? pl[(int64_t)(y0 + 1) * W + x0 + 1] : 0.f;
        out[(int64_t)c * nvox + i] = fmaf(se, wx1 * wy1, fmaf(sw, wx0 * wy1, fmaf(ne, wx1 * wy0, nw * (wx0 * wy0))));
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
// weights re-laid as w[tap][ci][co] (co fastest) so that a thread's CT output channels are contiguous and
// wave-uniform => the compiler fetches them through the scalar cache (s_load) and feeds v_fma from SGPRs.
__global__ void conv3d_pack_kernel(const float* __restrict__ w, int ci_real, int co_real, int cin_pad, int cout_pad,
                                   int s_ci, int s_co, int flip, float* __restrict__ packed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = 27 * cin_pad * cout_pad;
    if (i >= total) return;
    const int co = i % cout_pad, ci = (i / cout_pad) % cin_pad;
    int tap = i / (cout_pad * cin_pad);
    if (flip) tap = 26 - tap;                                                    // spatially mirrored kernel (data gradient)
    packed[i] = (ci < ci_real && co < co_real) ? w[(int64_t)ci * s_ci + (int64_t)co * s_co + tap] : 0.f;
}

extern "C" int mvsnerf_conv3d_pack_weights(const float* w, int ci_real, int co_real, int cin_pad, int cout_pad,
                                           int s_ci, int s_co, int flip, float* packed, void* stream)
{
    if (!w || !packed || ci_real < 1 || co_real < 1 || cin_pad < ci_real || cout_pad < co_real) return MVSNERF_EINVAL;
    conv3d_pack_kernel<<<mvs_cdiv(27 * cin_pad * cout_pad, 256), 256, 0, (hipStream_t)stream>>>(w, ci_real, co_real, cin_pad, cout_pad, s_ci, s_co, flip, packed);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// Every re-layout a training step needs (the weights change with every optimizer step: ~50 of them, each a 5-us launch of its own) in ONE
// launch.  A job gathers straight from the nn.Conv / nn.ConvTranspose weight tensor:
//   value(tap, ci, co) = (ci < ci_real && co < co_real) ? w[ci * s_ci + co * s_co + (flip ? ntaps - 1 - tap : tap)] : 0
// into layout kind 0: dst[tap][ci][co] (conv3d_pack_weights / conv2d_pack_weights), 1: dst[ci/4][tap][co][ci%4] (conv3d_pack_weights_c8),
// 2: dst[tap][ci/8][co][ci%8] (conv3d_pack_weights_mfma), 3 / 4: the bf16 B fragments of conv3d_bf16.hip for a convolution / a transposed
// convolution (dst is a __bf16 buffer of mvsnerf_conv3d_bf16_packed_elems(ci_pad, co_pad, kind == 4) elements; 27 taps).
constexpr int MVS_PACK_JOBS = 64;

struct PackJobs {
    const float* w[MVS_PACK_JOBS];
    float* dst[MVS_PACK_JOBS];
    int ci_real[MVS_PACK_JOBS], co_real[MVS_PACK_JOBS], ci_pad[MVS_PACK_JOBS], co_pad[MVS_PACK_JOBS], s_ci[MVS_PACK_JOBS], s_co[MVS_PACK_JOBS];
    short ntaps[MVS_PACK_JOBS];
    signed char flip[MVS_PACK_JOBS], kind[MVS_PACK_JOBS];
    int blk[MVS_PACK_JOBS + 1];
    int n_elems[MVS_PACK_JOBS];
    int n;
};

__global__ __launch_bounds__(256) void pack_weights_multi_kernel(PackJobs J)
{
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= J.blk[j + 1]) ++j;
    const int i = (blockIdx.x - J.blk[j]) * 256 + threadIdx.x;
    const int ntaps = J.ntaps[j], cip = J.ci_pad[j], cop = J.co_pad[j];
    if (J.kind[j] >= 3) {                                         // bf16 fragments (conv3d_bf16.hip)
        if (i >= J.n_elems[j]) return;
        int tap, ci, co;
        float v = 0.f;
        if (mvs_conv3d_bf16_coords(i, cip, cop, J.kind[j] == 4, tap, ci, co, ntaps) && ci < J.ci_real[j] && co < J.co_real[j]) {
            if (J.flip[j]) tap = ntaps - 1 - tap;
            v = J.w[j][(int64_t)ci * J.s_ci[j] + (int64_t)co * J.s_co[j] + tap];
        }
        reinterpret_cast<__bf16*>(J.dst[j])[i] = (__bf16)v;
        return;
    }
    if (i >= ntaps * cip * cop) return;
    int tap, ci, co;
    if (J.kind[j] == 0) { co = i % cop; ci = (i / cop) % cip; tap = i / (cop * cip); }
    else if (J.kind[j] == 1) { co = (i >> 2) % cop; tap = (i / (4 * cop)) % ntaps; ci = (i / (4 * cop * ntaps)) * 4 + (i & 3); }
    else { co = (i >> 3) % cop; ci = ((i / (8 * cop)) % (cip / 8)) * 8 + (i & 7); tap = i / (cip * cop); }
    if (J.flip[j]) tap = ntaps - 1 - tap;
    J.dst[j][i] = (ci < J.ci_real[j] && co < J.co_real[j]) ? J.w[j][(int64_t)ci * J.s_ci[j] + (int64_t)co * J.s_co[j] + tap] : 0.f;
}

// params[j] = {kind, ntaps, ci_real, co_real, ci_pad, co_pad, s_ci, s_co, flip}; host arrays
extern "C" int mvsnerf_pack_weights_multi(int n_jobs, const float* const* w, float* const* dst, const int* params, void* stream)
{
    if (n_jobs < 1 || n_jobs > MVS_PACK_JOBS || !w || !dst || !params) return MVSNERF_EINVAL;
    PackJobs J;
    J.n = n_jobs;
    int b = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const int* q = params + 9 * j;
        if (!w[j] || !dst[j] || q[0] < 0 || q[0] > 4 || q[1] < 1 || q[1] > 27 || q[2] < 1 || q[3] < 1 || q[4] < q[2] || q[5] < q[3]) return MVSNERF_EINVAL;
        size_t n_el = (size_t)q[1] * q[4] * q[5];
        if (q[0] >= 3) {
            n_el = mvs_conv3d_bf16_elems(q[4], q[5], q[0] == 4, q[1]);
            if (n_el == 0) return MVSNERF_EUNSUPPORTED;
        }
        if ((q[0] == 1 && (q[4] & 3)) || (q[0] == 2 && (q[4] & 7))) return MVSNERF_EINVAL;
        J.w[j] = w[j]; J.dst[j] = dst[j];
        J.kind[j] = (signed char)q[0]; J.ntaps[j] = (short)q[1]; J.ci_real[j] = q[2]; J.co_real[j] = q[3]; J.ci_pad[j] = q[4]; J.co_pad[j] = q[5];
        J.s_ci[j] = q[6]; J.s_co[j] = q[7]; J.flip[j] = (signed char)(q[8] ? 1 : 0);
        J.blk[j] = b;
        J.n_elems[j] = (int)n_el;
        b += (int)((n_el + 255) / 256);
    }
    J.blk[n_jobs] = b;
    pack_weights_multi_kernel<<<b, 256, 0, (hipStream_t)stream>>>(J);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// Direct 3x3x3 convolution, padding 1, stride S.  One thread = one output voxel x CT output channels.
// Neighbouring threads re-read each other's input voxels through L1/L2 (27-fold reuse).
template <int CIN, int CT, int S, int COUT>      // COUT a template constant: weight offsets become s_load immediates
__global__ __launch_bounds__(256) void conv3d_k3_kernel(ActSrc a, ActSrc b, int ld, int Di, int Hi, int Wi,
                                                       const float* __restrict__ wp,
                                                       float* __restrict__ out, int Do, int Ho, int Wo, int swz)
{
    const int64_t nvox = (int64_t)Do * Ho * Wo;
    const int64_t i = (int64_t)(swz ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x) * blockDim.x + threadIdx.x;   // an XCD walks contiguous voxels
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
        const float* wt = wp + (int64_t)tap * CIN * COUT + cg;
#pragma unroll 2
        for (int c = 0; c < CIN; c += 4) {
            f32x4 v;
            load_act4<CIN>(a, b, vox, ld, c, v);
            if (!in) v = f32x4{0, 0, 0, 0};              // zero padding of the *activated* input
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int k = 0; k < CT; ++k) acc[k] = fmaf(v[k4], wt[(c + k4) * COUT + k], acc[k]);
        }
    }
    float* o = out + i * COUT + cg;
#pragma unroll
    for (int k = 0; k < CT; k += 4) *reinterpret_cast<f32x4*>(o + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
}

// LDS-tiled variant for the stride-1 layers (conv0 = 74.5 % of CostRegNet's FLOPs).  A workgroup owns a 4x8x8
// block of output voxels; the (6x10x10)-voxel input halo is staged through LDS in chunks of <= 12 channels with
// fully coalesced loads (a voxel's channels are contiguous; the generic kernel's per-tap gathers at a 176-B lane
// stride touch one cache line per lane).  The pending InPlaceABN of the producer is applied once per staged
// element instead of once per tap.  Chunk sizes are compile-time and the x-taps are unrolled so that the LDS reads
// and the scalar weight loads of one (dz,dy) step are in flight while the previous step's FMAs issue.
template <int CIN, int CT, int COUT, int C0, int CKC>     // one channel chunk [C0, C0+CKC) of the tile
__device__ __forceinline__ void conv_tile_chunk(const ActSrc& a, const ActSrc& b, int ld, int D, int H, int W, int x0, int y0, int z0,
                                                const float* __restrict__ wp, int cg, float* __restrict__ tile,
                                                int tid, int tx, int ty, int tz, float (&acc)[CT])
{
    constexpr int CK = 12, IY = 10, IX = 10, NV = 6 * IY * IX, K4 = CKC / 4;
    __syncthreads();
    for (int idx = tid; idx < NV * K4; idx += 256) {
        const int v = idx / K4, c4 = idx - v * K4;
        const int vx = v % IX, vy = (v / IX) % IY, vz = v / (IX * IY);
        const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        f32x4 val = {0, 0, 0, 0};
        if (gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D)
            load_act4<CIN>(a, b, ((int64_t)gz * H + gy) * W + gx, ld, C0 + c4 * 4, val);
        *reinterpret_cast<f32x4*>(tile + v * CK + c4 * 4) = val;
    }
    __syncthreads();
#pragma unroll 1
    for (int dzy = 0; dzy < 9; ++dzy) {
        const int dz = dzy / 3, dy = dzy - dz * 3;
        const float* tv = tile + (((tz + dz) * IY + ty + dy) * IX + tx) * CK;
        const float* wt = wp + ((int64_t)(dzy * 3) * CIN + C0) * COUT + cg;   // COUT is a template constant: the weight offsets below are s_load immediates
        f32x4 v[3][K4];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int c4 = 0; c4 < K4; ++c4) v[dx][c4] = *reinterpret_cast<const f32x4*>(tv + dx * CK + c4 * 4);
        if constexpr (CT <= 16) {
            // Weights in groups of 32 (scalar loads, SGPRs), group g+1 requested before the FMAs of group g and a scheduling
            // fence after each group: left alone the scheduler hoists several groups of s_load_dwordx16, runs out of SGPRs and
            // parks weights in VGPR lanes (v_writelane / v_readlane per weight) - and this kernel is bound by the NUMBER of
            // instructions a SIMD can issue, scalar ones included (PMC: 0.6 scalar instructions per packed FMA cost 27 % of
            // the issue cycles; with Cout a template constant the weight offsets are s_load immediates: conv0 2.0 -> 1.5 ms).
            // Tried and dropped: groups of 16 with the requests for the next pair issued after the first FMAs of a pair (so that
            // the compiler's lgkmcnt(0) only covers loads that are ~50 cycles old): CostRegNet 3.03 -> 3.39 ms; 8 instead of 12
            // channels staged per pass (8 instead of 5 workgroups per CU): -1 %, 4 channels: +6 %.
            constexpr int KPG = CT <= 8 ? 4 : 2;                 // k4 values per weight group
            constexpr int GPC = 4 / KPG, NG = 3 * K4 * GPC;       // groups per float4 of x, groups per row
            float w[2][KPG * CT];
            auto wload = [&](float (&wd)[KPG * CT], int g) {
                const int dx = g / (K4 * GPC), r = g - dx * (K4 * GPC), c4 = r / GPC, kb = (r - c4 * GPC) * KPG;
#pragma unroll
                for (int j = 0; j < KPG; ++j)
#pragma unroll
                    for (int k = 0; k < CT; ++k) wd[j * CT + k] = wt[(dx * CIN + c4 * 4 + kb + j) * COUT + k];
            };
            wload(w[0], 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) wload(w[(g + 1) & 1], g + 1);
                const int dx = g / (K4 * GPC), r = g - dx * (K4 * GPC), c4 = r / GPC, kb = (r - c4 * GPC) * KPG;
#pragma unroll
                for (int j = 0; j < KPG; ++j)
#pragma unroll
                    for (int k = 0; k < CT; ++k) acc[k] = fmaf(v[dx][c4][kb + j], w[g & 1][j * CT + k], acc[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int c4 = 0; c4 < K4; ++c4)
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                        for (int k = 0; k < CT; ++k)
                            acc[k] = fmaf(v[dx][c4][k4], wt[(dx * CIN + c4 * 4 + k4) * COUT + k], acc[k]);
        }
    }
}

// Tiles are renumbered so that each XCD walks a contiguous range of them (halo neighbours then share an L2: fabric fetch of conv0
// 6.4 -> 4.4 GB, layer -5 %).  Measured and dropped (conv0, 2.19 ms baseline): two output voxels per thread (half the scalar-load
// traffic and 2/3 of the LDS reads per FMA) -2.5 %; the same with 8-channel chunks -4 %; next chunk's halo prefetched into
// registers during the FMAs (software pipeline, 128 VGPRs) +9 %.  PMC of the kernel as it is: waves spend 62 % of their cycles in
// s_waitcnt, VALU issues 45 % of the time, scalar-cache miss rate 1.5 %, LDS busy 24 % (63 % of that bank conflicts).
template <int CIN, int CT, int COUT>
__global__ __launch_bounds__(256) void conv3d_k3s1_tiled_kernel(ActSrc a, ActSrc b, int ld, int D, int H, int W,
                                                               const float* __restrict__ wp, float* __restrict__ out, int swz,
                                                               float* __restrict__ stats = nullptr, const int* __restrict__ run_if = nullptr)
{
    if (run_if && *run_if == 0) return;                          // fp32 half of a guarded sequence (mvsnerf_conv3d_f16x3_guarded_fwd)
    __shared__ __attribute__((aligned(16))) float tile[600 * 12];
    const int nbx = (W + 7) / 8, nby = (H + 7) / 8;
    const int tile_id = swz ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;   // halo neighbours share an XCD's L2
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, bz = tile_id / (nbx * nby);
    const int cg = blockIdx.y * CT;
    const int tid = threadIdx.x, tx = tid & 7, ty = (tid >> 3) & 7, tz = tid >> 6;
    const int x0 = bx * 8 - 1, y0 = by * 8 - 1, z0 = bz * 4 - 1;
    float acc[CT];
#pragma unroll
    for (int k = 0; k < CT; ++k) acc[k] = 0.f;
#define MVS_CHUNK(C0_, CKC_) conv_tile_chunk<CIN, CT, COUT, C0_, CKC_>(a, b, ld, D, H, W, x0, y0, z0, wp, cg, tile, tid, tx, ty, tz, acc)
    if constexpr (CIN >= 12) MVS_CHUNK(0, 12); else MVS_CHUNK(0, CIN);
    if constexpr (CIN >= 24) MVS_CHUNK(12, 12); else if constexpr (CIN > 12) MVS_CHUNK(12, CIN - 12);
    if constexpr (CIN >= 36) MVS_CHUNK(24, 12); else if constexpr (CIN > 24) MVS_CHUNK(24, CIN - 24);
    if constexpr (CIN >= 48) MVS_CHUNK(36, 12); else if constexpr (CIN > 36) MVS_CHUNK(36, CIN - 36);
    if constexpr (CIN >= 60) MVS_CHUNK(48, 12); else if constexpr (CIN > 48) MVS_CHUNK(48, CIN - 48);
    if constexpr (CIN > 60) MVS_CHUNK(60, CIN - 60);
#undef MVS_CHUNK
    const int ox = bx * 8 + tx, oy = by * 8 + ty, oz = bz * 4 + tz;
    const bool inside = ox < W && oy < H && oz < D;
    if (inside) {
        float* o = out + (((int64_t)oz * H + oy) * W + ox) * COUT + cg;
#pragma unroll
        for (int k = 0; k < CT; k += 4) *reinterpret_cast<f32x4*>(o + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
    }
    if constexpr (256 % CT == 0 && CT >= 4 && CT * 256 <= 600 * 12) {     // the channel values must fit the input tile they reuse
        if (stats) {
            // InPlaceABN partial sums of this tile's CT channels (conv2: no statistics pass re-reads the output).  The 256 voxel values of a
            // channel go through the (now free) input tile: channel k gets 256 / CT lanes of one wave, each adds CT values and their squares,
            // a shuffle tree finishes; fixed order.  Slot = tile, gridDim.x slots (abn_part_at).
            constexpr int LPC = 256 / CT;                        // lanes per channel (8 .. 64: inside a wave)
            __syncthreads();
#pragma unroll
            for (int k = 0; k < CT; ++k) tile[k * 256 + tid] = inside ? acc[k] : 0.0f;
            __syncthreads();
            const int k = tid / LPC, part = tid % LPC;
            float ssum = 0.f, ssq = 0.f;
#pragma unroll
            for (int i = 0; i < CT; ++i) { const float v = tile[k * 256 + part * CT + i]; ssum += v; ssq = fmaf(v, v, ssq); }
#pragma unroll
            for (int o = 1; o < LPC; o <<= 1) { ssum += __shfl_xor(ssum, o); ssq += __shfl_xor(ssq, o); }
            if (part == 0) {
                stats[abn_part_at(0, cg + k, COUT, tile_id, gridDim.x)] = ssum;
                stats[abn_part_at(1, cg + k, COUT, tile_id, gridDim.x)] = ssq;
            }
        }
    }
}

// Transposed 3x3x3 convolution, stride 2, padding 1, output_padding 1 (models.py:739-752): output size = 2x input.
// out[o] = sum over taps k with o = 2*i - 1 + k.  One thread = one output voxel x CT channels; per dimension an even
// output coordinate has one tap (k=1), an odd one two (k=0 from i=(o+1)/2, k=2 from i=(o-1)/2).  The tap loop is wave-uniform
// (only its predicate is per lane), so weights stay on the scalar path; a wave spans one (z,y) row, i.e. it skips the
// (kz,ky) combinations of the wrong parity as a whole.  Measured alternatives, both slower: one workgroup per output parity
// class with strided stores (conv11 436 -> ~600 us), and one thread per x pair with three weight sets per channel chunk.
template <int CIN, int CT, int COUT>
__global__ __launch_bounds__(256) void convT3d_k3s2_kernel(ActSrc a, ActSrc b, int Di, int Hi, int Wi,
                                                          const float* __restrict__ wp, float* __restrict__ out, int swz)
{
    const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
    const int64_t nvox = (int64_t)Do * Ho * Wo;
    const int64_t i = (int64_t)(swz ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x) * blockDim.x + threadIdx.x;
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
        const float* wt = wp + (int64_t)tap * CIN * COUT + cg;
#pragma unroll 1                                            // unrolled, a tap's CIN*CT weights overflow the SGPR file into VGPR lanes
        for (int c = 0; c < CIN; c += 4) {
            f32x4 v;
            load_act4<CIN>(a, b, vox, CIN, c, v);
#pragma unroll
            for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
                for (int k = 0; k < CT; ++k) acc[k] = fmaf(v[k4], wt[(c + k4) * COUT + k], acc[k]);
        }
    }
    float* o = out + i * COUT + cg;
#pragma unroll
    for (int k = 0; k < CT; k += 4) *reinterpret_cast<f32x4*>(o + k) = f32x4{acc[k], acc[k + 1], acc[k + 2], acc[k + 3]};
}

int mvs_conv3d_c8_mfma(const ActSrc& a, const ActSrc& b, int Cin, int cin_real, int cin_ld, int D, int H, int W, const float* wpacked, float* out,
                       int xcd, hipStream_t st);
int mvs_conv3d_mfma32(const ActSrc& a, const ActSrc& b, int Cin, int cin_ld, int D, int H, int W, const float* w32, int Cout, int stride,
                      float* out, float* stats, hipStream_t st, const int* run_if = nullptr);
int mvs_conv3d_mfma32_tiles(int D, int H, int W, int stride);
bool mvs_conv3d_mfma32_supported(int Cin, int Cout, int stride);
int mvs_conv_w32_repack(const float* wpacked, float* w32, int Cin, int Cout, hipStream_t st);
int mvs_convT3d_mfma32(const float* x, int Cin, int D, int H, int W, const float* w32, int Cout, float* out, float* stats, hipStream_t st);
int mvs_convT3d_mfma32_tiles(int Cin, int Cout, int D, int H, int W);
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
#define MVS_CONV(CIN, CT, S, COUT)                                                                    \
    conv3d_k3_kernel<CIN, CT, S, COUT><<<dim3(mvs_cdiv(nvox, 256), COUT / CT), 256, 0, st>>>(a, b, cin_ld, D, H, W, wpacked, out, Do, Ho, Wo, g_conv_xcd)
#define MVS_CONV_TILED(CIN, CT, COUT)                                                                 \
    conv3d_k3s1_tiled_kernel<CIN, CT, COUT><<<dim3(((W + 7) / 8) * ((H + 7) / 8) * ((D + 3) / 4), COUT / CT), 256, 0, st>>>(a, b, cin_ld, D, H, W, wpacked, out, g_conv_xcd)
    if (Cout == 8 && stride == 1 && g_conv_mfma) {
        const int rc = mvs_conv3d_c8_mfma(a, b, Cin, Cin, cin_ld, D, H, W, wpacked, out, g_conv_xcd, st);
        if (rc != MVSNERF_EUNSUPPORTED) return rc;
    }
    // (Cin rounded up to a multiple of 4 by the caller's channel padding; Cout in {8,16,32,64})
    const int key = Cin * 1000 + Cout * 10 + stride;
    switch (key) {
        case 44 * 1000 + 8 * 10 + 1:  if (g_conv_tiled) MVS_CONV_TILED(44, 8, 8); else MVS_CONV(44, 8, 1, 8); break;     // conv0 (41 real channels + 3 zero pad)
        case 8 * 1000 + 44 * 10 + 1:  MVS_CONV_TILED(8, 44, 44); break;  // data gradient of conv0
        // conv0 for other source-view counts: Cin = 32 + 3V rounded up to 4 (V=0: plain variance volume; V=1,2,5,6,7,8)
        case 32 * 1000 + 8 * 10 + 1:  MVS_CONV_TILED(32, 8, 8); break;   case 8 * 1000 + 32 * 10 + 1:  MVS_CONV_TILED(8, 32, 32); break;
        case 36 * 1000 + 8 * 10 + 1:  MVS_CONV_TILED(36, 8, 8); break;   case 8 * 1000 + 36 * 10 + 1:  MVS_CONV_TILED(8, 36, 36); break;
        case 40 * 1000 + 8 * 10 + 1:  MVS_CONV_TILED(40, 8, 8); break;   case 8 * 1000 + 40 * 10 + 1:  MVS_CONV_TILED(8, 40, 40); break;
        case 48 * 1000 + 8 * 10 + 1:  MVS_CONV_TILED(48, 8, 8); break;   case 8 * 1000 + 48 * 10 + 1:  MVS_CONV_TILED(8, 48, 48); break;
        case 52 * 1000 + 8 * 10 + 1:  MVS_CONV_TILED(52, 8, 8); break;   case 8 * 1000 + 52 * 10 + 1:  MVS_CONV_TILED(8, 52, 52); break;
        case 56 * 1000 + 8 * 10 + 1:  MVS_CONV_TILED(56, 8, 8); break;   case 8 * 1000 + 56 * 10 + 1:  MVS_CONV_TILED(8, 56, 56); break;
        case 8 * 1000 + 16 * 10 + 2:  MVS_CONV(8, 16, 2, 16); break;     // conv1
        case 16 * 1000 + 16 * 10 + 1: if (g_conv_tiled) MVS_CONV_TILED(16, 16, 16); else MVS_CONV(16, 16, 1, 16); break;    // conv2
        case 16 * 1000 + 32 * 10 + 2: MVS_CONV(16, 16, 2, 32); break;    // conv3
        case 32 * 1000 + 32 * 10 + 1: if (g_conv_tiled) MVS_CONV_TILED(32, 16, 32); else MVS_CONV(32, 16, 1, 32); break;    // conv4
        case 32 * 1000 + 64 * 10 + 2: MVS_CONV(32, 16, 2, 64); break;    // conv5
        case 64 * 1000 + 64 * 10 + 1: if (g_conv_tiled) MVS_CONV_TILED(64, 16, 64); else MVS_CONV(64, 16, 1, 64); break;    // conv6
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_CONV
#undef MVS_CONV_TILED
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// conv2 (16 -> 16, stride 1: the LDS-tiled VALU kernel) leaving the InPlaceABN partial sums of its raw output from the same launch:
// stats_part[2][16][mvsnerf_conv3d_tiled_tiles(D, H, W)] for mvsnerf_abn_finalize.  Other shapes: MVSNERF_EUNSUPPORTED (mvsnerf_conv3d_fwd
// + mvsnerf_abn_stats, or the matrix-core entries with their own statistics).
extern "C" int mvsnerf_conv3d_tiled_tiles(int D, int H, int W) { return (D < 1 || H < 1 || W < 1) ? 0 : ((W + 7) / 8) * ((H + 7) / 8) * ((D + 3) / 4); }

extern "C" int mvsnerf_conv3d_fwd_stats(const float* x1, const float* scale1, const float* shift1, int Cin, int cin_ld, int D, int H, int W,
                                        const float* wpacked, int Cout, int stride, float* out, float* stats_part, void* stream)
{
    if (!act_ok(x1, scale1, shift1) || !wpacked || !out || !stats_part || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if ((cin_ld & 3) || cin_ld < Cin || !mvs_aligned16(out)) return MVSNERF_EALIGN;
    if (Cin != 16 || Cout != 16 || stride != 1 || !g_conv_tiled) return MVSNERF_EUNSUPPORTED;
    const ActSrc a{x1, scale1, shift1}, b{nullptr, nullptr, nullptr};
    conv3d_k3s1_tiled_kernel<16, 16, 16><<<dim3((unsigned)mvsnerf_conv3d_tiled_tiles(D, H, W), 1), 256, 0, (hipStream_t)stream>>>(a, b, cin_ld, D, H, W, wpacked, out,
                                                                                                                         g_conv_xcd, stats_part);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

int mvs_conv3d_c8_mfma4(const float* x4, int Cin, int cin_real, int D, int H, int W, const float* wq, float* out, int xcd, float* stats, hipStream_t st,
                        const int* run_if = nullptr);
int mvs_conv3d_c8_mfma4_tiles(int D, int H, int W);
int mvs_conv_w4_repack(const float* wpacked, float* wq, int Cin, hipStream_t st);

extern "C" int mvsnerf_conv3d_pack_weights_c8(const float* wpacked, int Cin, float* wq, void* stream)
{
    if (!wpacked || !wq || Cin < 4 || (Cin & 3)) return MVSNERF_EINVAL;
    return mvs_conv_w4_repack(wpacked, wq, Cin, (hipStream_t)stream);
}

extern "C" int mvsnerf_conv3d_c8_blocked_fwd(const float* x_blocked, int Cin, int Cin_real, int D, int H, int W, const float* wq, float* out, void* stream)
{
    if (!x_blocked || !wq || !out || D < 1 || H < 1 || W < 1 || Cin < 4 || (Cin & 3) || Cin_real < 1 || Cin_real > Cin) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x_blocked) || !mvs_aligned16(out) || !mvs_aligned16(wq)) return MVSNERF_EALIGN;
    return mvs_conv3d_c8_mfma4(x_blocked, Cin, Cin_real, D, H, W, wq, out, g_conv_xcd, nullptr, (hipStream_t)stream);
}

// The same convolution leaving the InPlaceABN statistics of its output as per-workgroup partial sums (stats_part: 2 * 8 floats per
// workgroup, mvsnerf_conv3d_c8_blocked_tiles(D, H, W) workgroups; finish with mvsnerf_abn_finalize): the 150 MB output is not read again.
extern "C" int mvsnerf_conv3d_c8_blocked_tiles(int D, int H, int W) { return mvs_conv3d_c8_mfma4_tiles(D, H, W); }

extern "C" int mvsnerf_conv3d_c8_blocked_fwd_stats(const float* x_blocked, int Cin, int Cin_real, int D, int H, int W, const float* wq, float* out,
                                                   float* stats_part, void* stream)
{
    if (!x_blocked || !wq || !out || !stats_part || D < 1 || H < 1 || W < 1 || Cin < 4 || (Cin & 3) || Cin_real < 1 || Cin_real > Cin) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x_blocked) || !mvs_aligned16(out) || !mvs_aligned16(wq)) return MVSNERF_EALIGN;
    return mvs_conv3d_c8_mfma4(x_blocked, Cin, Cin_real, D, H, W, wq, out, g_conv_xcd, stats_part, (hipStream_t)stream);
}

// ---- the deep layers (32 / 64 output channels) on v_mfma_f32_32x32x2_f32 (conv_mfma.hip)
extern "C" int mvsnerf_conv3d_mfma_supported(int Cin, int Cout, int stride) { return (g_conv_mfma && mvs_conv3d_mfma32_supported(Cin, Cout, stride)) ? 1 : 0; }

extern "C" int mvsnerf_conv3d_pack_weights_mfma(const float* wpacked, int Cin, int Cout, float* w32, void* stream)
{
    if (!wpacked || !w32 || Cin < 8 || (Cin & 7) || Cout < 8 || (Cout & 7)) return MVSNERF_EINVAL;
    return mvs_conv_w32_repack(wpacked, w32, Cin, Cout, (hipStream_t)stream);
}

extern "C" int mvsnerf_conv3d_mfma_tiles(int D, int H, int W, int stride) { return (stride == 1 || stride == 2) ? mvs_conv3d_mfma32_tiles(D, H, W, stride) : 0; }

extern "C" int mvsnerf_conv3d_mfma_fwd(const float* x1, const float* scale1, const float* shift1, int Cin, int cin_ld, int D, int H, int W,
                                       const float* w32, int Cout, int stride, float* out, float* stats_part, void* stream)
{
    if (!act_ok(x1, scale1, shift1) || !w32 || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (stride != 1 && stride != 2) return MVSNERF_EUNSUPPORTED;
    if ((cin_ld & 3) || cin_ld < Cin || !mvs_aligned16(out) || !mvs_aligned16(w32)) return MVSNERF_EALIGN;
    const ActSrc a{x1, scale1, shift1}, b{nullptr, nullptr, nullptr};
    return mvs_conv3d_mfma32(a, b, Cin, cin_ld, D, H, W, w32, Cout, stride, out, stats_part, (hipStream_t)stream);
}

extern "C" int mvsnerf_conv_transpose3d_mfma_supported(int Cin, int Cout)
{
    const int k = Cin * 100 + Cout;
    return (g_conv_mfma && (k == 6432 || k == 3216 || k == 1608)) ? 1 : 0;
}

extern "C" int mvsnerf_conv_transpose3d_mfma_fwd(const float* x, int Cin, int D, int H, int W, const float* w32, int Cout, float* out, void* stream)
{
    if (!x || !w32 || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x) || !mvs_aligned16(w32) || !mvs_aligned16(out)) return MVSNERF_EALIGN;
    return mvs_convT3d_mfma32(x, Cin, D, H, W, w32, Cout, out, nullptr, (hipStream_t)stream);
}

// the same launch + the InPlaceABN partial sums of the raw output (conv7 / conv9: no statistics pass re-reads it):
// stats_part[2][Cout][mvsnerf_conv_transpose3d_mfma_tiles(Cin, Cout, D, H, W)] for mvsnerf_abn_finalize
extern "C" int mvsnerf_conv_transpose3d_mfma_tiles(int Cin, int Cout, int D, int H, int W)
{
    if (!mvsnerf_conv_transpose3d_mfma_supported(Cin, Cout) || D < 1 || H < 1 || W < 1) return 0;
    return mvs_convT3d_mfma32_tiles(Cin, Cout, D, H, W);
}

extern "C" int mvsnerf_conv_transpose3d_mfma_fwd_stats(const float* x, int Cin, int D, int H, int W, const float* w32, int Cout, float* out,
                                                       float* stats_part, void* stream)
{
    if (!x || !w32 || !out || !stats_part || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x) || !mvs_aligned16(w32)) return MVSNERF_EALIGN;
    if (!mvsnerf_conv_transpose3d_mfma_supported(Cin, Cout)) return MVSNERF_EUNSUPPORTED;
    return mvs_convT3d_mfma32(x, Cin, D, H, W, w32, Cout, out, stats_part, (hipStream_t)stream);
}

int mvs_convT3d_c16to8_mfma4(const ActSrc& xa, const ActSrc& xb, int D, int H, int W, const float* wq, float* out, int xcd, float* stats, hipStream_t st);
int mvs_convT3d_c16to8_tiles(int D, int H, int W);

// ConvTranspose3d(16, 8, 3, stride 2, padding 1, output_padding 1) of a plain tensor x[D][H][W][16] -> out[2D][2H][2W][8] on
// v_mfma_f32_4x4x1 without padded products; wq: the layer's weights as [ci/4][tap][co][4].  1 when the "conv_mfma" switch is on.
extern "C" int mvsnerf_conv_transpose3d_c8_supported(int Cin, int Cout) { return (g_conv_mfma && Cin == 16 && Cout == 8) ? 1 : 0; }

extern "C" int mvsnerf_conv_transpose3d_c8_fwd(const float* x1, const float* scale1, const float* shift1,
                                               const float* x2, const float* scale2, const float* shift2,
                                               int Cin, int D, int H, int W, const float* wq, float* out, float* stats_part, void* stream)
{
    if (!act_ok(x1, scale1, shift1) || !wq || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (x2 && (!act_ok(x2, scale2, shift2) || !scale2)) return MVSNERF_EINVAL;
    if (Cin != 16) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(wq) || !mvs_aligned16(out)) return MVSNERF_EALIGN;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    return mvs_convT3d_c16to8_mfma4(a, b, D, H, W, wq, out, g_conv_xcd, stats_part, (hipStream_t)stream);
}

extern "C" int mvsnerf_conv_transpose3d_c8_tiles(int D, int H, int W) { return mvs_convT3d_c16to8_tiles(D, H, W); }

extern "C" int mvsnerf_conv_transpose3d_fwd(const float* x1, const float* scale1, const float* shift1,
                                            const float* x2, const float* scale2, const float* shift2,
                                            int Cin, int D, int H, int W, const float* wpacked, int Cout, float* out, void* stream)
{
    if (!act_ok(x1, scale1, shift1) || !wpacked || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (x2 && (!act_ok(x2, scale2, shift2) || !scale2)) return MVSNERF_EINVAL;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    const int64_t nvox = (int64_t)8 * D * H * W;
    hipStream_t st = (hipStream_t)stream;
#define MVS_CONVT(CIN, CT, COUT) convT3d_k3s2_kernel<CIN, CT, COUT><<<dim3(mvs_cdiv(nvox, 256), COUT / CT), 256, 0, st>>>(a, b, D, H, W, wpacked, out, g_conv_xcd)
    switch (Cin * 100 + Cout) {
        case 64 * 100 + 32: MVS_CONVT(64, 16, 32); break;   // conv7
        case 32 * 100 + 16: MVS_CONVT(32, 16, 16); break;   // conv9
        case 16 * 100 + 8:  MVS_CONVT(16, 8, 8); break;    // conv11
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
__global__ __launch_bounds__(256) void abn_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part, const int* __restrict__ run_if = nullptr)
{
    if (run_if && *run_if == 0) return;                          // fp32 half of a guarded sequence (mvsnerf_sweep_conv0_guarded_fwd)
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
        part[abn_part_at(which, c, C, blockIdx.x, gridDim.x)] = acc;
    }
}

__global__ __launch_bounds__(1024) void abn_finalize_kernel(const float* __restrict__ part, int nblocks, int C, int64_t n,
                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                    float* __restrict__ running_mean, float* __restrict__ running_var,
                                    float momentum, float eps, float* __restrict__ scale, float* __restrict__ shift,
                                    float* __restrict__ mean_out, float* __restrict__ invstd_out)
{
    // one workgroup of 1024 threads per channel: thread t adds the partials of slots t, t+1024, ... - two contiguous runs of the
    // channel-major partial array (abn_part_at), four independent loads in flight per run (a full-resolution layer has ~9 k slots: three
    // rounds; 256 threads needed nine, each a round trip to L2) - then a shuffle tree inside the wave and the 16 wave sums in a fixed
    // order (deterministic), fp64 throughout
    const int c = blockIdx.x, t = threadIdx.x;
    const float* ps = part + abn_part_at(0, c, C, 0, nblocks);
    const float* pq = part + abn_part_at(1, c, C, 0, nblocks);
    double s = 0.0, q = 0.0;
    int b = t;
    for (; b + 3072 < nblocks; b += 4096) {
        const float s0 = ps[b], s1 = ps[b + 1024], s2 = ps[b + 2048], s3 = ps[b + 3072];
        const float q0 = pq[b], q1 = pq[b + 1024], q2 = pq[b + 2048], q3 = pq[b + 3072];
        s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
        q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
    }
    {   // the (up to three) remaining rounds: all loads first
        float sv[3] = {0.f, 0.f, 0.f}, qv[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (b + k * 1024 < nblocks) { sv[k] = ps[b + k * 1024]; qv[k] = pq[b + k * 1024]; }
        s += ((double)sv[0] + (double)sv[1]) + (double)sv[2];
        q += ((double)qv[0] + (double)qv[1]) + (double)qv[2];
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
    __shared__ double rs[16], rq[16];
    if ((t & 63) == 0) { rs[t >> 6] = s; rq[t >> 6] = q; }
    __syncthreads();
    if (t != 0) return;
    s = 0.0; q = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { s += rs[w]; q += rq[w]; }
    const double mean = s / (double)n;
    double var = q / (double)n - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float gamma = fabsf(weight[c]) + eps;                    // InPlaceABN: |weight| + eps
    const float sc = gamma * invstd;
    scale[c] = sc;
    shift[c] = bias[c] - (float)mean * sc;
    if (mean_out) { mean_out[c] = (float)mean; invstd_out[c] = invstd; }
    if (running_mean) {
        running_mean[c] = (1.0f - momentum) * running_mean[c] + momentum * (float)mean;
        const double unb = n > 1 ? var * (double)n / (double)(n - 1) : var;
        running_var[c] = (1.0f - momentum) * running_var[c] + momentum * (float)unb;
    }
}

// (Measured and dropped: folding this kernel into abn_partial_kernel through a "last workgroup finalizes" ticket.  With an agent-scope
// release fence per workgroup the 5 us partial kernel took 60 us (one L2 write-back per workgroup); with write-through partial stores and
// a parallel read by the last workgroup 20 us - the last workgroup reads nb x 2C partials that no cache holds, alone, after everybody
// else has left.  Two small launches, 5 + 5.5 us, are faster.)
extern "C" size_t mvsnerf_abn_workspace_floats(int C) { return (size_t)1024 * 2 * C + 2 * C; }

extern "C" int mvsnerf_abn_stats(const float* x, int64_t n_vox, int C, const float* weight, const float* bias,
                                 float* running_mean, float* running_var, float momentum, float eps,
                                 float* scale, float* shift, float* mean_out, float* invstd_out, float* workspace, void* stream)
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
    abn_finalize_kernel<<<C, 1024, 0, st>>>(workspace, nb, C, n_vox, weight, bias, running_mean, running_var, momentum, eps, scale, shift, mean_out, invstd_out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// stage 2 alone, for producers that leave the per-workgroup sums themselves (part[({sum, sum of squares} * C + c) * n_blocks + b], b < n_blocks)
extern "C" int mvsnerf_abn_finalize(const float* part, int n_blocks, int C, int64_t n_vox, const float* weight, const float* bias,
                                    float* running_mean, float* running_var, float momentum, float eps,
                                    float* scale, float* shift, float* mean_out, float* invstd_out, void* stream)
{
    if (!part || n_blocks < 1 || C < 1 || n_vox < 1 || !weight || !bias || !scale || !shift) return MVSNERF_EINVAL;
    if ((running_mean == nullptr) != (running_var == nullptr)) return MVSNERF_EINVAL;
    abn_finalize_kernel<<<C, 1024, 0, (hipStream_t)stream>>>(part, n_blocks, C, n_vox, weight, bias, running_mean, running_var, momentum, eps, scale, shift,
                                                            mean_out, invstd_out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---- guarded scene-encode head (include/mvsnerf_hip.h "guarded 16-bit sequences"): models.py:839-893 + conv0 of :756
int mvs_conv0_f16x3_fwd(const void* x16, int Cin, int D, int H, int W, const void* packed, float* out, float* stats_part, int* guard, hipStream_t st);   // conv_f16x3.hip

__global__ void guard_consume_kernel(int* guard)
{
    if (guard[0]) { guard[1] += 1; guard[0] = 0; }
}

// for entries outside this file (raymarch.hip)
int mvs_guard_consume(int* guard, hipStream_t st)
{
    guard_consume_kernel<<<1, 1, 0, st>>>(guard);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_sweep_conv0_guarded_fwd(const mvsnerf_sweep_conv0_args* a, void* stream)
{
    if (!a || !a->guard || !a->cost16x2 || !a->cost32 || !a->w_f16x3 || !a->w_c8 || !a->out || !a->imgs_cl) return MVSNERF_EINVAL;
    if (a->Cin != 3 * a->V + 32 || a->CP != ((a->Cin + 3) & ~3)) return MVSNERF_EINVAL;
    if (!mvs_aligned16(a->cost32) || !mvs_aligned16(a->out) || !mvs_aligned16(a->w_c8)) return MVSNERF_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    const int Hp = a->H + 2 * a->pad, Wp = a->W + 2 * a->pad;
    int rc;
    // 1. the fp16 pair; both report through guard[0]
    if ((rc = mvs_planesweep_launch(a->feats_cl, a->imgs_cl, a->proj, a->depth, a->V, 32, a->H, a->W, a->D, a->pad, reinterpret_cast<float*>(a->cost16x2), a->CP,
                                a->masks, 1, 3, stream, a->guard, nullptr))) return rc;
    if ((rc = mvs_conv0_f16x3_fwd(a->cost16x2, a->Cin, a->D, Hp, Wp, a->w_f16x3, a->out, a->stats_part, a->guard, st))) return rc;
    // 2. the fp32 pair, predicated on guard[0]: same masks, same output, the statistics in the fp16 kernel's slots
    if ((rc = mvs_planesweep_launch(a->feats_cl, a->imgs_cl, a->proj, a->depth, a->V, 32, a->H, a->W, a->D, a->pad, a->cost32, a->CP,
                                a->masks, 1, 1, stream, nullptr, a->guard))) return rc;
    if ((rc = mvs_conv3d_c8_mfma4(a->cost32, a->CP, a->Cin, a->D, Hp, Wp, a->w_c8, a->out, g_conv_xcd, nullptr, st, a->guard))) return rc;
    if (a->stats_part) {
        const int slots = ((Wp + 15) / 16) * ((Hp + 7) / 8) * ((a->D + 3) / 4);           // mvsnerf_conv0_bf16_tiles(D, Hp, Wp)
        abn_partial_kernel<8><<<slots, 256, 0, st>>>(a->out, (int64_t)a->D * Hp * Wp, a->stats_part, a->guard);
        MVS_LAUNCH_CHECK();
    }
    // 3. count the event, re-arm
    return mvs_guard_consume(a->guard, st);
}

// conv1 / conv2 of a no-grad encode as a guarded sequence (include/mvsnerf_hip.h): the fp16x3 LDS-tiled kernel (conv_f16x3_tiled.hip) reporting through guard[0],
// then the layer's fp32 kernel and a statistics pass over its output, both predicated on that word (they cost a launch when it is clear, recompute the
// layer - the same out, the same stats_part slots - when it is set).  consume != 0: count the event in guard[1] and re-arm guard[0] (the LAST guarded
// layer of the encode passes 1; earlier ones may pass 0: a set guard then also makes the later layers take their fp32 kernels, which is harmless).
int mvs_conv3d_f16x3_tiled_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int D, int H, int W, const void* wq, int Cout,
                               int stride, float* out, float* stats_part, int* guard, hipStream_t st);                     // conv_f16x3_tiled.hip
extern "C" int mvsnerf_conv3d_f16x3_slots(void);

extern "C" int mvsnerf_conv3d_f16x3_guarded_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int D, int H, int W,
                                                const void* w_f16x3, const float* w_f32, int Cout, int stride, float* out, float* stats_part,
                                                int* guard, int consume, void* stream)
{
    if (!act_ok(x, scale, shift) || !w_f16x3 || !w_f32 || !out || !guard || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (Cout != 16 || !((Cin == 8 && stride == 2) || (Cin == 16 && stride == 1))) return MVSNERF_EUNSUPPORTED;
    if ((cin_ld & 3) || cin_ld < Cin || !mvs_aligned16(out)) return MVSNERF_EALIGN;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    // 1. fp16x3; the guard is set from the (NaN) partial sums, so the statistics are always computed
    if ((rc = mvs_conv3d_f16x3_tiled_fwd(x, scale, shift, Cin, cin_ld, D, H, W, w_f16x3, Cout, stride, out, stats_part, guard, st))) return rc;
    // 2. the fp32 kernel of the layer, predicated: conv1 = conv3d_k3_mfma16_kernel<8, 2> (w_f32: mvsnerf_conv3d_pack_weights_mfma layout),
    //    conv2 = conv3d_k3s1_tiled_kernel<16, 16, 16> (w_f32: mvsnerf_conv3d_pack_weights layout)
    const ActSrc a{x, scale, shift}, b{nullptr, nullptr, nullptr};
    if (Cin == 8) {
        if ((rc = mvs_conv3d_mfma32(a, b, Cin, cin_ld, D, H, W, w_f32, Cout, stride, out, nullptr, st, guard))) return rc;
    } else {
        conv3d_k3s1_tiled_kernel<16, 16, 16><<<dim3((unsigned)mvsnerf_conv3d_tiled_tiles(D, H, W), 1), 256, 0, st>>>(a, b, cin_ld, D, H, W, w_f32, out, g_conv_xcd, nullptr, guard);
        MVS_LAUNCH_CHECK();
    }
    if (stats_part) {
        const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
        abn_partial_kernel<16><<<mvsnerf_conv3d_f16x3_slots(), 256, 0, st>>>(out, (int64_t)Do * Ho * Wo, stats_part, guard);
        MVS_LAUNCH_CHECK();
    }
    return consume ? mvs_guard_consume(guard, st) : MVSNERF_OK;
}

// out = leaky(x1*scale1+shift1) [+ leaky(x2*scale2+shift2)]  (materialises an activated tensor, e.g. the final
// 8-channel neural volume  conv0 + conv11(x), models.py:766)
__global__ __launch_bounds__(256) void abn_apply_add_kernel(ActSrc a, ActSrc b, int64_t n4, int C, float* __restrict__ out)
{
    // a thread keeps its channel group for the whole grid-stride walk (the stride is a multiple of C/4), so the per-channel
    // (scale, shift) are loaded once and no 64-bit modulo sits on the streaming path (the first version did one per float4
    // and ran at 1.7 TB/s)
    const int g4 = C >> 2;
    const int64_t t0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;          // multiple of 256, hence of g4 (C in {8,16,32,64})
    const int c = (int)(t0 % g4) * 4;
    f32x4 sa, ha, sb = {0, 0, 0, 0}, hb = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) { sa[k] = a.scale[c + k]; ha[k] = a.shift[c + k]; if (b.x) { sb[k] = b.scale[c + k]; hb[k] = b.shift[c + k]; } }
    for (int64_t i = t0; i < n4; i += stride) {
        f32x4 v = *reinterpret_cast<const f32x4*>(a.x + i * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_apply(v[k], sa[k], ha[k]);
        if (b.x) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(b.x + i * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += act_apply(t[k], sb[k], hb[k]);
        }
        *reinterpret_cast<f32x4*>(out + i * 4) = v;
    }
}

extern "C" int mvsnerf_abn_apply_add(const float* x1, const float* scale1, const float* shift1,
                                     const float* x2, const float* scale2, const float* shift2,
                                     int64_t n_vox, int C, float* out, void* stream)
{
    if (!x1 || !scale1 || !shift1 || !out || n_vox < 1 || (C & 3) || (1024 % C)) return MVSNERF_EINVAL;
    if (x2 && (!scale2 || !shift2)) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x1) || !mvs_aligned16(out) || (x2 && !mvs_aligned16(x2))) return MVSNERF_EALIGN;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    const int64_t n4 = n_vox * C / 4;
    const unsigned nblk = mvs_cdiv(n4, 256) < 8192u ? mvs_cdiv(n4, 256) : 8192u;       // <= 32 workgroups per CU, grid-stride beyond
    abn_apply_add_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(a, b, n4, C, out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// The 8-channel sum of two lazily-activated tensors, written DEPTH-FASTEST: a, b [D][H][W][8] -> out[H][W][D][8] (MVSNERF_VOL_HWDC, the
// layout the ray march reads best: sample_dev.h).  A workgroup owns a tile of 32 depth planes x 16 x-columns of one row y: it reads 32
// runs of 512 B (along x), applies leaky(x * scale + shift) to both operands, adds, parks the tile in LDS and writes 16 runs of 1 KB (along
// depth): 82 us at config 2, what the channel-last epilogue takes for the same 450 MB (78 us); a 32 x 32 tile (34 KB of LDS, four
// workgroups per CU instead of nine) took 101 us.  Both sides of the transpose are contiguous; nothing else differs from abn_apply_add_kernel (same operation order per element).
__global__ __launch_bounds__(256) void abn_apply_add_hwdc_kernel(ActSrc a, ActSrc b, int D, int H, int W, float* __restrict__ out)
{
    constexpr int TZ = 32, TX = 16, ROW = TX * 8 + 8;             // LDS row of a depth plane: 128 floats + 8 pad (spreads the column reads over the banks)
    __shared__ __attribute__((aligned(16))) float tile[TZ * ROW];
    const int nbx = (W + TX - 1) / TX, nbz = (D + TZ - 1) / TZ;
    const int bx = blockIdx.x % nbx, bz = (blockIdx.x / nbx) % nbz, y = blockIdx.x / (nbx * nbz);
    const int x0 = bx * TX, z0 = bz * TZ;
    const int tid = threadIdx.x;
    const int c4 = (tid & 1) * 4;                                  // this thread's channel quad, in both phases
    f32x4 sa, ha, sb = {0, 0, 0, 0}, hb = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; ++k) { sa[k] = a.scale[c4 + k]; ha[k] = a.shift[c4 + k]; if (b.x) { sb[k] = b.scale[c4 + k]; hb[k] = b.shift[c4 + k]; } }
    // phase 1: float4 number f of the tile = (plane f / 32, column (f % 32) / 2, quad f % 2): consecutive threads read consecutive 16 bytes
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = tid + 256 * r, zi = f >> 5, xi = (f & 31) >> 1;
        const int z = z0 + zi, x = x0 + xi;
        if (z < D && x < W) {
            const int64_t at = ((((int64_t)z * H + y) * W + x) << 3) + c4;
            f32x4 v = *reinterpret_cast<const f32x4*>(a.x + at);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = act_apply(v[k], sa[k], ha[k]);
            if (b.x) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(b.x + at);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += act_apply(t[k], sb[k], hb[k]);
            }
            *reinterpret_cast<f32x4*>(tile + zi * ROW + xi * 8 + c4) = v;
        }
    }
    __syncthreads();
    // phase 2: float4 number f = (column f / 64, plane (f % 64) / 2, quad f % 2): consecutive threads write consecutive 16 bytes of a column's run
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int f = tid + 256 * r, xi = f >> 6, zi = (f & 63) >> 1;
        const int z = z0 + zi, x = x0 + xi;
        if (z < D && x < W)
            *reinterpret_cast<f32x4*>(out + (((((int64_t)y * W + x) * D + z) << 3) + c4)) = *reinterpret_cast<const f32x4*>(tile + zi * ROW + xi * 8 + c4);
    }
}

extern "C" int mvsnerf_abn_apply_add_hwdc(const float* x1, const float* scale1, const float* shift1,
                                          const float* x2, const float* scale2, const float* shift2,
                                          int D, int H, int W, float* out, void* stream)
{
    if (!x1 || !scale1 || !shift1 || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (x2 && (!scale2 || !shift2)) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x1) || !mvs_aligned16(out) || (x2 && !mvs_aligned16(x2))) return MVSNERF_EALIGN;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    const int64_t nblk = (int64_t)((W + 15) / 16) * ((D + 31) / 32) * H;
    if (nblk >= ((int64_t)1 << 31)) return MVSNERF_EUNSUPPORTED;
    abn_apply_add_hwdc_kernel<<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(a, b, D, H, W, out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// =============================================================================================
// BACKWARD of the encoder (generalizable training, train_mvs_nerf_pl.py:104-168)
// =============================================================================================
// ---- train-mode InPlaceABN backward.  y = leaky(g*xhat + b), xhat = (x-mean)*invstd, g = |w|+eps.
//   gp = gy * leaky'(pre);  S1 = sum gp;  S2 = sum gp*xhat
//   gx = g*invstd * (gp - S1/n - xhat*S2/n);   d b = S1;  d w = sign(w) * S2
// gy may be the sum of two upstream tensors (U-Net skips fan out).
template <int C>
__global__ __launch_bounds__(256) void abn_bwd_partial_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ g1, const float* __restrict__ g2, int64_t n, float* __restrict__ part)
{
    constexpr int G = C / 4;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)gridDim.x * blockDim.x;
    const int g = (int)(t % G);
    f32x4 sc, sh, mu, is;
#pragma unroll
    for (int k = 0; k < 4; ++k) { sc[k] = scale[g * 4 + k]; sh[k] = shift[g * 4 + k]; mu[k] = mean[g * 4 + k]; is[k] = invstd[g * 4 + k]; }
    f32x4 s = {0, 0, 0, 0}, q = {0, 0, 0, 0};
    for (int64_t v = t / G; v < n; v += total / G) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + v * C + g * 4);
        f32x4 gy = *reinterpret_cast<const f32x4*>(g1 + v * C + g * 4);
        if (g2) gy += *reinterpret_cast<const f32x4*>(g2 + v * C + g * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float pre = fmaf(a[k], sc[k], sh[k]);
            const float gp = pre > 0.f ? gy[k] : 0.01f * gy[k];
            s[k] += gp;
            q[k] += gp * ((a[k] - mu[k]) * is[k]);
        }
    }
    __shared__ float shm[256 * 8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { shm[threadIdx.x * 8 + k] = s[k]; shm[threadIdx.x * 8 + 4 + k] = q[k]; }
    __syncthreads();
    if (threadIdx.x < C * 2) {
        const int c = threadIdx.x % C, which = threadIdx.x / C;
        const int gg = c / 4, k = c % 4;
        float acc = 0.f;
        for (int j = gg; j < 256; j += G) acc += shm[j * 8 + which * 4 + k];
        part[abn_part_at(which, c, C, blockIdx.x, gridDim.x)] = acc;
    }
}

__global__ __launch_bounds__(64) void abn_bwd_finalize_kernel(const float* __restrict__ part, int nblocks, int C, int64_t n,
                                                             const float* __restrict__ weight, float* __restrict__ m1, float* __restrict__ m2,
                                                             float* __restrict__ g_weight, float* __restrict__ g_bias)
{
    const int c = blockIdx.x, lane = threadIdx.x;
    const float* ps = part + abn_part_at(0, c, C, 0, nblocks);
    const float* pq = part + abn_part_at(1, c, C, 0, nblocks);
    double s = 0.0, q = 0.0;
    int b = lane;
    for (; b + 192 < nblocks; b += 256) {                              // contiguous runs (channel-major partials), four loads in flight
        const float s0 = ps[b], s1 = ps[b + 64], s2 = ps[b + 128], s3 = ps[b + 192];
        const float q0 = pq[b], q1 = pq[b + 64], q2 = pq[b + 128], q3 = pq[b + 192];
        s += ((double)s0 + (double)s1) + ((double)s2 + (double)s3);
        q += ((double)q0 + (double)q1) + ((double)q2 + (double)q3);
    }
    for (; b < nblocks; b += 64) { s += (double)ps[b]; q += (double)pq[b]; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { s += __shfl_xor(s, d); q += __shfl_xor(q, d); }
    if (lane != 0) return;
    m1[c] = (float)(s / (double)n);
    m2[c] = (float)(q / (double)n);
    g_bias[c] = (float)s;
    g_weight[c] = weight[c] < 0.f ? -(float)q : (float)q;          // d|w|/dw
}

template <int C>
__global__ __launch_bounds__(256) void abn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ g1, const float* __restrict__ g2,
                                                           const float* __restrict__ m1, const float* __restrict__ m2, int64_t n4, float* __restrict__ gx)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)((i * 4) % C);
    const f32x4 a = *reinterpret_cast<const f32x4*>(x + i * 4);
    f32x4 gy = *reinterpret_cast<const f32x4*>(g1 + i * 4);
    if (g2) gy += *reinterpret_cast<const f32x4*>(g2 + i * 4);
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float pre = fmaf(a[k], scale[c + k], shift[c + k]);
        const float gp = pre > 0.f ? gy[k] : 0.01f * gy[k];
        const float xh = (a[k] - mean[c + k]) * invstd[c + k];
        o[k] = scale[c + k] * (gp - m1[c + k] - xh * m2[c + k]);
    }
    *reinterpret_cast<f32x4*>(gx + i * 4) = o;
}

extern "C" int mvsnerf_abn_bwd(const float* x, int64_t n_vox, int C, const float* weight, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* g1, const float* g2,
                               float* gx, float* g_weight, float* g_bias, float* workspace, void* stream)
{
    if (!x || !weight || !scale || !shift || !mean || !invstd || !g1 || !gx || !g_weight || !g_bias || !workspace || n_vox < 1) return MVSNERF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int nb = (int)((n_vox * (C / 4) + 255) / 256);
    if (nb > 1024) nb = 1024;
    float* m1 = workspace + (size_t)1024 * 2 * C;       // workspace: mvsnerf_abn_workspace_floats(C) + 2*C
    float* m2 = m1 + C;
    const int64_t n4 = n_vox * C / 4;
#define MVS_ABNB(CC)                                                                                                            \
    abn_bwd_partial_kernel<CC><<<nb, 256, 0, st>>>(x, scale, shift, mean, invstd, g1, g2, n_vox, workspace);                        \
    abn_bwd_finalize_kernel<<<CC, 64, 0, st>>>(workspace, nb, CC, n_vox, weight, m1, m2, g_weight, g_bias);                         \
    abn_bwd_apply_kernel<CC><<<mvs_cdiv(n4, 256), 256, 0, st>>>(x, scale, shift, mean, invstd, g1, g2, m1, m2, n4, gx)
    switch (C) {
        case 8:  MVS_ABNB(8); break;
        case 16: MVS_ABNB(16); break;
        case 32: MVS_ABNB(32); break;
        case 64: MVS_ABNB(64); break;
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_ABNB
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---- weight gradient of a k3 convolution (stride S, padding 1):
//   gW[a][b][tap] = sum_o G[o][a] * X[o*S - 1 + tap][b]
// G lives on the conv's OUTPUT grid (A channels), X on its INPUT grid (B channels).  For Conv3d G = grad of the raw
// output and X = the (lazily activated) input; for ConvTranspose3d the roles swap (G = activated coarse input,
// X = grad of the fine raw output) and the result is already in ConvTranspose3d's (Cin,Cout,27) layout.
// Organisation: a lane owns one (dz,dy) tap row x 4 consecutive input channels (one 16-byte vector of X) and
// accumulates all 3 dx taps x 4 channels x 8 `a` channels = 96 sums in registers while it walks ONE output row along x.
// Walking along x the three dx taps slide over the same X vectors, so a step costs one new 16-byte X load (two for
// stride 2) and one 32-byte G load per 96 FMAs (the pair-per-thread predecessor did one 4-byte load per 8 FMAs and
// ran at 7.5 TFLOP/s).  npr = 9*ceil(B/4) lanes cover one output row; a 256-thread workgroup carries R = 256/npr rows
// at a time and grid-strides over row groups; every (workgroup,row slot) leaves one partial that stage 2 reduces.
__device__ __forceinline__ f32x4 wg_load_x(const ActSrc& x1, const ActSrc& x2, const float* p1, const float* p2, int ix, int Wi, int ldx, bool row_ok,
                                           const f32x4& s1, const f32x4& h1, const f32x4& s2, const f32x4& h2)
{
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row_ok && (unsigned)ix < (unsigned)Wi) {
        v = *reinterpret_cast<const f32x4*>(p1 + (int64_t)ix * ldx);
        if (x1.scale) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = act_apply(v[k], s1[k], h1[k]);
        }
        if (x2.x) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(p2 + (int64_t)ix * ldx);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += act_apply(t[k], s2[k], h2[k]);
        }
    }
    return v;
}

struct G8 { f32x4 lo, hi; };

__device__ __forceinline__ G8 wg_load_g(const ActSrc& g1, const ActSrc& g2, const float* p1, const float* p2, int ox, int Wo, int A, int a0, bool live)
{
    G8 g{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (live && ox < Wo) {
        g.lo = *reinterpret_cast<const f32x4*>(p1 + (int64_t)ox * A);
        g.hi = *reinterpret_cast<const f32x4*>(p1 + (int64_t)ox * A + 4);
        if (g1.scale) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { g.lo[k] = act_apply(g.lo[k], g1.scale[a0 + k], g1.shift[a0 + k]); g.hi[k] = act_apply(g.hi[k], g1.scale[a0 + 4 + k], g1.shift[a0 + 4 + k]); }
        }
        if (g2.x) {
            const f32x4 tl = *reinterpret_cast<const f32x4*>(p2 + (int64_t)ox * A), th = *reinterpret_cast<const f32x4*>(p2 + (int64_t)ox * A + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) { g.lo[k] += act_apply(tl[k], g2.scale[a0 + k], g2.shift[a0 + k]); g.hi[k] += act_apply(th[k], g2.scale[a0 + 4 + k], g2.shift[a0 + 4 + k]); }
        }
    }
    return g;
}

// Built without packed fp32 instructions: with them the compiler broadcasts w[k] through `v_pk_fma_f32 ... op_sel:[0,1,0]` (the high half of src1 feeding
// the low result), the one packed form that returns wrong values in lanes 48-63 while the fp16x3 conv0 runs on another stream
// (profiles/r05_pk_fma_opsel_reproducer.txt).  csrc/check_isa.sh fails the build if that form shows up in any code object of the library.
#if defined(__HIP_DEVICE_COMPILE__)
#define MVS_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define MVS_NO_PACKED_FP32                                  // the host pass does not know the device feature
#endif
template <int S>
__global__ __launch_bounds__(256) MVS_NO_PACKED_FP32 void conv3d_wgrad_rows_kernel(ActSrc g1, ActSrc g2, int A, ActSrc x1, ActSrc x2, int B, int ldx,
                                                               int Do, int Ho, int Wo, int Di, int Hi, int Wi, int nb4, int R,
                                                               float* __restrict__ partial)
{
    const int a0 = blockIdx.y * 8;
    const int npr = nb4 * 9;
    const int r = threadIdx.x / npr, it = threadIdx.x - r * npr;
    const bool active = r < R;
    const int b4 = it % nb4, dyz = it / nb4;
    const int dy = dyz % 3 - 1, dz = dyz / 3 - 1;
    const int c0 = b4 * 4;
    f32x4 s1 = {1, 1, 1, 1}, h1 = {0, 0, 0, 0}, s2 = s1, h2 = h1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + k < B ? c0 + k : B - 1;
        if (x1.scale) { s1[k] = x1.scale[c]; h1[k] = x1.shift[c]; }
        if (x2.x) { s2[k] = x2.scale[c]; h2[k] = x2.shift[c]; }
    }
    float acc[3][4][8];
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int a = 0; a < 8; ++a) acc[d][k][a] = 0.f;
    const int nrows = Do * Ho;
    for (int grp = blockIdx.x; grp * R < nrows; grp += gridDim.x) {
        const int row = grp * R + r;
        const bool live = active && row < nrows;
        const int oz = live ? row / Ho : 0, oy = live ? row - oz * Ho : 0;
        const int zi = oz * S + dz, yi = oy * S + dy;
        const bool row_ok = live && (unsigned)zi < (unsigned)Di && (unsigned)yi < (unsigned)Hi;
        const int64_t xoff = row_ok ? (((int64_t)zi * Hi + yi) * Wi) * ldx + c0 : 0;
        const float* px1 = x1.x + xoff;
        const float* px2 = x2.x ? x2.x + xoff : nullptr;
        const int64_t goff = live ? ((int64_t)row * Wo) * A + a0 : 0;
        const float* pg1 = g1.x + goff;
        const float* pg2 = g2.x ? g2.x + goff : nullptr;
#define MVS_LX(ix) wg_load_x(x1, x2, px1, px2, (ix), Wi, ldx, row_ok, s1, h1, s2, h2)
#define MVS_LG(ox) wg_load_g(g1, g2, pg1, pg2, (ox), Wo, A, a0, live)
        f32x4 w0 = MVS_LX(-1), w1 = MVS_LX(0), w2 = MVS_LX(1);
        G8 gc = MVS_LG(0);
#pragma unroll 3                                            // three steps per trip: the sliding window (w0 <- w1 <- w2) renames instead of moving 20 registers per step
        for (int ox = 0; ox < Wo; ++ox) {
            // next step's operands are requested before this step's 96 FMAs
            const G8 gn = MVS_LG(ox + 1);
            f32x4 n1, n2;
            if (S == 1) { n2 = MVS_LX(ox + 2); }
            else        { n1 = MVS_LX(2 * ox + 2); n2 = MVS_LX(2 * ox + 3); }
            const float gv[8] = {gc.lo[0], gc.lo[1], gc.lo[2], gc.lo[3], gc.hi[0], gc.hi[1], gc.hi[2], gc.hi[3]};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int a = 0; a < 8; ++a) {
                    acc[0][k][a] = fmaf(gv[a], w0[k], acc[0][k][a]);
                    acc[1][k][a] = fmaf(gv[a], w1[k], acc[1][k][a]);
                    acc[2][k][a] = fmaf(gv[a], w2[k], acc[2][k][a]);
                }
            if (S == 1) { w0 = w1; w1 = w2; w2 = n2; }
            else        { w0 = w2; w1 = n1; w2 = n2; }
            gc = gn;
        }
#undef MVS_LX
#undef MVS_LG
    }
    if (!active) return;
    // partial[(blockIdx.x*R + r)][a][b][tap], tap = (dz*3 + dy)*3 + dx
    float* po = partial + ((int64_t)(blockIdx.x * R + r) * A + a0) * B * 27;
    const int tap0 = ((dz + 1) * 3 + (dy + 1)) * 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (c0 + k >= B) continue;
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int d = 0; d < 3; ++d) po[((int64_t)a * B + c0 + k) * 27 + tap0 + d] = acc[d][k][a];
    }
}

int mvs_conv3d_wgrad_mfma4_parts(int A, int B, int Do, int Ho, int Wo, int stride, int cap_parts);
int mvs_conv3d_wgrad_mfma4(const ActSrc& g1, const ActSrc& g2, int A, const ActSrc& x1, const ActSrc& x2, int B, int ldx, int Do, int Ho, int Wo,
                           int Di, int Hi, int Wi, int stride, float* partial, int cap_parts, hipStream_t st);

// number of partial results the workspace is sized for
static int wgrad3d_part_cap(int A, int B)
{
    const int64_t n_out = (int64_t)A * B * 27, by_mem = ((int64_t)16 << 20) / n_out;
    return (int)(by_mem < 2048 ? 2048 : (by_mem > 16384 ? 16384 : by_mem));
}

extern "C" size_t mvsnerf_conv3d_wgrad_workspace_floats(int A, int B) { return (size_t)(wgrad3d_part_cap(A, B) + MVS_RED_SLICES) * A * B * 27; }

extern "C" int mvsnerf_conv3d_wgrad(const float* g1, const float* g1_scale, const float* g1_shift,
                                    const float* g2, const float* g2_scale, const float* g2_shift, int A,
                                    const float* x1, const float* x1_scale, const float* x1_shift,
                                    const float* x2, const float* x2_scale, const float* x2_shift, int B, int ldx,
                                    int Do, int Ho, int Wo, int Di, int Hi, int Wi, int stride,
                                    float* gw, float* workspace, void* stream)
{
    if (!g1 || !x1 || !workspace || A < 8 || (A & 7) || B < 1 || ldx < B || (ldx & 3)) return MVSNERF_EINVAL;
    if (stride != 1 && stride != 2) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(g1) || !mvs_aligned16(x1) || (g2 && !mvs_aligned16(g2)) || (x2 && !mvs_aligned16(x2))) return MVSNERF_EALIGN;
    const int nb4 = (B + 3) / 4, npr = nb4 * 9;
    if (npr > 256 || (nb4 * 4 > ldx)) return MVSNERF_EUNSUPPORTED;
    const ActSrc G1{g1, g1_scale, g1_shift}, G2{g2, g2_scale, g2_shift}, X1{x1, x1_scale, x1_shift}, X2{x2, x2_scale, x2_shift};
    if (g_conv_mfma && !x2) {                              // 16 / 32 / 64 `a` channels: matrix cores (wgrad_mfma.hip)
        const int cap = wgrad3d_part_cap(A, B);
        const int rc = mvs_conv3d_wgrad_mfma4(G1, G2, A, X1, X2, B, ldx, Do, Ho, Wo, Di, Hi, Wi, stride, workspace, cap, (hipStream_t)stream);
        if (rc != MVSNERF_EUNSUPPORTED) {
            if (rc != MVSNERF_OK || !gw) return rc;
            const int64_t n_out = (int64_t)A * B * 27;
            mvs_partial_sum(workspace, mvs_conv3d_wgrad_mfma4_parts(A, B, Do, Ho, Wo, stride, cap), n_out, workspace + (size_t)cap * n_out, gw,
                            (hipStream_t)stream);
            MVS_LAUNCH_CHECK();
            return MVSNERF_OK;
        }
    }
    const int R = 256 / npr;
    const int nrows = Do * Ho, ngroups = (nrows + R - 1) / R;
    const int cap = wgrad3d_part_cap(A, B) / R;
    const int nwg = ngroups < cap ? ngroups : cap;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(nwg, A / 8);
    if (stride == 1) conv3d_wgrad_rows_kernel<1><<<grid, 256, 0, st>>>(G1, G2, A, X1, X2, B, ldx, Do, Ho, Wo, Di, Hi, Wi, nb4, R, workspace);
    else             conv3d_wgrad_rows_kernel<2><<<grid, 256, 0, st>>>(G1, G2, A, X1, X2, B, ldx, Do, Ho, Wo, Di, Hi, Wi, nb4, R, workspace);
    MVS_LAUNCH_CHECK();
    if (!gw) return MVSNERF_OK;                            // partials left in `workspace` for mvsnerf_partial_sum_multi
    const int64_t n_out = (int64_t)A * B * 27;
    mvs_partial_sum(workspace, nwg * R, n_out, workspace + (size_t)wgrad3d_part_cap(A, B) * n_out, gw, st);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// number of partial results mvsnerf_conv3d_wgrad leaves at the start of its workspace (rows of A*B*27 floats)
extern "C" int mvsnerf_conv3d_wgrad_parts(int A, int B, int Do, int Ho, int Wo, int stride, int two_x_sources)
{
    if (A < 8 || B < 1) return 0;
    if (g_conv_mfma && !two_x_sources) {
        const int n = mvs_conv3d_wgrad_mfma4_parts(A, B, Do, Ho, Wo, stride, wgrad3d_part_cap(A, B));
        if (n > 0) return n;
    }
    const int npr = ((B + 3) / 4) * 9;
    if (npr > 256) return 0;
    const int R = 256 / npr, ngroups = (Do * Ho + R - 1) / R, cap = wgrad3d_part_cap(A, B) / R;
    return (ngroups < cap ? ngroups : cap) * R;
}

// dst_j[i] = sum_p partial_j[p][i] for n_jobs independent reductions in two launches (host arrays, copied into the kernel arguments).
// scratch: mvsnerf_partial_sum_multi_scratch_floats(sum of n_out) floats.
extern "C" size_t mvsnerf_partial_sum_multi_scratch_floats(int64_t total_n_out) { return (size_t)MVS_RED_SLICES * (size_t)total_n_out; }

extern "C" int mvsnerf_partial_sum_multi(int n_jobs, const float* const* partial, const int* n_part, const int64_t* n_out, float* const* dst,
                                         float* scratch, void* stream)
{
    if (n_jobs < 1 || n_jobs > MVS_PSUM_JOBS || !partial || !n_part || !n_out || !dst || !scratch) return MVSNERF_EINVAL;
    PsumJobs J;
    J.n = n_jobs;
    J.vec4 = 0;
    int b1 = 0, b2 = 0;
    size_t off = 0;
    for (int j = 0; j < n_jobs; ++j) {
        if (!partial[j] || !dst[j] || n_part[j] < 1 || n_out[j] < 1) return MVSNERF_EINVAL;
        // scratch + off stays 16-byte aligned as long as every job before this one was a vector job (slices * n_out multiples of 4)
        const bool vec4 = (n_out[j] & 3) == 0 && mvs_aligned16(partial[j]) && mvs_aligned16(dst[j]) && mvs_aligned16(scratch + off);
        if (vec4) J.vec4 |= 1u << j;
        const int gx = mvs_psum_gx(n_out[j], vec4);
        const int chunk = n_part[j] <= MVS_RED_SLICES ? n_part[j] : (n_part[j] + MVS_RED_SLICES - 1) / MVS_RED_SLICES;
        const int slices = (n_part[j] + chunk - 1) / chunk;
        J.partial[j] = partial[j]; J.dst[j] = dst[j]; J.scratch[j] = scratch + off; J.n_out[j] = n_out[j];
        J.n_part[j] = n_part[j]; J.chunk[j] = chunk; J.slices[j] = slices;
        J.blk1[j] = b1; J.blk2[j] = b2;
        b1 += gx * slices;
        b2 += slices > 1 ? gx : 0;
        off += (size_t)slices * (size_t)n_out[j];
    }
    J.blk1[n_jobs] = b1; J.blk2[n_jobs] = b2;
    hipStream_t st = (hipStream_t)stream;
    mvs_partial_sum_multi_kernel<1><<<b1, 256, 0, st>>>(J);
    MVS_LAUNCH_CHECK();
    if (b2 > 0) {
        // stage 2 walks only the jobs that have slices: give the others an empty block range
        mvs_partial_sum_multi_kernel<2><<<b2, 256, 0, st>>>(J);
        MVS_LAUNCH_CHECK();
    }
    return MVSNERF_OK;
}

// conv0's weight gradient with the cost volume in channel blocks of four (what mvsnerf_planesweep_costvar_blocked_fwd writes): matrix
// cores, conv_mfma.hip.  g: gradient of conv0's raw output, channel-last [D][H][W][8].  gw: (8, Cin_real, 3,3,3).  workspace:
// mvsnerf_conv3d_wgrad_workspace_floats(8, Cin_real) floats.  Deterministic (fixed partial-sum order).
int mvs_conv3d_c8_wgrad4(const float* x4, int Cin, int cin_real, int D, int H, int W, const float* g, float* gw, float* workspace, int cap_parts,
                         hipStream_t st);

int mvs_conv3d_c8_wgrad4_parts(int Cin, int D, int H, int W, int cap_parts);
extern "C" int mvsnerf_conv3d_c8_blocked_wgrad_parts(int Cin, int Cin_real, int D, int H, int W)
{
    return mvs_conv3d_c8_wgrad4_parts(Cin, D, H, W, wgrad3d_part_cap(8, Cin_real));
}

extern "C" int mvsnerf_conv3d_c8_blocked_wgrad(const float* x_blocked, int Cin, int Cin_real, int D, int H, int W, const float* g, float* gw,
                                               float* workspace, void* stream)
{
    if (!x_blocked || !g || !workspace || D < 1 || H < 1 || W < 1 || Cin < 4 || (Cin & 3) || Cin_real < 1 || Cin_real > Cin || Cin_real <= Cin - 4)
        return MVSNERF_EINVAL;
    if (!mvs_aligned16(x_blocked) || !mvs_aligned16(g)) return MVSNERF_EALIGN;
    if ((int64_t)D * H * W * 8 >= (int64_t)1 << 31) return MVSNERF_EUNSUPPORTED;
    return mvs_conv3d_c8_wgrad4(x_blocked, Cin, Cin_real, D, H, W, g, gw, workspace, wgrad3d_part_cap(8, Cin_real), (hipStream_t)stream);
}

// ---- plane-sweep backward: d cost[...variance channels] -> d feats (channel-last [V][H][W][32]).
// var_c = s2*inv - (s*inv)^2 with s = sum of warped values, so for every contributing value w:
//   d w = g_var * 2*inv*(w - s*inv).   Warped values are bilinear gathers => their gradient is a bilinear scatter
// (float atomics).  Masks/counts are step functions (no gradient); thumbnails carry no parameters.
template <int C>
__global__ __launch_bounds__(256) void planesweep_bwd_kernel(const float* __restrict__ feat, const float* __restrict__ proj, const float* __restrict__ depth,
                                                            int V, int H, int W, int D, int pad, const float* __restrict__ g_cost, int CP, int c_var,
                                                            float* __restrict__ g_feat)
{
    // Two phases per workgroup of 256 voxels.  (1) one thread per voxel: projection into every source view (7 divisions per view),
    // bilinear weights and tap indices -> LDS.  (2) one thread per (voxel, channel), 8 voxels per pass: the 32 lanes of a voxel read /
    // atomically update one contiguous 128-byte channel vector per tap (a per-voxel thread with a channel loop issues 4-byte atomics
    // 128 B apart: 10x slower; eight lanes x four channels per voxel: 4x the L2 atomic line visits, slower as well).  Phase 1 used to
    // be repeated by all 32 lanes of every voxel, and the kernel was bound by that arithmetic (2.3 ms).
    constexpr int VPB = 256;
    extern __shared__ __attribute__((aligned(16))) float geo[];    // [VPB][GS]: per source view {w_nw,w_ne,w_sw,w_se, t_nw,t_ne,t_sw,t_se}, then {1/count, ref pixel}
    const int GS = (V - 1) * 8 + 2;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int64_t nvox = (int64_t)D * Hp * Wp;
    const int64_t v0 = (int64_t)blockIdx.x * VPB;
    {
        const int64_t i = v0 + threadIdx.x;
        if (i < nvox) {
            float* o = geo + threadIdx.x * GS;
            const int x = (int)(i % Wp), y = (int)((i / Wp) % Hp), d = (int)(i / ((int64_t)Wp * Hp));
            const float u = (float)(x - pad), v = (float)(y - pad), dep = depth[d];
            const bool interior = x >= pad && x < W + pad && y >= pad && y < H + pad;
            float cnt = 1.0f;
            for (int vv = 1; vv < V; ++vv) {
                const float* P = proj + vv * 12;
                const float p0 = fmaf(P[2], 1.0f, fmaf(P[1], v, P[0] * u)) + P[3] / dep;
                const float p1 = fmaf(P[6], 1.0f, fmaf(P[5], v, P[4] * u)) + P[7] / dep;
                const float p2 = fmaf(P[10], 1.0f, fmaf(P[9], v, P[8] * u)) + P[11] / dep;
                const float gx = (p0 / p2) / ((float)(W - 1) / 2.0f) - 1.0f, gy = (p1 / p2) / ((float)(H - 1) / 2.0f) - 1.0f;
                cnt += (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;
                const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
                const float fx = floorf(ix), fy = floorf(iy);
                const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
                const bool x0in = fx >= 0.f && fx <= (float)(W - 1), x1in = fx + 1.f >= 0.f && fx + 1.f <= (float)(W - 1);
                const bool y0in = fy >= 0.f && fy <= (float)(H - 1), y1in = fy + 1.f >= 0.f && fy + 1.f <= (float)(H - 1);
                float* ov = o + (vv - 1) * 8;
                ov[0] = (x0in && y0in) ? wx0 * wy0 : 0.f; ov[1] = (x1in && y0in) ? wx1 * wy0 : 0.f;
                ov[2] = (x0in && y1in) ? wx0 * wy1 : 0.f; ov[3] = (x1in && y1in) ? wx1 * wy1 : 0.f;
                const bool any = (x0in || x1in) && (y0in || y1in);
                const int xa = any ? min(max((int)fx, 0), W - 1) : 0, xb = any ? min(max((int)fx + 1, 0), W - 1) : 0;
                const int ya = any ? min(max((int)fy, 0), H - 1) : 0, yb = any ? min(max((int)fy + 1, 0), H - 1) : 0;
                ov[4] = __int_as_float(ya * W + xa); ov[5] = __int_as_float(ya * W + xb);
                ov[6] = __int_as_float(yb * W + xa); ov[7] = __int_as_float(yb * W + xb);
            }
            o[(V - 1) * 8] = 1.0f / cnt;
            o[(V - 1) * 8 + 1] = __int_as_float(interior ? (y - pad) * W + (x - pad) : -1);
        }
    }
    __syncthreads();
    const int c = threadIdx.x & (C - 1);
    constexpr int MAXV = 8;
    for (int vl = threadIdx.x / C; vl < VPB; vl += 256 / C) {
        const int64_t i = v0 + vl;
        if (i >= nvox) break;
        const float* o = geo + vl * GS;
        const float inv = o[(V - 1) * 8];
        const int refpix = __float_as_int(o[(V - 1) * 8 + 1]);
        const float gv = g_cost[i * CP + c_var + c];
        const float ref = refpix >= 0 ? feat[(int64_t)refpix * C + c] : 0.f;
        float s = ref;
        float wv[MAXV];
        for (int vv = 1; vv < V; ++vv) {
            const float* ov = o + (vv - 1) * 8;
            const float* fb = feat + (int64_t)vv * H * W * C + c;
            wv[vv] = fmaf(fb[(int64_t)__float_as_int(ov[7]) * C], ov[3], fmaf(fb[(int64_t)__float_as_int(ov[6]) * C], ov[2],
                          fmaf(fb[(int64_t)__float_as_int(ov[5]) * C], ov[1], fb[(int64_t)__float_as_int(ov[4]) * C] * ov[0])));
            s += wv[vv];
        }
        const float k2 = gv * 2.0f * inv, mean = s * inv;
        if (refpix >= 0) atomicAdd(g_feat + (int64_t)refpix * C + c, k2 * (ref - mean));
        for (int vv = 1; vv < V; ++vv) {
            const float* ov = o + (vv - 1) * 8;
            float* gb = g_feat + (int64_t)vv * H * W * C + c;
            const float gw_ = k2 * (wv[vv] - mean);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ov[k] != 0.f) atomicAdd(gb + (int64_t)__float_as_int(ov[4 + k]) * C, gw_ * ov[k]);
        }
    }
}


// ---- helpers of the deterministic plane-sweep backward: largest magnitudes (float bits order like unsigned integers for x >= 0), the
// fixed-point scale every thread derives from them, and the conversion back
__global__ __launch_bounds__(256) void absmax_strided_kernel(const float* __restrict__ x, int64_t n_rows, int ld, int c0, int nc, unsigned* __restrict__ out)
{
    float m = 0.f;
    const int64_t n = n_rows * nc;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / nc;
        m = fmaxf(m, fabsf(x[r * ld + c0 + (int)(i - r * nc)]));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}
__device__ __forceinline__ float psw_fix_scale(const unsigned* mx)
{
    const float b = __uint_as_float(mx[0]) * __uint_as_float(mx[1]);
    if (!(b > 0.f) || !(b < 3.0e38f)) return 1.0f;
    int e;
    frexpf(b, &e);                                                    // b in [2^(e-1), 2^e)
    return ldexpf(1.0f, max(-120, min(120, 36 - e)));
}
__global__ __launch_bounds__(256) void psw_fix_to_float_kernel(const long long* __restrict__ fix, int64_t n, float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float inv = 1.0f / psw_fix_scale(reinterpret_cast<const unsigned*>(fix + n));
    const long long v = fix[i];
    if (v != 0) dst[i] += (float)((double)v * (double)inv);
}

// ---- plane-sweep backward, column form.  A thread owns ONE channel of ONE voxel column (x, y) and walks the depth planes itself.  Along
// a column the sample point in a source view moves by the disparity step - a fraction of a pixel per plane for any rig the sweep is meant
// for (0.07 px at config 2) - so the four tap PIXELS stay the same for many consecutive planes; only the bilinear weights change.  The
// thread therefore keeps, per source view, the four tap values and four gradient accumulators in registers and touches memory only when
// the tap set changes: it sends the accumulators (one float atomic per tap with a non-zero sum) and gathers the new taps.  At config 2
// that is ~10 tap sets per column and view over 128 planes: the gathers and the atomics of the scatter kernel divided by ~13, no LDS
// atomics, no patch, and each gradient value of the cost volume is read exactly once (the tile form read it once per channel half too,
// but per plane it paid 36 gathers and 36 LDS atomics per thread).  What remains per plane and thread is one coalesced 4-byte load,
// the geometry from LDS and ~30 multiply-adds.  A rig whose taps move every plane degrades to the scatter kernel's traffic, not below it.
//   Workgroup = NV = 8 consecutive columns of one row x 32 channels.  Geometry (7 divisions per voxel and view) is computed once per
// voxel, DCH planes at a time: NV x NSRC x DCH items over the 256 threads -> LDS -> read back by the 32 channel lanes (broadcast reads).
// The reference view's gradient of a pixel has exactly one column contributing: register sum over all planes, one plain read-add-write.
// FIX (the deterministic variant, mvsnerf_planesweep_costvar_bwd_det): the source views' sums go to 64-bit FIXED-POINT accumulators - integer
// additions commute, so the result does not depend on the order in which the columns' atomics arrive (float atomics: last-bit differences
// from run to run).  fix = {int64 [NSRC][H][W][C] | float bits of max |g_cost|, max |feat|}: a contribution is rounded to a multiple of
// 2^-k with k = 36 - ceil(log2(max|g| * max|feat|)) (relative 2^-36 of the largest possible contribution: far below fp32's 2^-24) and
// 2^24.3 voxels x 4 |g| |feat| 2^k stay below 2^63.
template <int C, int NSRC, bool FIX>
__global__ __launch_bounds__(256) void planesweep_bwd_columns_kernel(const float* __restrict__ feat, const float* __restrict__ proj,
                                                                     const float* __restrict__ depth, int H, int W, int D, int pad,
                                                                     const float* __restrict__ g_cost, int CP, int c_var, float* __restrict__ g_feat,
                                                                     long long* __restrict__ fix)
{
    static_assert(C == 32, "one lane per channel, 8 columns per workgroup");
    constexpr int NV = 8, DCH = 16, GI = 12;                          // columns per workgroup, planes per geometry batch, floats per item
    // per (plane, column, source view): {w_nw,w_ne,w_sw,w_se | byte offset of the 4 tap pixels in a feature map | view counts (0/1), -, -, -}
    __shared__ __attribute__((aligned(16))) float geo[DCH * NV * NSRC * GI];
    const int tid = threadIdx.x;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int nbx = (Wp + NV - 1) / NV;
    const int y = blockIdx.x / nbx, x0 = (blockIdx.x - y * nbx) * NV;
    const float sx = (float)(W - 1) / 2.0f, sy = (float)(H - 1) / 2.0f;
    const int vl = tid >> 5, c = tid & 31, x = x0 + vl;
    const bool valid = x < Wp;
    const bool interior = valid && x >= pad && x < W + pad && y >= pad && y < H + pad;
    const int64_t refoff = interior ? ((int64_t)(y - pad) * W + (x - pad)) * C + c : 0;
    const float ref = interior ? feat[refoff] : 0.f;
    float racc = 0.f;
    int4 cur[NSRC];                                                   // tap set in the registers (byte offsets), -1: none yet
    float tap[NSRC][4], acc[NSRC][4];
#pragma unroll
    for (int vs = 0; vs < NSRC; ++vs) {
        cur[vs] = int4{-1, -1, -1, -1};
#pragma unroll
        for (int k = 0; k < 4; ++k) { tap[vs][k] = 0.f; acc[vs][k] = 0.f; }
    }
    float fix_scale = 1.0f;
    if (FIX) fix_scale = psw_fix_scale(reinterpret_cast<const unsigned*>(fix + (int64_t)NSRC * H * W * C));
    auto send = [&](int vs) {
        float* gview = g_feat + (int64_t)(vs + 1) * H * W * C + c;
        const int o[4] = {cur[vs].x, cur[vs].y, cur[vs].z, cur[vs].w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (acc[vs][k] != 0.f) {
                if (FIX) atomicAdd(reinterpret_cast<unsigned long long*>(fix + (int64_t)vs * H * W * C + c + (o[k] >> 2)),
                                   (unsigned long long)(long long)rintf(acc[vs][k] * fix_scale));
                else atomicAdd(gview + (o[k] >> 2), acc[vs][k]);
                acc[vs][k] = 0.f;
            }
    };
    const float* gcol = g_cost + ((int64_t)y * Wp + x) * CP + c_var + c;
    const int64_t gplane = (int64_t)Hp * Wp * CP;
    for (int d0 = 0; d0 < D; d0 += DCH) {
        // ---- geometry of DCH planes x NV columns x NSRC views
        for (int it = tid; it < DCH * NV * NSRC; it += 256) {
            const int dl = it / (NV * NSRC), r = it - dl * (NV * NSRC), pv = r / NSRC, vs = r - pv * NSRC;
            const int d = min(d0 + dl, D - 1), vx = x0 + pv;
            const bool pvalid = vx < Wp;
            const float u = (float)(vx - pad), v = (float)(y - pad), dep = depth[d];
            const float* P = proj + (vs + 1) * 12;
            const float p0 = fmaf(P[2], 1.0f, fmaf(P[1], v, P[0] * u)) + P[3] / dep;
            const float p1 = fmaf(P[6], 1.0f, fmaf(P[5], v, P[4] * u)) + P[7] / dep;
            const float p2 = fmaf(P[10], 1.0f, fmaf(P[9], v, P[8] * u)) + P[11] / dep;
            const float gx = (p0 / p2) / sx - 1.0f, gy = (p1 / p2) / sy - 1.0f;
            const bool inside = gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f;
            const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
            const float fxx = floorf(ix), fyy = floorf(iy);
            const float wx1 = ix - fxx, wx0 = (fxx + 1.0f) - ix, wy1 = iy - fyy, wy0 = (fyy + 1.0f) - iy;
            const bool x0in = fxx >= 0.f && fxx <= (float)(W - 1), x1in = fxx + 1.f >= 0.f && fxx + 1.f <= (float)(W - 1);
            const bool y0in = fyy >= 0.f && fyy <= (float)(H - 1), y1in = fyy + 1.f >= 0.f && fyy + 1.f <= (float)(H - 1);
            const bool any = pvalid && (x0in || x1in) && (y0in || y1in);
            const int xa = any ? min(max((int)fxx, 0), W - 1) : 0, xb = any ? min(max((int)fxx + 1, 0), W - 1) : 0;
            const int ya = any ? min(max((int)fyy, 0), H - 1) : 0, yb = any ? min(max((int)fyy + 1, 0), H - 1) : 0;
            float* ov = geo + it * GI;
            *reinterpret_cast<f32x4*>(ov) = f32x4{(any && x0in && y0in) ? wx0 * wy0 : 0.f, (any && x1in && y0in) ? wx1 * wy0 : 0.f,
                                                  (any && x0in && y1in) ? wx0 * wy1 : 0.f, (any && x1in && y1in) ? wx1 * wy1 : 0.f};
            *reinterpret_cast<int4*>(ov + 4) = int4{(ya * W + xa) * C * 4, (ya * W + xb) * C * 4, (yb * W + xa) * C * 4, (yb * W + xb) * C * 4};
            ov[8] = (pvalid && inside) ? 1.0f : 0.0f;
        }
        // this thread's DCH gradient values: all loads in flight before the first is used
        float gv[DCH];
#pragma unroll
        for (int dl = 0; dl < DCH; ++dl) gv[dl] = (valid && d0 + dl < D) ? gcol[(int64_t)(d0 + dl) * gplane] : 0.f;
        __syncthreads();
#pragma unroll 2
        for (int dl = 0; dl < DCH; ++dl) {
            const float* o = geo + (dl * NV + vl) * NSRC * GI;
            float cnt = 1.0f, s = ref, wv[NSRC];
            f32x4 wq[NSRC];
#pragma unroll
            for (int vs = 0; vs < NSRC; ++vs) {
                wq[vs] = *reinterpret_cast<const f32x4*>(o + vs * GI);
                const int4 gb = *reinterpret_cast<const int4*>(o + vs * GI + 4);
                cnt += o[vs * GI + 8];
                if ((gb.x != cur[vs].x) | (gb.w != cur[vs].w)) {   // new tap set (the nw and se offsets determine all four pixels): send, gather
                    send(vs);
                    cur[vs] = gb;
                    const char* fb = reinterpret_cast<const char*>(feat + (int64_t)(vs + 1) * H * W * C + c);
                    tap[vs][0] = *reinterpret_cast<const float*>(fb + gb.x); tap[vs][1] = *reinterpret_cast<const float*>(fb + gb.y);
                    tap[vs][2] = *reinterpret_cast<const float*>(fb + gb.z); tap[vs][3] = *reinterpret_cast<const float*>(fb + gb.w);
                }
                wv[vs] = fmaf(tap[vs][3], wq[vs][3], fmaf(tap[vs][2], wq[vs][2], fmaf(tap[vs][1], wq[vs][1], tap[vs][0] * wq[vs][0])));
                s += wv[vs];
            }
            const float inv = __builtin_amdgcn_rcpf(cnt);                // cnt = 1 .. V: v_rcp_f32 (<= 1 ulp) instead of the ten-instruction IEEE division, per plane and thread
            const float k2 = gv[dl] * 2.0f * inv, mean = s * inv;
            racc += k2 * (ref - mean);
#pragma unroll
            for (int vs = 0; vs < NSRC; ++vs) {
                const float gw_ = k2 * (wv[vs] - mean);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[vs][k] = fmaf(gw_, wq[vs][k], acc[vs][k]);
            }
        }
        __syncthreads();                                              // the next batch overwrites geo
    }
#pragma unroll
    for (int vs = 0; vs < NSRC; ++vs) send(vs);
    if (interior) g_feat[refoff] += racc;                             // the only contribution to this element of view 0
}


template <int NSRC>
static int planesweep_bwd_columns_launch(const float* feat, const float* proj, const float* depth, int H, int W, int D, int pad, const float* g_cost,
                                         int CP, int c_var, float* g_feat, long long* fix, hipStream_t st)
{
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const unsigned grid = (unsigned)(Hp * ((Wp + 7) / 8));
    if (!fix) {
        planesweep_bwd_columns_kernel<32, NSRC, false><<<grid, 256, 0, st>>>(feat, proj, depth, H, W, D, pad, g_cost, CP, c_var, g_feat, nullptr);
        MVS_LAUNCH_CHECK();
        return MVSNERF_OK;
    }
    const int64_t n = (int64_t)NSRC * H * W * 32;
    unsigned* mx = reinterpret_cast<unsigned*>(fix + n);
    absmax_strided_kernel<<<1024, 256, 0, st>>>(g_cost, (int64_t)D * Hp * Wp, CP, c_var, 32, mx);
    absmax_strided_kernel<<<256, 256, 0, st>>>(feat, (int64_t)(NSRC + 1) * H * W, 32, 0, 32, mx + 1);
    planesweep_bwd_columns_kernel<32, NSRC, true><<<grid, 256, 0, st>>>(feat, proj, depth, H, W, D, pad, g_cost, CP, c_var, g_feat, fix);
    psw_fix_to_float_kernel<<<mvs_cdiv(n, 256), 256, 0, st>>>(fix, n, g_feat + (int64_t)H * W * 32);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}


static int planesweep_costvar_bwd_impl(const float* feats_cl, const float* proj, const float* depth, int V, int C, int H, int W, int D, int pad,
                                       const float* g_cost, int CP, int with_img, float* g_feats_cl, long long* fix, void* stream)
{
    if (!feats_cl || !proj || !depth || !g_cost || !g_feats_cl || V < 1 || V > 8 || H < 2 || W < 2 || D < 1 || pad < 0) return MVSNERF_EINVAL;
    if (C != 32) return MVSNERF_EUNSUPPORTED;
    const int64_t nvox = (int64_t)D * (H + 2 * pad) * (W + 2 * pad);
    if (V >= 2) {                                           // column form; a single view has no taps to keep: the per-voxel scatter below
        hipStream_t st = (hipStream_t)stream;
        const int cv = with_img ? 3 * V : 0;
        switch (V - 1) {
#define MVS_PBC(N) case N: return planesweep_bwd_columns_launch<N>(feats_cl, proj, depth, H, W, D, pad, g_cost, CP, cv, g_feats_cl, fix, st)
            MVS_PBC(1); MVS_PBC(2); MVS_PBC(3); MVS_PBC(4); MVS_PBC(5); MVS_PBC(6); MVS_PBC(7);
#undef MVS_PBC
        }
    }
    const size_t lds = (size_t)256 * ((V - 1) * 8 + 2) * sizeof(float);
    planesweep_bwd_kernel<32><<<mvs_cdiv(nvox, 256), 256, lds, (hipStream_t)stream>>>(feats_cl, proj, depth, V, H, W, D, pad, g_cost, CP,
                                                                                    with_img ? 3 * V : 0, g_feats_cl);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_planesweep_costvar_bwd(const float* feats_cl, const float* proj, const float* depth, int V, int C, int H, int W, int D, int pad,
                                              const float* g_cost, int CP, int with_img, float* g_feats_cl, void* stream)
{
    return planesweep_costvar_bwd_impl(feats_cl, proj, depth, V, C, H, W, D, pad, g_cost, CP, with_img, g_feats_cl, nullptr, stream);
}

// int64 words of the deterministic variant's workspace (zeroed by the caller before every call)
extern "C" size_t mvsnerf_planesweep_costvar_bwd_det_workspace_words(int V, int C, int H, int W)
{
    return V >= 1 && C == 32 && H >= 1 && W >= 1 ? (size_t)(V > 1 ? V - 1 : 0) * H * W * C + 1 : 0;
}

extern "C" int mvsnerf_planesweep_costvar_bwd_det(const float* feats_cl, const float* proj, const float* depth, int V, int C, int H, int W, int D, int pad,
                                                  const float* g_cost, int CP, int with_img, float* g_feats_cl, void* workspace_zeroed, void* stream)
{
    if (!workspace_zeroed || ((uintptr_t)workspace_zeroed & 7)) return workspace_zeroed ? MVSNERF_EALIGN : MVSNERF_EINVAL;
    return planesweep_costvar_bwd_impl(feats_cl, proj, depth, V, C, H, W, D, pad, g_cost, CP, with_img, g_feats_cl,
                                       reinterpret_cast<long long*>(workspace_zeroed), stream);
}
