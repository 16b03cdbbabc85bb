// Alpha compositing along each ray: raw2alpha + raw2outputs (renderer.py:18-26, 65-92).
// One 64-lane wavefront per ray.  Lane l owns the contiguous chunk of ceil(S/64) samples starting at
// l*chunk; the transmittance T_i = prod_{j<i}(1-alpha_j+1e-10) is a lane-local running product seeded
// by a wave-level exclusive multiplicative scan (6 __shfl_up steps) of the per-lane chunk products.
// HBM-bound: 20 B read + 8 B written per sample.
#include "common.h"
#include "composite_wave.h"

template <int CHUNK>   // CHUNK > 0: samples per lane known at compile time (values stay in registers, composite_wave.h)
__global__ __launch_bounds__(256) void composite_kernel(
    const float* __restrict__ raw, const float* __restrict__ z, int64_t N, int S, int chunk_rt, CompositeOut o)
{
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (o.guard && blockIdx.x == 0 && threadIdx.x == 0) mvs_guard_consume(o.guard);
    if (ray >= N) return;                                   // wave-uniform
    const float* rr = raw + ray * S * 4;
    const float* zr = z + ray * S;
    if constexpr (CHUNK > 0) {
        f32x4 rv[CHUNK];
#pragma unroll
        for (int i = 0; i < CHUNK; ++i) {
            const int s = lane * CHUNK + i;
            rv[i] = (s < S) ? *reinterpret_cast<const f32x4*>(rr + s * 4) : f32x4{0, 0, 0, 0};
        }
        composite_wave<CHUNK>(rv, zr, ray, S, lane, o);
    } else {
        // long rays (more than 4 samples per lane): the same walk with the samples re-read from memory
        const int chunk = chunk_rt, s0 = lane * chunk;
        float prod = 1.0f;
        for (int i = 0; i < chunk; ++i) {
            const int s = s0 + i;
            if (s < S) prod *= (1.0f - (1.0f - expf(-rr[s * 4 + 3]))) + 1e-10f;
        }
        float scan = prod;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float t = __shfl_up(scan, d);
            if (lane >= d) scan *= t;
        }
        float T = __shfl_up(scan, 1);
        if (lane == 0) T = 1.0f;
        float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
        for (int i = 0; i < chunk; ++i) {
            const int s = s0 + i;
            if (s >= S) break;
            const f32x4 v = *reinterpret_cast<const f32x4*>(rr + s * 4);
            const float a = 1.0f - expf(-v[3]);
            const float w = a * T;
            T *= (1.0f - a) + 1e-10f;
            sr += w * v[0]; sg += w * v[1]; sb += w * v[2];
            sd += w * zr[s]; sa += w;
            if (o.weights) o.weights[ray * S + s] = w;
            if (o.alpha_out) o.alpha_out[ray * S + s] = a;
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            sr += __shfl_xor(sr, d); sg += __shfl_xor(sg, d); sb += __shfl_xor(sb, d);
            sd += __shfl_xor(sd, d); sa += __shfl_xor(sa, d);
        }
        if (lane == 0) {
            const float dsp = 1.0f / fmaxf(1e-10f, sd / sa);
            if (o.white_bkgd) { const float bg = 1.0f - sa; sr += bg; sg += bg; sb += bg; }
            if (o.rgb_map) { o.rgb_map[ray * 3] = sr; o.rgb_map[ray * 3 + 1] = sg; o.rgb_map[ray * 3 + 2] = sb; }
            if (o.depth_map) o.depth_map[ray] = sd;
            if (o.acc_map) o.acc_map[ray] = sa;
            if (o.disp) o.disp[ray] = dsp;
        }
    }
}

int mvs_composite_fwd(const float* raw, const float* z, int64_t N, int S, int white_bkgd, float* rgb_map, float* disp, float* acc, float* weights,
                      float* depth, float* alpha, int* guard, void* stream);

extern "C" int mvsnerf_composite_fwd(const float* raw, const float* z, int64_t N, int S, int white_bkgd,
                                     float* rgb_map, float* disp, float* acc, float* weights,
                                     float* depth, float* alpha, void* stream)
{
    return mvs_composite_fwd(raw, z, N, S, white_bkgd, rgb_map, disp, acc, weights, depth, alpha, nullptr, stream);
}

// guard != NULL: the launch also ends a guarded 16-bit sequence (raymarch.hip)
int mvs_composite_fwd(const float* raw, const float* z, int64_t N, int S, int white_bkgd, float* rgb_map, float* disp, float* acc, float* weights,
                      float* depth, float* alpha, int* guard, void* stream)
{
    if (!raw || !z || N < 0 || S < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(raw)) return MVSNERF_EALIGN;
    if (N == 0) return MVSNERF_OK;
    hipStream_t st = (hipStream_t)stream;
    const int chunk = (S + 63) / 64;
    const unsigned grid = mvs_cdiv(N, 4);
    const CompositeOut o{rgb_map, disp, acc, weights, depth, alpha, white_bkgd, guard};
#define MVS_COMP(C) composite_kernel<C><<<grid, 256, 0, st>>>(raw, z, N, S, chunk, o)
    switch (chunk) {
        case 1: MVS_COMP(1); break;
        case 2: MVS_COMP(2); break;
        case 3: MVS_COMP(3); break;
        case 4: MVS_COMP(4); break;
        default: MVS_COMP(0); break;
    }
#undef MVS_COMP
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---------------------------------------------------------------------------------------------
// Backward of the compositing (autograd of renderer.py:18-26, 65-92 w.r.t. raw).
// With G_i = g_rgb.c_i + g_depth z_i + g_acc + g_w[i]  (total gradient reaching w_i = a_i T_i):
//   d raw[i][c]   = w_i g_rgb[c]
//   d sigma_i     = (1-a_i) * (G_i T_i + g_alpha[i]  -  S_i / t_i),   S_i = sum_{j>i} G_j w_j,  t_i = 1-a_i+1e-10
// One wave per ray: forward quantities are recomputed, S is a reverse (suffix) scan: lane-local reverse
// loop seeded by a wave-level exclusive suffix sum over lanes (__shfl_down).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float* __restrict__ raw, const float* __restrict__ z, int64_t N, int S, int chunk, int white_bkgd,
    const float* __restrict__ g_rgb, const float* __restrict__ g_depth, const float* __restrict__ g_acc,
    const float* __restrict__ g_w, const float* __restrict__ g_alpha, float* __restrict__ d_raw)
{
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= N) return;
    const int s0 = lane * chunk;
    const float* rr = raw + ray * S * 4;
    const float* zr = z + ray * S;
    const float gr = g_rgb ? g_rgb[ray * 3] : 0.f, gg = g_rgb ? g_rgb[ray * 3 + 1] : 0.f, gb = g_rgb ? g_rgb[ray * 3 + 2] : 0.f;
    const float gd = g_depth ? g_depth[ray] : 0.f;
    // white background adds (1 - acc) to every rgb channel: d/dacc -= sum(g_rgb)
    const float ga = (g_acc ? g_acc[ray] : 0.f) - (white_bkgd ? (gr + gg + gb) : 0.f);

    float prod = 1.0f;
    for (int i = 0; i < chunk; ++i) { const int s = s0 + i; if (s < S) prod *= (1.0f - (1.0f - expf(-rr[s * 4 + 3]))) + 1e-10f; }
    float scan = prod;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_up(scan, d); if (lane >= d) scan *= o; }
    float T = __shfl_up(scan, 1);
    if (lane == 0) T = 1.0f;
    const float T0 = T;
    // pass 1: lane-local sum of G_i w_i
    float loc = 0.f;
    for (int i = 0; i < chunk; ++i) {
        const int s = s0 + i; if (s >= S) break;
        const f32x4 v = *reinterpret_cast<const f32x4*>(rr + s * 4);
        const float a = 1.0f - expf(-v[3]);
        const float w = a * T;
        T *= (1.0f - a) + 1e-10f;
        const float G = gr * v[0] + gg * v[1] + gb * v[2] + gd * zr[s] + ga + (g_w ? g_w[ray * S + s] : 0.f);
        loc += G * w;
    }
    // exclusive suffix sum over lanes
    float suf = loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const float o = __shfl_down(suf, d); if (lane + d < 64) suf += o; }
    float Sx = suf - loc;                                   // sum over later lanes
    // pass 2: forward re-walk; S_i = Sx + (sum of this lane's G_j w_j with j > i) = Sx + (loc - inclusive prefix)
    T = T0;
    float pre = 0.f;
    for (int i = 0; i < chunk; ++i) {
        const int s = s0 + i; if (s >= S) break;
        const f32x4 v = *reinterpret_cast<const f32x4*>(rr + s * 4);
        const float a = 1.0f - expf(-v[3]);
        const float t = (1.0f - a) + 1e-10f;
        const float w = a * T;
        const float G = gr * v[0] + gg * v[1] + gb * v[2] + gd * zr[s] + ga + (g_w ? g_w[ray * S + s] : 0.f);
        pre += G * w;
        const float Si = Sx + (loc - pre);
        const float dsig = (1.0f - a) * (G * T + (g_alpha ? g_alpha[ray * S + s] : 0.f) - Si / t);
        *reinterpret_cast<f32x4*>(d_raw + (ray * S + s) * 4) = f32x4{w * gr, w * gg, w * gb, dsig};
        T *= t;
    }
}

extern "C" int mvsnerf_composite_bwd(const float* raw, const float* z, int64_t N, int S, int white_bkgd,
                                     const float* g_rgb, const float* g_depth, const float* g_acc,
                                     const float* g_weights, const float* g_alpha, float* d_raw, void* stream)
{
    if (!raw || !z || !d_raw || N < 0 || S < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(raw) || !mvs_aligned16(d_raw)) return MVSNERF_EALIGN;
    if (N == 0) return MVSNERF_OK;
    composite_bwd_kernel<<<mvs_cdiv(N, 4), 256, 0, (hipStream_t)stream>>>(raw, z, N, S, (S + 63) / 64, white_bkgd,
                                                                          g_rgb, g_depth, g_acc, g_weights, g_alpha, d_raw);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
