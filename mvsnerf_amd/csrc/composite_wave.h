// raw2alpha + raw2outputs (renderer.py:18-26, 65-92) for ONE ray by one 64-lane wave.  Separate multiplies and adds with
// contraction off (the reference sums products it has rounded), so the result does not depend on the surrounding code.
// (Measured and dropped: calling this from the epilogue of the MLP kernel when a workgroup holds one whole ray, S == 128 -
// bit-identical, one launch less, but the serial tail of the compositing wave costs the MLP kernel the 5 us the launch saved.)
#pragma once
#include "common.h"

struct CompositeOut {
    float* rgb_map; float* disp; float* acc_map; float* weights; float* depth_map; float* alpha_out; int white_bkgd;
    int* guard;      // non-NULL: this launch ends a guarded 16-bit sequence - count a tripped guard and re-arm it (mvs_guard_consume)
};

// Last step of a guarded 16-bit sequence (include/mvsnerf_hip.h): by one thread of a kernel that is stream-ordered behind the predicated
// fp32 kernels.  guard[0]: tripped (re-armed here), guard[1]: number of sequences that fell back so far.
__device__ __forceinline__ void mvs_guard_consume(int* guard)
{
    if (guard[0]) { guard[1] += 1; guard[0] = 0; }
}

// lane l owns samples s0 = l*NR .. s0+NR-1 (values rv[i] = {r,g,b,sigma}; samples >= S are ignored), zr = z_vals of the ray
template <int NR>
__device__ __forceinline__ void composite_wave(const f32x4 (&rv)[NR], const float* __restrict__ zr, int64_t ray, int S, int lane, const CompositeOut& o)
{
#pragma clang fp contract(off)
    const int s0 = lane * NR;
    float prod = 1.0f;
#pragma unroll
    for (int i = 0; i < NR; ++i)
        if (s0 + i < S) prod *= (1.0f - (1.0f - expf(-rv[i][3]))) + 1e-10f;
    // inclusive multiplicative scan over lanes, then shift to exclusive
    float scan = prod;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float t = __shfl_up(scan, d);
        if (lane >= d) scan *= t;
    }
    float T = __shfl_up(scan, 1);
    if (lane == 0) T = 1.0f;
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int s = s0 + i;
        if (s < S) {
            const f32x4 v = rv[i];
            const float a = 1.0f - expf(-v[3]);                 // renderer.py:22  (dist ignored)
            const float w = a * T;                              // :25
            T *= (1.0f - a) + 1e-10f;                           // :24
            sr += w * v[0]; sg += w * v[1]; sb += w * v[2];
            sd += w * zr[s]; sa += w;
            if (o.weights) o.weights[ray * S + s] = w;
            if (o.alpha_out) o.alpha_out[ray * S + s] = a;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        sr += __shfl_xor(sr, d); sg += __shfl_xor(sg, d); sb += __shfl_xor(sb, d);
        sd += __shfl_xor(sd, d); sa += __shfl_xor(sa, d);
    }
    if (lane == 0) {
        const float dsp = 1.0f / fmaxf(1e-10f, sd / sa);                                 // :87
        if (o.white_bkgd) { const float bg = 1.0f - sa; sr += bg; sg += bg; sb += bg; }   // :90-91
        if (o.rgb_map) { o.rgb_map[ray * 3] = sr; o.rgb_map[ray * 3 + 1] = sg; o.rgb_map[ray * 3 + 2] = sb; }
        if (o.depth_map) o.depth_map[ray] = sd;
        if (o.acc_map) o.acc_map[ray] = sa;
        if (o.disp) o.disp[ray] = dsp;
    }
}

