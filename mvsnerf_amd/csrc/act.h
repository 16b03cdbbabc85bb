// Lazily-activated operands shared by the 3-D (encoder.hip) and 2-D (featnet.hip) convolution kernels.
#pragma once
#include "common.h"

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

__device__ __forceinline__ float act1(const ActSrc& s, int64_t idx, int c)
{
    float v = s.x[idx];
    if (s.scale) v = act_apply(v, s.scale[c], s.shift[c]);
    return v;
}


// Two-stage reduction of per-workgroup partial results: dst[i] = sum_p partial[p][i].  A single pass with one thread per
// output walks n_part strided values serially on a handful of workgroups (hundreds of microseconds for 2048 partials);
// stage 1 spreads the partial index over MVS_RED_SLICES workgroup rows, stage 2 folds the slices.
constexpr int MVS_RED_SLICES = 32;    // scratch: MVS_RED_SLICES * n_out floats

static __global__ __launch_bounds__(256) void mvs_partial_sum_kernel(const float* __restrict__ partial, int n_part, int64_t n_out, int chunk,
                                                                     float* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const int p0 = blockIdx.y * chunk, p1 = p0 + chunk < n_part ? p0 + chunk : n_part;
    float s0 = 0.f, s1 = 0.f;
    int p = p0;
    for (; p + 1 < p1; p += 2) { s0 += partial[(int64_t)p * n_out + i]; s1 += partial[(int64_t)(p + 1) * n_out + i]; }
    if (p < p1) s0 += partial[(int64_t)p * n_out + i];
    dst[(int64_t)blockIdx.y * n_out + i] = s0 + s1;
}

static inline void mvs_partial_sum(const float* partial, int n_part, int64_t n_out, float* scratch, float* dst, hipStream_t st)
{
    const unsigned gx = mvs_cdiv(n_out, 256);
    if (n_part <= MVS_RED_SLICES) {
        mvs_partial_sum_kernel<<<dim3(gx, 1), 256, 0, st>>>(partial, n_part, n_out, n_part, dst);
        return;
    }
    const int chunk = (n_part + MVS_RED_SLICES - 1) / MVS_RED_SLICES, slices = (n_part + chunk - 1) / chunk;
    mvs_partial_sum_kernel<<<dim3(gx, slices), 256, 0, st>>>(partial, n_part, n_out, chunk, scratch);
    mvs_partial_sum_kernel<<<dim3(gx, 1), 256, 0, st>>>(scratch, slices, n_out, slices, dst);
}
