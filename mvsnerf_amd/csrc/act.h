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

// The same reduction for up to MVS_PSUM_JOBS independent (partial, dst) pairs in TWO launches in total: a training step has ~30 weight
// gradients, each of which used to pay its own two launches (60 x 9 us).  The jobs travel in the kernel arguments.
constexpr int MVS_PSUM_JOBS = 32;
struct PsumJobs {
    const float* partial[MVS_PSUM_JOBS];
    float* dst[MVS_PSUM_JOBS];
    float* scratch[MVS_PSUM_JOBS];     // [slices][n_out] (unused when slices == 1)
    long long n_out[MVS_PSUM_JOBS];
    int n_part[MVS_PSUM_JOBS], chunk[MVS_PSUM_JOBS], slices[MVS_PSUM_JOBS];
    int blk1[MVS_PSUM_JOBS + 1], blk2[MVS_PSUM_JOBS + 1];     // first workgroup of each job in stage 1 / stage 2
    int n;
    unsigned vec4;                     // bit j: job j walks four outputs per thread (n_out % 4 == 0, 16-byte aligned buffers)
};

// T = float, or f32x4 for the jobs whose rows are whole 16-byte vectors (every convolution's A * B * taps; 6 M threads of 16 scalar loads
// each reduced the 112 MB of a training step's 3-D partials at 2.2 TB/s): the same additions per output element in the same order.
template <typename T>
static __device__ __forceinline__ void mvs_psum_walk(const float* __restrict__ src, long long n_out, long long i, int p0, int p1, float* __restrict__ out)
{
    // eight independent running sums: the loads of one round are in flight together (a chunk is up to 128 rows, each a latency-bound
    // strided read); the order of the additions is fixed
    T s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = T{};
    int p = p0;
    for (; p + 7 < p1; p += 8) {
        T v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const T*>(src + (long long)(p + k) * n_out + i);
#pragma unroll
        for (int k = 0; k < 8; ++k) s[k] += v[k];
    }
    for (int k = 0; p < p1; ++p, ++k) s[k] += *reinterpret_cast<const T*>(src + (long long)p * n_out + i);
    *reinterpret_cast<T*>(out + i) = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

static inline int mvs_psum_gx(long long n_out, bool vec4) { return (int)((n_out + (vec4 ? 1023 : 255)) / (vec4 ? 1024 : 256)); }

template <int STAGE>
static __global__ __launch_bounds__(256) void mvs_partial_sum_multi_kernel(PsumJobs J)
{
    const int* blk = STAGE == 1 ? J.blk1 : J.blk2;
    int j = 0;
    while (j + 1 < J.n && (int)blockIdx.x >= blk[j + 1]) ++j;
    const int local = blockIdx.x - blk[j];
    const long long n_out = J.n_out[j];
    const bool vec4 = (J.vec4 >> j) & 1;
    const int gx = (int)((n_out + (vec4 ? 1023 : 255)) / (vec4 ? 1024 : 256));
    const int slice = local / gx;
    const long long i = ((long long)(local - slice * gx) * 256 + threadIdx.x) * (vec4 ? 4 : 1);
    if (i >= n_out) return;
    const float* src = STAGE == 1 ? J.partial[j] : J.scratch[j];
    const int n_src = STAGE == 1 ? J.n_part[j] : J.slices[j];
    const int chunk = STAGE == 1 ? J.chunk[j] : n_src;
    const int p0 = slice * chunk, p1 = p0 + chunk < n_src ? p0 + chunk : n_src;
    float* out = (STAGE == 1 && J.slices[j] > 1) ? J.scratch[j] + (long long)slice * n_out : J.dst[j];
    if (vec4) mvs_psum_walk<f32x4>(src, n_out, i, p0, p1, out);
    else mvs_psum_walk<float>(src, n_out, i, p0, p1, out);
}

