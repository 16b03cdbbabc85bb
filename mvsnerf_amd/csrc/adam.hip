// torch.optim.Adam's update (the reference's optimizer: train_mvs_nerf_pl.py:84-88, betas (0.9, 0.999), no weight decay, no amsgrad) for ALL
// parameter tensors of a step in one launch.  torch's fused Adam spends three launches of ~25 us on the 78 small tensors of the generalizable
// step (multi_tensor_apply carries at most a few dozen tensors per launch); the job table of this kernel travels in the kernel arguments.
//     m = m + (1 - beta1) (g - m)                        (lerp, as ATen writes it)
//     v = beta2 v + (1 - beta2) g g
//     p = p - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)         bc_i = 1 - beta_i^step, computed by the caller in double
#include "common.h"

constexpr int ADAM_JOBS = 84;                                     // 84 x 40 bytes + block table: inside the 4 KB of kernel arguments
struct AdamJobs {
    float* p[ADAM_JOBS]; const float* g[ADAM_JOBS]; float* m[ADAM_JOBS]; float* v[ADAM_JOBS];
    long long numel[ADAM_JOBS];
    int blk[ADAM_JOBS + 1];                                       // first workgroup of each job (1024 elements per workgroup)
    int n;
};

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamJobs J, float step_size, float omb1, float beta2, float omb2, float eps, float bc2_sqrt)
{
    int lo = 0, hi = J.n;                                         // job of this workgroup: last j with blk[j] <= blockIdx.x
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (J.blk[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
    const int j = lo;
    const long long n = J.numel[j], base = ((long long)blockIdx.x - J.blk[j]) * 1024 + threadIdx.x * 4;
    float* __restrict__ p = J.p[j]; const float* __restrict__ g = J.g[j]; float* __restrict__ m = J.m[j]; float* __restrict__ v = J.v[j];
    auto one = [&](long long i) {
        const float gi = g[i];
        const float mi = m[i] + omb1 * (gi - m[i]);
        const float vi = beta2 * v[i] + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) / bc2_sqrt + eps);
    };
    if (base + 3 < n && ((n & 3) == 0)) {                          // whole quads of a tensor whose size is a multiple of four (base is then 16-byte aligned)
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(g + base);
        f32x4 m4 = *reinterpret_cast<f32x4*>(m + base), v4 = *reinterpret_cast<f32x4*>(v + base), p4 = *reinterpret_cast<f32x4*>(p + base);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m4[k] = m4[k] + omb1 * (g4[k] - m4[k]);
            v4[k] = beta2 * v4[k] + omb2 * g4[k] * g4[k];
            p4[k] -= step_size * m4[k] / (sqrtf(v4[k]) / bc2_sqrt + eps);
        }
        *reinterpret_cast<f32x4*>(m + base) = m4; *reinterpret_cast<f32x4*>(v + base) = v4; *reinterpret_cast<f32x4*>(p + base) = p4;
    } else {
        for (int k = 0; k < 4; ++k)
            if (base + k < n) one(base + k);
    }
}

// One Adam step over n tensors (host arrays of device pointers; fp32, contiguous).  step_size = lr / (1 - beta1^step), bc2_sqrt = sqrt(1 - beta2^step);
// the scalars arrive as doubles (what torch holds them in) and are rounded once, after 1 - beta has been formed.
extern "C" int mvsnerf_adam_step_multi(int n, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* numel,
                                       double step_size, double beta1, double beta2, double eps, double bc2_sqrt, void* stream)
{
    if (n < 0 || (n > 0 && (!p || !g || !m || !v || !numel))) return MVSNERF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    for (int first = 0; first < n; first += ADAM_JOBS) {
        AdamJobs J;
        const int cnt = n - first < ADAM_JOBS ? n - first : ADAM_JOBS;
        int b = 0;
        for (int j = 0; j < cnt; ++j) {
            const int i = first + j;
            if (!p[i] || !g[i] || !m[i] || !v[i] || numel[i] < 0) return MVSNERF_EINVAL;
            if (numel[i] >= ((int64_t)1 << 40)) return MVSNERF_EUNSUPPORTED;
            // 16-byte accesses need 16-byte aligned tensors when the size is a multiple of four
            if ((numel[i] & 3) == 0 && (!mvs_aligned16(p[i]) || !mvs_aligned16(g[i]) || !mvs_aligned16(m[i]) || !mvs_aligned16(v[i]))) return MVSNERF_EALIGN;
            J.p[j] = p[i]; J.g[j] = g[i]; J.m[j] = m[i]; J.v[j] = v[i]; J.numel[j] = numel[i];
            J.blk[j] = b;
            const int64_t nb = (numel[i] + 1023) / 1024;
            if (b + nb >= ((int64_t)1 << 30)) return MVSNERF_EUNSUPPORTED;
            b += (int)nb;
        }
        J.blk[cnt] = b; J.n = cnt;
        // 1 - beta in DOUBLE, then rounded: 1.0f - 0.999f is 1.3e-5 off 0.001 (torch forms the weights from Python doubles)
        if (b > 0) adam_multi_kernel<<<b, 256, 0, st>>>(J, (float)step_size, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)bc2_sqrt);
    }
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
