// Importance (hierarchical) sampling of the per-scene fine-tuning option `--use_density_volume`:
//   sample_pdf        data/ray_utils.py:96-139   inverse-CDF sampling of a piecewise-constant pdf
//   ray_marcher_fine  data/ray_utils.py:199-224  density lookup -> alpha/weights -> sample_pdf -> sort(cat(z_samples, z_vals))
//   ray_points        o + d*z and get_ndc_coordinate (utils.py:112-146) for caller-supplied rays (fine-tuning's per-step
//                     `ray_marcher` output, train_mvs_nerf_finetuning_pl.py:147-156)
// The uniform draws `u` stay with the caller (the reference draws them with torch.rand on the device), so results are a
// deterministic function of the inputs.  One 64-lane wave per ray; per-wave arrays live in LDS.  All HBM/latency-bound
// integer+fp32 work: no MFMA.
#include "common.h"

constexpr int IMP_MAX = 512;     // max samples / bins / importance samples per ray held in LDS

// lanes of ONE wave exchange data through LDS: order the accesses (no s_barrier: waves of a workgroup are independent here)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// inclusive scan of v[0..n) in LDS (sum or product), lane-chunked: lane l owns the contiguous chunk [l*c, (l+1)*c).
// Accumulated in fp64 and rounded per element, which is what ATen's CPU cumsum/cumprod do (acc_type<float> = double
// there): with empty pdf bins (denominators ~1e-5) an fp32 scan's few-ulp differences in the cdf are amplified into
// 1e-4-level differences of the samples.
template <bool MUL>
__device__ __forceinline__ void wave_scan_inclusive(float* v, int n, int lane)
{
    const int c = (n + 63) >> 6;
    const int b = lane * c, e = b + c < n ? b + c : n;
    double run = MUL ? 1.0 : 0.0;
    for (int i = b; i < e; ++i) run = MUL ? run * (double)v[i] : run + (double)v[i];
    double tot = run;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const double o = __shfl_up(tot, d);
        if (lane >= d) tot = MUL ? tot * o : tot + o;
    }
    double pre = __shfl_up(tot, 1);
    if (lane == 0) pre = MUL ? 1.0 : 0.0;
    for (int i = b; i < e; ++i) { pre = MUL ? pre * (double)v[i] : pre + (double)v[i]; v[i] = (float)pre; }
}

__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d);
    return x;
}

// number of elements of the ascending array a[0..n) that are <= x (torch.searchsorted(..., right=True))
__device__ __forceinline__ int count_le(const float* a, int n, float x)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int count_lt(const float* a, int n, float x)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < x) lo = mid + 1; else hi = mid; }
    return lo;
}

// wts[0..nb-1) raw weights in LDS (overwritten by the cdf), bins[0..nb) in LDS, u[0..NI) global -> zs[0..NI) in LDS.
// cdf[0] = 0, cdf[k] = sum_{j<k} (w_j+1e-5)/sum  (data/ray_utils.py:99-102)
__device__ __forceinline__ void wave_sample_pdf(float* wts, float* cdf, const float* bins, int nb, const float* __restrict__ u, int NI,
                                                float* zs, int lane)
{
    float part = 0.f;
    for (int j = lane; j < nb - 1; j += 64) { const float w = wts[j] + 1e-5f; wts[j] = w; part += w; }
    const float tot = wave_sum(part);
    WAVE_SYNC();
    for (int j = lane; j < nb - 1; j += 64) cdf[j + 1] = wts[j] / tot;
    if (lane == 0) cdf[0] = 0.0f;
    WAVE_SYNC();
    wave_scan_inclusive<false>(cdf + 1, nb - 1, lane);
    WAVE_SYNC();
    for (int i = lane; i < NI; i += 64) {
        const float ui = u[i];
        const int inds = count_le(cdf, nb, ui);                                  // :126
        const int below = inds - 1 > 0 ? inds - 1 : 0;
        const int above = inds < nb - 1 ? inds : nb - 1;
        float denom = cdf[above] - cdf[below];
        if (denom < 1e-5f) denom = 1.0f;                                         // :135
        const float t = (ui - cdf[below]) / denom;
        zs[i] = bins[below] + t * (bins[above] - bins[below]);
    }
    WAVE_SYNC();
}

__global__ __launch_bounds__(256) void sample_pdf_kernel(const float* __restrict__ bins, const float* __restrict__ weights,
                                                         const float* __restrict__ u, int64_t N, int nb, int NI, float* __restrict__ out)
{
    __shared__ float sm[4][3 * IMP_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    if (n >= N) return;
    float* wts = sm[wave]; float* cdf = wts + IMP_MAX; float* b = cdf + IMP_MAX;
    for (int j = lane; j < nb; j += 64) b[j] = bins[n * nb + j];
    for (int j = lane; j < nb - 1; j += 64) wts[j] = weights[n * (nb - 1) + j];
    WAVE_SYNC();
    // zs go straight to global memory (out is not read back here)
    wave_sample_pdf(wts, cdf, b, nb, u + n * NI, NI, out + n * NI, lane);
}

extern "C" int mvsnerf_sample_pdf_fwd(const float* bins, const float* weights, const float* u, int64_t N, int n_bins, int n_importance,
                                      float* samples, void* stream)
{
    if (!bins || !weights || !u || !samples || N < 0 || n_bins < 2 || n_importance < 1) return MVSNERF_EINVAL;
    if (n_bins > IMP_MAX) return MVSNERF_EUNSUPPORTED;
    if (N == 0) return MVSNERF_OK;
    sample_pdf_kernel<<<mvs_cdiv(N, 4), 256, 0, (hipStream_t)stream>>>(bins, weights, u, N, n_bins, n_importance, samples);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// trilinear lookup of a single-channel volume exactly as index_point_feature does it (grid = coord*2-1, zeros padding,
// align_corners=True), for a coordinate that the caller has ALREADY mapped once (ray_marcher_fine hands pts_NDC*2-1 to
// index_point_feature, data/ray_utils.py:209-210 - the double transform is the reference's behaviour)
__device__ __forceinline__ float density_lookup(const float* __restrict__ vol, int D, int H, int W, float cx, float cy, float cz)
{
    const float gx = cx * 2.0f - 1.0f, gy = cy * 2.0f - 1.0f, gz = cz * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int zc = k >> 2, yc = (k >> 1) & 1, xc = k & 1;
        const float x = fx + xc, y = fy + yc, z = fz + zc;
        const float w = (xc ? ix - fx : fx + 1.0f - ix) * (yc ? iy - fy : fy + 1.0f - iy) * (zc ? iz - fz : fz + 1.0f - iz);
        if (x >= 0.0f && x <= (float)(W - 1) && y >= 0.0f && y <= (float)(H - 1) && z >= 0.0f && z <= (float)(D - 1))
            acc += vol[(((int64_t)z * H + (int)y) * W + (int)x)] * w;
    }
    return acc;
}

__global__ __launch_bounds__(256) void ray_marcher_fine_kernel(const float* __restrict__ density, int D, int H, int W,
                                                               const float* __restrict__ ndc, const float* __restrict__ z_vals,
                                                               const float* __restrict__ u, int64_t N, int S, int NI,
                                                               float* __restrict__ z_out)
{
    __shared__ float sm[4][5 * IMP_MAX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + wave;
    if (n >= N) return;
    float* w = sm[wave];              // alpha / weights, then the raw pdf weights
    float* t = w + IMP_MAX;           // 1-alpha+1e-10 / transmittance, then the cdf
    float* z = t + IMP_MAX;           // coarse depths
    float* bins = z + IMP_MAX;        // interval mid points
    float* zs = bins + IMP_MAX;       // importance samples (unsorted, then sorted)
    for (int s = lane; s < S; s += 64) {
        const float* c = ndc + (n * S + s) * 3;
        const float sigma = density_lookup(density, D, H, W, c[0] * 2.0f - 1.0f, c[1] * 2.0f - 1.0f, c[2] * 2.0f - 1.0f);
        const float a = 1.0f - expf(-fmaxf(sigma, 0.0f));                       // :212
        w[s] = a;
        t[s] = (1.0f - a) + 1e-10f;
        z[s] = z_vals[n * S + s];
    }
    WAVE_SYNC();
    wave_scan_inclusive<true>(t, S, lane);                                      // inclusive product; exclusive = shifted by one
    WAVE_SYNC();
    // weights[s] = alpha[s] * T_excl[s];  pdf weights = weights[1:-1]  (:213-217), staged in `zs` (consumed by sample_pdf
    // before it writes the samples there)
    for (int j = lane; j < S - 2; j += 64) zs[j] = w[j + 1] * t[j];              // T_excl[j+1] = T_incl[j]
    for (int j = lane; j < S - 1; j += 64) bins[j] = 0.5f * (z[j] + z[j + 1]);  // :216
    WAVE_SYNC();
    wave_sample_pdf(zs, t, bins, S - 1, u + n * NI, NI, zs, lane);
    // sort the NI samples by rank (ties broken by index), into `bins` (free now)
    float* zsort = bins;
    for (int i = lane; i < NI; i += 64) {
        const float v = zs[i];
        int r = 0;
        for (int j = 0; j < NI; ++j) { const float o = zs[j]; r += (o < v) || (o == v && j < i); }
        zsort[r] = v;
    }
    WAVE_SYNC();
    // merge with the (ascending) coarse depths: a sample lands after the coarse depths <= it, a coarse depth after the
    // samples < it => unique positions, ascending output = torch.sort(cat(z_samples, z_vals))  (:219)
    float* o = z_out + n * (S + NI);
    for (int i = lane; i < NI; i += 64) o[i + count_le(z, S, zsort[i])] = zsort[i];
    for (int k = lane; k < S; k += 64) o[k + count_lt(zsort, NI, z[k])] = z[k];
}

extern "C" int mvsnerf_ray_marcher_fine_fwd(const float* density, int D, int H, int W, const float* ndc, const float* z_vals, const float* u,
                                            int64_t N, int S, int n_importance, float* z_out, void* stream)
{
    if (!density || !ndc || !z_vals || !u || !z_out || D < 1 || H < 1 || W < 1 || N < 0 || S < 3 || n_importance < 1) return MVSNERF_EINVAL;
    if (S > IMP_MAX || n_importance > IMP_MAX) return MVSNERF_EUNSUPPORTED;
    if (N == 0) return MVSNERF_OK;
    ray_marcher_fine_kernel<<<mvs_cdiv(N, 4), 256, 0, (hipStream_t)stream>>>(density, D, H, W, ndc, z_vals, u, N, S, n_importance, z_out);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// pts = o + d*z and their reference-view NDC coordinates (get_ndc_coordinate, utils.py:112-146), one thread per sample.
// rays_o: [N][3], or [1][3] broadcast when o_stride == 0.
__global__ __launch_bounds__(256) void ray_points_kernel(const float* __restrict__ rays_o, int o_stride, const float* __restrict__ rays_d,
                                                         const float* __restrict__ z_vals, const float* __restrict__ w2c, const float* __restrict__ Kr,
                                                         const float* __restrict__ nf_ref, int W_ref, int H_ref, int pad, int lindisp,
                                                         int64_t N, int S, float* __restrict__ pts, float* __restrict__ ndc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * S) return;
    const int64_t n = t / S;
    const float z = z_vals[t];
    const float* o = rays_o + n * o_stride;
    const float* d = rays_d + n * 3;
    const float px = o[0] + d[0] * z, py = o[1] + d[1] * z, pz = o[2] + d[2] * z;      // o + d*z
    pts[t * 3 + 0] = px; pts[t * 3 + 1] = py; pts[t * 3 + 2] = pz;
    if (!ndc) return;
    const float near_ref = nf_ref[0], far_ref = nf_ref[1];
    const float cx = fmaf(pz, w2c[2], fmaf(py, w2c[1], px * w2c[0])) + w2c[3];
    const float cy = fmaf(pz, w2c[6], fmaf(py, w2c[5], px * w2c[4])) + w2c[7];
    const float cz = fmaf(pz, w2c[10], fmaf(py, w2c[9], px * w2c[8])) + w2c[11];
    const float qx = fmaf(cz, Kr[2], fmaf(cy, Kr[1], cx * Kr[0]));
    const float qy = fmaf(cz, Kr[5], fmaf(cy, Kr[4], cx * Kr[3]));
    const float qz = fmaf(cz, Kr[8], fmaf(cy, Kr[7], cx * Kr[6]));
    float nx = (qx / qz + 0.0f) / (float)(W_ref - 1);
    float ny = (qy / qz + 0.0f) / (float)(H_ref - 1);
    const float nz = lindisp ? (1.0f / qz - 1.0f / near_ref) / (1.0f / far_ref - 1.0f / near_ref) : (qz - near_ref) / (far_ref - near_ref);
    if (pad > 0) {
        const float Wf = (float)W_ref / 4.0f, Hf = (float)H_ref / 4.0f;               // (inv_scale+1)/4
        ny = ny * Hf / (Hf + (float)(pad * 2)) + (float)pad / (Hf + (float)(pad * 2));
        nx = nx * Wf / (Wf + (float)(pad * 2)) + (float)pad / (Wf + (float)(pad * 2));
    }
    ndc[t * 3 + 0] = nx; ndc[t * 3 + 1] = ny; ndc[t * 3 + 2] = nz;
}

extern "C" int mvsnerf_ray_points_fwd(const float* rays_o, int o_is_per_ray, const float* rays_d, const float* z_vals,
                                      const float* w2c_ref, const float* K_ref, const float* near_far_ref, int W_ref, int H_ref, int pad, int lindisp,
                                      int64_t N, int S, float* rays_pts, float* rays_ndc, void* stream)
{
    if (!rays_o || !rays_d || !z_vals || !rays_pts || N < 0 || S < 1) return MVSNERF_EINVAL;
    if (rays_ndc && (!w2c_ref || !K_ref || !near_far_ref || W_ref < 2 || H_ref < 2 || pad < 0)) return MVSNERF_EINVAL;
    if (N == 0) return MVSNERF_OK;
    ray_points_kernel<<<mvs_cdiv(N * S, 256), 256, 0, (hipStream_t)stream>>>(rays_o, o_is_per_ray ? 3 : 0, rays_d, z_vals, w2c_ref, K_ref, near_far_ref,
                                                                            W_ref, H_ref, pad, lindisp, N, S, rays_pts, rays_ndc);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
