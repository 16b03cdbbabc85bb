// WHERE on the chip does a `v_pk_fma_f32 ... op_sel:[0,1,0]` go wrong next to a wave spinning on v_mfma_f32_16x16x32_f16?  The victim of pk_probe.hip (mode 4) and the aggressor of
// pk_hog.hip, each recording its place: key = XCC_ID << 12 | HW_ID[15:4] (se, sh, cu, pipe, simd).  victim: runs[key] += 1, bad[key] += mismatching iterations;  hog: hog_at[key] += 1.
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned where_am_i()
{
    const unsigned hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);      // HW_REG_HW_ID, bits 15:0
    const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);    // HW_REG_XCC_ID, bits 3:0
    return (xcc & 15) << 12 | ((hw >> 4) & 0xfff);
}

__global__ __launch_bounds__(64) void victim_kernel(int iters, unsigned* __restrict__ runs, unsigned* __restrict__ bad_at)
{
    const int lane = threadIdx.x;
    const float seed = 1.0f + 1e-3f * (float)((blockIdx.x * 64 + lane) % 977);
    f32x2 a = {0.99990f + 1e-6f * lane, 0.99985f - 1e-6f * lane}, b = {seed * 1e-3f, seed * 2e-3f};
    f32x2 p = {seed, -seed};
    float s0 = seed, s1 = -seed;
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(p) : "v"(p), "v"(a), "v"(b));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(s0), "v"(a.y), "v"(b.x));
            asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(s1), "v"(a.y), "v"(b.y));
        }
        bad += (__float_as_uint(p.x) != __float_as_uint(s0)) | (__float_as_uint(p.y) != __float_as_uint(s1));
        p.x = s0; p.y = s1;
    }
    const unsigned key = where_am_i();
    if (lane == 0) atomicAdd(runs + key, 1u);
    if (bad) atomicAdd(bad_at + key, bad);
}

__global__ __launch_bounds__(64) void hog_kernel(int iters, unsigned* __restrict__ hog_at, float* __restrict__ sink)
{
    f32x4 acc = {0, 0, 0, 0};
    f16x8 aa = {1, 1, 1, 1, 1, 1, 1, 1};
    if (threadIdx.x == 0) atomicAdd(hog_at + where_am_i(), 1u);
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aa, aa, acc, 0, 0, 0);
    if (acc[0] == -1.0f) sink[0] = acc[1];
}

extern "C" int pk_where_victim(int n_wg, int iters, unsigned* runs, unsigned* bad_at, void* stream)
{
    victim_kernel<<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, runs, bad_at);
    return (int)hipGetLastError();
}
extern "C" int pk_where_hog(int n_wg, int iters, unsigned* hog_at, float* sink, void* stream)
{
    hog_kernel<<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, hog_at, sink);
    return (int)hipGetLastError();
}
