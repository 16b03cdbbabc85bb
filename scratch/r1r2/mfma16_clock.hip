#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// issue rate of v_mfma_f32_16x16x4_f32 vs v_mfma_f32_32x32x2_f32 (4 independent accumulators, 1024 workgroups of 4 waves)
__global__ __launch_bounds__(256) void k16(float* out, int nit)
{
    f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-4f;
    for (int i = 0; i < nit; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
    }
    float s = 0; for (int r = 0; r < 4; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k32(float* out, int nit)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-4f;
    for (int i = 0; i < nit; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    const int blocks = 2048, nit = 20000;
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0); k16<<<blocks, 256>>>(out, nit); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("16x16x4 f32: %.3f ms -> %.1f TFLOP/s\n", ms, (double)blocks * 4 * nit * 4.0 * 2 * 16 * 16 * 4 / (ms * 1e-3) / 1e12);
        hipEventRecord(e0); k32<<<blocks, 256>>>(out, nit); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("32x32x2 f32: %.3f ms -> %.1f TFLOP/s\n", ms, (double)blocks * 4 * nit * 4.0 * 2 * 32 * 32 * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
