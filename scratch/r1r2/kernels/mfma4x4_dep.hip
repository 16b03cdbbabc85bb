// v_mfma_f32_4x4x1_16B_f32: issue rate against the number of accumulators cycled through (dependent-issue latency)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void rate(float* out, int nit)
{
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 0.5f;
    for (int i = 0; i < nit; ++i) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(float* out)
{
    const int nit = 4000, blocks = 1024;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); rate<NACC><<<blocks, 256>>>(out, nit); hipEventRecord(e1); hipEventSynchronize(e1); }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("1 wave/SIMD x 4 (1024 WGs of 256): %d accumulators: %.3f ms -> %.1f TFLOP/s\n", NACC, ms, (double)blocks * 4 * nit * 24 * 512 / (ms * 1e-3) / 1e12);
}
int main()
{
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    run<1>(out); run<2>(out); run<3>(out); run<4>(out); run<6>(out); run<8>(out);
    return 0;
}
