// Probe of v_mfma_f32_4x4x1_16B_f32 on gfx950: (1) operand / result lane layout, (2) issue rate alone and with one ds_read_b128 per 4 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out)
{
    const int l = threadIdx.x;
    // A[block][i] = 100*block + 10*(i+1) ; B[block][j] = (j+1) + 0.01*block : D[block][i][j] = A*B
    const float a = 100.f * (l / 4) + 10.f * (l % 4 + 1), b = (l % 4 + 1);
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}

template <int LDS>
__global__ __launch_bounds__(256) void rate(float* out, int nit)
{
    __shared__ __attribute__((aligned(16))) float sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = i * 1e-4f;
    __syncthreads();
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
    f32x4 a[8];
    for (int t = 0; t < 8; ++t) a[t] = f32x4{threadIdx.x * 1e-3f, 1.f, 2.f, 3.f};
    f32x4 b = {1.f, 0.5f, 0.25f, 0.125f};
    const float* p = sm + (threadIdx.x & 63) * 12;
    for (int i = 0; i < nit; ++i) {
        if (LDS) {
#pragma unroll
            for (int t = 0; t < 8; ++t) a[t] = *reinterpret_cast<const f32x4*>(p + ((i * 8 + t) & 7) * 408);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[t][k], b[k], acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    float* out; hipMalloc(&out, 2048 * 256 * 4 * 4);
    layout<<<1, 64>>>(out);
    std::vector<float> h(256);
    hipMemcpy(h.data(), out, 256 * 4, hipMemcpyDeviceToHost);
    printf("layout: lane l reg r -> value (expect A[blk][i]*B[blk][j])\n");
    for (int l : {0, 1, 2, 3, 4, 5, 63}) printf("  lane %2d: %8.1f %8.1f %8.1f %8.1f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    const int nit = 4000;
    for (int mode = 0; mode < 2; ++mode)
        for (int blocks : {256, 512, 1024, 2048}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode) rate<1><<<blocks, 256>>>(out, nit); else rate<0><<<blocks, 256>>>(out, nit);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * nit * 32.0 * 512;
            printf("mode %d (ds_read_b128 per 4 mfma: %d) blocks %4d: %.3f ms -> %.1f TFLOP/s\n", mode, mode, blocks, ms, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
