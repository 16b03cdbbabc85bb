#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// every wave: NIT x 4 back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators; records shader cycles (s_memtime)
// and the 100 MHz constant clock (wall_clock64) around the loop
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, long long* wall, int nit)
{
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-4f;
    const long long c0 = __builtin_amdgcn_s_memtime(), w0 = wall_clock64();
    for (int i = 0; i < nit; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
    const long long c1 = __builtin_amdgcn_s_memtime(), w1 = wall_clock64();
    float s = 0; for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; wall[blockIdx.x] = w1 - w0; }
}
int main() {
    int wcr = 0; hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("wall clock rate %d kHz, max shader clock %d kHz\n", wcr, clk);
    const int nit = 20000;
    for (int blocks : {1, 256, 512, 2048}) {
      for (int rep = 0; rep < 2; ++rep) {
        float* out; long long *cyc, *wall;
        hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&wall, blocks * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k<<<blocks, 256>>>(out, cyc, wall, nit);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> c(blocks), w(blocks);
        hipMemcpy(c.data(), cyc, blocks * 8, hipMemcpyDeviceToHost); hipMemcpy(w.data(), wall, blocks * 8, hipMemcpyDeviceToHost);
        double cm = 0, wm = 0; for (int i = 0; i < blocks; ++i) { cm += c[i]; wm += w[i]; } cm /= blocks; wm /= blocks;
        const double secs = wm / (wcr * 1e3);
        const double flops = (double)blocks * 4 /*waves*/ * nit * 4.0 * 2 * 32 * 32 * 2;
        printf("blocks %5d: kernel %.3f ms; per wave: %.0f s_memtime ticks, %.1f us wall -> %.1f ticks/MFMA, tick rate %.3f GHz, %.2f ns/MFMA; chip %.1f TFLOP/s\n",
               blocks, ms, cm, secs * 1e6, cm / (nit * 4.0), cm / secs / 1e9, secs * 1e9 / (nit * 4.0), flops / (ms * 1e-3) / 1e12);
        hipFree(out); hipFree(cyc); hipFree(wall);
      }
    }
    return 0;
}
