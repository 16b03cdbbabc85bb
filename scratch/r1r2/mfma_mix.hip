// Does other work issued between v_mfma_f32_32x32x2_f32 instructions cost matrix-pipe throughput on gfx950?
// Per MFMA: NV extra instructions of one kind (VALU / LDS read / SALU), everything pinned with asm volatile.
// Occupancy (waves per SIMD) is set with the dynamic LDS size.  hipcc --offload-arch=gfx950 -O3 mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA(acc) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y))

template <int KIND, int NV>
__device__ __forceinline__ void extra(float& t0, float& t1, float& t2, float& t3, float c, f32x4& l, const float* lp, int& s)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (KIND == 0) {
            if ((i & 3) == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t0) : "v"(c));
            if ((i & 3) == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t1) : "v"(c));
            if ((i & 3) == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t2) : "v"(c));
            if ((i & 3) == 3) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(t3) : "v"(c));
        } else if (KIND == 1) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"((unsigned)(size_t)lp) : "memory");
        } else if (KIND == 2) {
            asm volatile("s_add_u32 %0, %0, 1" : "+s"(s) :: "scc");
        } else if (KIND == 3) {          // transcendental-rate VALU (quarter rate)
            asm volatile("v_exp_f32 %0, %0" : "+v"(t0));
        }
    }
}

template <int KIND, int NV>
__global__ __launch_bounds__(256) void kmix(float* out, int nit)
{
    extern __shared__ float lds[];
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f + blockIdx.x * 1e-4f;
    float t0 = x, t1 = y, t2 = x + y, t3 = x - y; const float c = 0.999f;
    f32x4 l = {0}; int s = 0;
    lds[threadIdx.x * 4] = x;
    __syncthreads();
    const float* lp = lds + (threadIdx.x & 63) * 4;
    for (int i = 0; i < nit; ++i) {
        MFMA(a0); extra<KIND, NV>(t0, t1, t2, t3, c, l, lp, s);
        MFMA(a1); extra<KIND, NV>(t0, t1, t2, t3, c, l, lp, s);
        MFMA(a2); extra<KIND, NV>(t0, t1, t2, t3, c, l, lp, s);
        MFMA(a3); extra<KIND, NV>(t0, t1, t2, t3, c, l, lp, s);
        if (KIND == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float r = t0 + t1 + t2 + t3 + l[0] + (float)s;
    for (int q = 0; q < 16; ++q) r += a0[q] + a1[q] + a2[q] + a3[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND, int NV>
void run(const char* name, float* out, int wgs_per_cu)
{
    fflush(stdout);
    const int nit = 4000, blocks = 256 * wgs_per_cu * 2;
    const size_t lds = wgs_per_cu == 1 ? 100 * 1024 : wgs_per_cu == 2 ? 70 * 1024 : 36 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kmix<KIND, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); kmix<KIND, NV><<<blocks, 256, lds>>>(out, nit); hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = (double)blocks * 4 * nit * 4.0 * 2 * 32 * 32 * 2 / (ms * 1e-3) / 1e12;
    printf("%-10s x%d per MFMA, %d wave(s)/SIMD: %.3f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", name, NV, wgs_per_cu, ms, tf, tf / 1.573);
}

int main()
{
    float* out; hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("none", out, w);
        run<0, 1>("v_fma", out, w); run<0, 2>("v_fma", out, w); run<0, 4>("v_fma", out, w); run<0, 8>("v_fma", out, w);
        run<1, 1>("ds_read128", out, w); run<1, 2>("ds_read128", out, w);
        run<2, 2>("s_add", out, w); run<2, 8>("s_add", out, w);
        run<3, 1>("v_exp", out, w); run<3, 2>("v_exp", out, w);
    }
    return 0;
}
