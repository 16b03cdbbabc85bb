// Issue rate of v_pk_fma_f32 on gfx950: all-VGPR operands, an SGPR-pair source, and the op_sel broadcast form the conv kernels use;
// plus plain v_fma_f32 for reference.  8 independent accumulator pairs per lane, 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int nit, float sw0, float sw1)
{
    f32x2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f32x2{threadIdx.x * 1e-3f + i, 1.0f};
    f32x2 x = {1.0f + threadIdx.x * 1e-6f, 0.5f};
    f32x2 wv = {sw0 + threadIdx.x * 1e-9f, sw1};
    f32x2 ws = {sw0, sw1};              // wave-uniform: lives in an SGPR pair
    for (int it = 0; it < nit; ++it) {
#define A(i) \
        if (KIND == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(wv)); \
        else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(ws)); \
        else if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(a[i]) : "v"(x), "s"(ws)); \
        else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(a[i]) : "v"(x), "s"(ws)); \
        else { asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].x) : "v"(x.x), "v"(wv.x)); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i].y) : "v"(x.y), "v"(wv.y)); }
        REP8(A) REP8(A)
#undef A
    }
    float r = 0; for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int KIND> void run(const char* name, float* out)
{
    const int blocks = 256 * 8, nit = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0); k<KIND><<<blocks, 256>>>(out, nit, 0.999f, 1.001f); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    }
    const double flops = (double)blocks * 256 * nit * 16 * 4;     // 16 packed instructions (or 32 scalar ones) x 4 flops per lane
    printf("%-44s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12); fflush(stdout);
}
int main()
{
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0>("v_pk_fma_f32 v, v, v", out);
    run<1>("v_pk_fma_f32 v, v, s[pair]", out);
    run<2>("v_pk_fma_f32 v, v, s[pair] op_sel_hi:[0,1,1]", out);
    run<3>("v_pk_fma_f32 v, v, s[pair] op_sel:[1,0,0]", out);
    run<4>("2 x v_fma_f32 v, v, v", out);
    return 0;
}
