// Minimal victim for the packed-fp32 / foreign-16-bit-MFMA interaction (profiles/r04_pk_mfma_hazard.txt, r05_psw_uninit_probe.txt): single-wave workgroups, each lane runs the
// SAME recurrence twice - once as v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on a register pair, once as two scalar v_fma_f32 / v_mul_f32 / v_add_f32 - and counts the
// iterations after which the two results differ in any bit.  No LDS, no memory traffic inside the loop.  On a quiet GPU the count is 0 by construction.
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>   // 0: fma, 1: mul + add, 2: mul op_sel_hi:[1,0], 3: fma op_sel_hi:[1,0,1], 4: fma op_sel:[0,1,0], 5: mul neg + add neg (the operand-select forms of the packed sweep build)
__global__ __launch_bounds__(64) void pk_probe_kernel(int iters, unsigned* __restrict__ bad_per_lane, unsigned long long* __restrict__ bad_total)
{
    const int lane = threadIdx.x;
    const float seed = 1.0f + 1e-3f * (float)((blockIdx.x * 64 + lane) % 977);
    f32x2 a = {0.99990f + 1e-6f * lane, 0.99985f - 1e-6f * lane}, b = {seed * 1e-3f, seed * 2e-3f};
    f32x2 p = {seed, -seed};                 // packed path
    float s0 = seed, s1 = -seed;             // scalar path
    unsigned bad = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(p) : "v"(p), "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(s0), "v"(a.x), "v"(b.x));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(s1), "v"(a.y), "v"(b.y));
            } else if (MODE == 2) {
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(p), "v"(a));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3" : "=&v"(s0) : "v"(s0), "v"(a.x), "v"(b.x));
                asm volatile("v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3" : "=&v"(s1) : "v"(s1), "v"(a.x), "v"(b.y));
            } else if (MODE == 3) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(p) : "v"(p), "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(s0), "v"(a.x), "v"(b.x));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(s1), "v"(a.x), "v"(b.y));
            } else if (MODE == 4) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(p) : "v"(p), "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(s0), "v"(a.y), "v"(b.x));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(s1), "v"(a.y), "v"(b.y));
            } else if (MODE == 6) {          // low result takes the HIGH half of src0 (the accumulator itself)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(p) : "v"(p), "v"(a), "v"(b));
                { float t0, t1;
                  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t0) : "v"(s1), "v"(a.x), "v"(b.x));
                  asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t1) : "v"(s1), "v"(a.y), "v"(b.y));
                  s0 = t0; s1 = t1; }
            } else if (MODE == 7) {          // low result takes the HIGH half of src2 (the addend)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(p) : "v"(p), "v"(a), "v"(b));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(s0), "v"(a.x), "v"(b.y));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(s1), "v"(a.y), "v"(b.y));
            } else if (MODE == 8) {          // the same select on a multiply and on an add
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(p), "v"(a));
                asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(p) : "v"(p), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3" : "=&v"(s0) : "v"(s0), "v"(a.y), "v"(b.y));
                asm volatile("v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3" : "=&v"(s1) : "v"(s1), "v"(a.y), "v"(b.y));
            } else if (MODE == 5) {
                asm volatile("v_pk_mul_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p) : "v"(p), "v"(a));
                asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(p) : "v"(p), "v"(b));
                asm volatile("v_mul_f32 %0, %1, -%2\n\tv_sub_f32 %0, %0, %3" : "=&v"(s0) : "v"(s0), "v"(a.x), "v"(b.x));
                asm volatile("v_mul_f32 %0, %1, -%2\n\tv_sub_f32 %0, %0, %3" : "=&v"(s1) : "v"(s1), "v"(a.y), "v"(b.y));
            } else {
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(a));
                asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p) : "v"(p), "v"(b));
                asm volatile("v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3" : "=&v"(s0) : "v"(s0), "v"(a.x), "v"(b.x));
                asm volatile("v_mul_f32 %0, %1, %2\n\tv_add_f32 %0, %0, %3" : "=&v"(s1) : "v"(s1), "v"(a.y), "v"(b.y));
            }
        }
        if (__float_as_uint(p.x) != __float_as_uint(s0) || __float_as_uint(p.y) != __float_as_uint(s1)) { ++bad; p.x = s0; p.y = s1; }     // count and re-synchronise
    }
    if (bad) { atomicAdd(bad_per_lane + lane, bad); atomicAdd(bad_total, (unsigned long long)bad); }
}

extern "C" int pk_probe_launch(int mode, int n_wg, int iters, unsigned* bad_per_lane, unsigned long long* bad_total, void* stream)
{
    switch (mode) {
        case 0: pk_probe_kernel<0><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
        case 1: pk_probe_kernel<1><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
        case 2: pk_probe_kernel<2><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
        case 3: pk_probe_kernel<3><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
        case 4: pk_probe_kernel<4><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
        case 5: pk_probe_kernel<5><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
        case 6: pk_probe_kernel<6><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
        case 7: pk_probe_kernel<7><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
        default: pk_probe_kernel<8><<<n_wg, 64, 0, (hipStream_t)stream>>>(iters, bad_per_lane, bad_total); break;
    }
    return (int)hipGetLastError();
}
