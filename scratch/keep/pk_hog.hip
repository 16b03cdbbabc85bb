// Aggressors WITHOUT matrix, LDS or memory instructions for scratch/keep/pk_probe.hip's victims: waves that only occupy resources of a SIMD while the victim kernel runs on another
// stream.  hog<NV, MODE>: NV VGPRs allocated per wave (the highest one is touched by inline asm), MODE 0: sleeps (s_sleep), 1: spins on v_add_f32, 2: spins on a 16x16x32 f16 MFMA.
// MODE 3: v_mfma_f32_32x32x16_f16, 4: v_mfma_f32_32x32x2_f32, 5: v_mfma_f32_16x16x4_f32, 6: v_mfma_f32_16x16x32_bf16, 7: v_mfma_f32_4x4x4_16B_f16, 8: v_mfma_f32_16x16x16_f16 (few VGPRs each).
// WAVES = waves per workgroup, LDS = dynamic LDS bytes requested (never touched).
#include <hip/hip_runtime.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

template <int NV, int MODE>
__global__ void hog_kernel(int iters, float* __restrict__ sink)
{
    extern __shared__ char dyn[];
    float x = (float)threadIdx.x;
    if (NV == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (NV == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (NV == 256) asm volatile("v_mov_b32 v255, 0" ::: "v255");
    f32x4 acc = {0, 0, 0, 0};
    f16x8 aa = {1, 1, 1, 1, 1, 1, 1, 1};
    f32x16 acc16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 bb = {1, 1, 1, 1, 1, 1, 1, 1};
    f16x4 a4 = {1, 1, 1, 1};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) __builtin_amdgcn_s_sleep(127);
        else if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < 256; ++u) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(1.0f));
        } else if (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(aa, aa, acc, 0, 0, 0);
        } else if (MODE == 3) {
#pragma unroll
            for (int u = 0; u < 32; ++u) acc16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(aa, aa, acc16, 0, 0, 0);
        } else if (MODE == 4) {
#pragma unroll
            for (int u = 0; u < 16; ++u) acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, acc16, 0, 0, 0);
        } else if (MODE == 5) {
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, acc, 0, 0, 0);
        } else if (MODE == 6) {
#pragma unroll
            for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, bb, acc, 0, 0, 0);
        } else if (MODE == 7) {
#pragma unroll
            for (int u = 0; u < 128; ++u) acc = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, a4, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int u = 0; u < 64; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, a4, acc, 0, 0, 0);
        }
    }
    if (x + acc[0] + acc16[0] == -1.0f) sink[0] = x;        // never true
}

extern "C" int pk_hog_launch(int nv, int mode, int n_wg, int waves, int lds_bytes, int iters, float* sink, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
#define HOG(NV_, M_) do { if (lds_bytes > 65536) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel<NV_, M_>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); if (e != hipSuccess) return (int)e; } \
        hog_kernel<NV_, M_><<<n_wg, 64 * waves, lds_bytes, st>>>(iters, sink); } while (0)
    const int key = nv * 10 + mode;
    switch (key) {
        case 640: HOG(64, 0); break;   case 641: HOG(64, 1); break;   case 642: HOG(64, 2); break;
        case 1280: HOG(128, 0); break; case 1281: HOG(128, 1); break; case 1282: HOG(128, 2); break;
        case 2560: HOG(256, 0); break; case 2561: HOG(256, 1); break; case 2562: HOG(256, 2); break;
        case 320: HOG(32, 0); break;   case 321: HOG(32, 1); break;   case 322: HOG(32, 2); break;
        case 323: HOG(32, 3); break;   case 324: HOG(32, 4); break;   case 325: HOG(32, 5); break;
        case 326: HOG(32, 6); break;   case 327: HOG(32, 7); break;   case 328: HOG(32, 8); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
