// Launch-shape / cache-hint variants of volume_sample_c8_kernel for scratch/r3/vs_variants.py (not part of the library).
#include "../../mvsnerf_amd/csrc/common.h"
#include "../../mvsnerf_amd/csrc/sample_dev.h"

template <int BLOCK, bool NT>
__global__ __launch_bounds__(BLOCK) void vs_kernel(const float* __restrict__ vol, int D, int H, int W, const float* __restrict__ ndc, int64_t P,
                                                   float* __restrict__ out, int out_stride)
{
#pragma clang fp contract(off)
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = (int)(tid & 3);
    const int xc = q >> 1, ch = (q & 1) * 4;
    const int64_t p = tid >> 2;
    const int64_t pc = p < P ? p : (P - 1);
    typedef float f32x3 __attribute__((ext_vector_type(3)));
    const f32x3 nd = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x3*>(ndc + pc * 3)) : *reinterpret_cast<const f32x3*>(ndc + pc * 3);
    const float gx = nd[0] * 2.0f - 1.0f, gy = nd[1] * 2.0f - 1.0f, gz = nd[2] * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1), iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const float wx = xc ? (ix - fx) : ((fx + 1.0f) - ix);
    const float cxf = fx + (float)xc;
    const bool x_in = (cxf >= 0.0f) && (cxf <= (float)(W - 1));
    f32x4 vv[4];
    float vw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int zc = k >> 1, yc = k & 1;
        const float cyf = fy + (float)yc, czf = fz + (float)zc;
        const bool in = x_in && (cyf >= 0.0f) && (cyf <= (float)(H - 1)) && (czf >= 0.0f) && (czf <= (float)(D - 1));
        vw[k] = (wx * (yc ? (iy - fy) : ((fy + 1.0f) - iy))) * (zc ? (iz - fz) : ((fz + 1.0f) - iz));
        const float* src = in ? vol + vox_off8<true>((int)czf, (int)cyf, (int)cxf, H, W) + ch : reinterpret_cast<const float*>(&g_zero_tap);
        vv[k] = ldg16(src);
    }
    const f32x4 acc = trilinear_fold_x0_lane(vv, vw);
    if (p < P && xc == 0) {
        if (NT) __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(out + p * out_stride + ch));
        else *reinterpret_cast<f32x4*>(out + p * out_stride + ch) = acc;
    }
}

extern "C" int vs_variant(int variant, const float* vol, int D, int H, int W, const float* ndc, int64_t P, float* out, int out_stride, void* stream)
{
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
        case 0: vs_kernel<256, false><<<mvs_cdiv(P * 4, 256), 256, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride); break;
        case 1: vs_kernel<128, false><<<mvs_cdiv(P * 4, 128), 128, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride); break;
        case 2: vs_kernel<64, false><<<mvs_cdiv(P * 4, 64), 64, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride); break;
        case 3: vs_kernel<256, true><<<mvs_cdiv(P * 4, 256), 256, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride); break;
        case 4: vs_kernel<512, false><<<mvs_cdiv(P * 4, 512), 512, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride); break;
        case 5: vs_kernel<128, true><<<mvs_cdiv(P * 4, 128), 128, 0, st>>>(vol, D, H, W, ndc, P, out, out_stride); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
