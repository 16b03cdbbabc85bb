// Device functions shared by the gather kernels (sample.hip): the arithmetic
// of the trilinear volume lookup, the per-view colour lookup and the view-direction feature lives here ONCE, so that every kernel that
// produces `input_feat` produces the same bits.
#pragma once
#include "common.h"

// A 16-byte block of zeros: out-of-volume taps read it instead of branching around the load (zeros padding, exactly).
static __device__ const f32x4 g_zero_tap = {0.0f, 0.0f, 0.0f, 0.0f};

// 16-byte load through the GLOBAL address space.  A pointer chosen between a kernel argument and &g_zero_tap reaches the load as a generic
// pointer and becomes flat_load_dwordx4 (aperture check, counts against lgkmcnt as well as vmcnt); both targets are global memory.
__device__ __forceinline__ f32x4 ldg16(const float* p)
{
    typedef const f32x4 __attribute__((address_space(1))) * gptr;
    return *(gptr)p;
}

// float offset of voxel (z,y,x)'s 8-channel vector.  SMALL: 32-bit arithmetic on full-rate 24-bit multiplies (the launcher
// checks D*H < 2^24, W < 2^24, D*H*W*8 < 2^31); v_mul_lo_u32 / v_mad_u64_u32 are quarter-rate and were 40 % of the lookup's
// instruction slots.
template <bool SMALL>
__device__ __forceinline__ int64_t vox_off8(int z, int y, int x, int H, int W)
{
    if constexpr (SMALL) return (int64_t)((__umul24(__umul24(z, H) + y, W) + x) << 3);
    else return (((int64_t)z * H + y) * W + x) << 3;
}

// swap with the lane two places away inside the quad (lanes 0<->2, 1<->3): what __shfl_xor(v, 2) returns, as one DPP move
__device__ __forceinline__ float quad_swap2(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4e, 0xf, 0xf, false));   // quad_perm:[2,3,0,1]
}

// ---------------------------------------------------------------------------------------------
// Shared arithmetic of the colour lookup and the direction feature.  Written with explicit fmaf/mul/add and fp
// contraction OFF so that the stand-alone kernels and the fused gather kernel produce the same bits whatever the
// surrounding code looks like to the optimiser.
// ---------------------------------------------------------------------------------------------
struct ColorTap { float gx, gy, wnw, wne, wsw, wse; int x0, y0; bool x1in, y1in; };

__device__ __forceinline__ ColorTap color_project(float x, float y, float z, const float* __restrict__ M, const float* __restrict__ K, int W, int H)
{
#pragma clang fp contract(off)
    ColorTap t;
    // get_ndc_coordinate utils.py:124: p_cam = pts @ R^T + T   (k-ordered fma chain like sgemm)
    const float cx = fmaf(z, M[2],  fmaf(y, M[1], x * M[0]))  + M[3];
    const float cy = fmaf(z, M[6],  fmaf(y, M[5], x * M[4]))  + M[7];
    const float cz = fmaf(z, M[10], fmaf(y, M[9], x * M[8]))  + M[11];
    // :128  q = p_cam @ K^T
    const float qx = fmaf(cz, K[2], fmaf(cy, K[1], cx * K[0]));
    const float qy = fmaf(cz, K[5], fmaf(cy, K[4], cx * K[3]));
    const float qz = fmaf(cz, K[8], fmaf(cy, K[7], cx * K[6]));
    // :129  /z, / inv_scale ; utils.py:317  grid = xy*2-1
    t.gx = ((qx / qz + 0.0f) / (float)(W - 1)) * 2.0f - 1.0f;
    t.gy = ((qy / qz + 0.0f) / (float)(H - 1)) * 2.0f - 1.0f;
    // grid_sample bilinear, border padding, align_corners=True
    float ix = ((t.gx + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((t.gy + 1.0f) / 2.0f) * (float)(H - 1);
    ix = fminf(fmaxf(ix, 0.0f), (float)(W - 1));   // clip_coordinates (NaN -> 0 like ATen's min/max order)
    iy = fminf(fmaxf(iy, 0.0f), (float)(H - 1));
    if (!(ix == ix)) ix = 0.0f;
    if (!(iy == iy)) iy = 0.0f;
    const float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx; t.y0 = (int)fy;
    const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
    t.x1in = t.x0 + 1 <= W - 1; t.y1in = t.y0 + 1 <= H - 1;
    t.wnw = wx0 * wy0; t.wne = wx1 * wy0; t.wsw = wx0 * wy1; t.wse = wx1 * wy1;
    return t;
}

__device__ __forceinline__ float color_blend(const ColorTap& t, float nw, float ne, float sw, float se)
{
#pragma clang fp contract(off)
    float acc = nw * t.wnw;
    if (t.x1in) acc = fmaf(ne, t.wne, acc);
    if (t.y1in) acc = fmaf(sw, t.wsw, acc);
    if (t.x1in && t.y1in) acc = fmaf(se, t.wse, acc);
    return acc;
}

__device__ __forceinline__ float color_mask(const ColorTap& t) { return (t.gx > -1.0f && t.gx < 1.0f && t.gy > -1.0f && t.gy < 1.0f) ? 1.0f : 0.0f; }

// dirs = normalise(d) @ R^T  (renderer.py:142-147, 111-122); R == null: no rotation
__device__ __forceinline__ void dir_feature_of(const float* __restrict__ d3, const float* __restrict__ R, int normalize, float* __restrict__ o3)
{
#pragma clang fp contract(off)
    const float dx = d3[0], dy = d3[1], dz = d3[2];
    const float nrm = normalize ? sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx))) : 1.0f;   // torch.norm
    const float ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
    if (R) {
        o3[0] = fmaf(uz, R[2],  fmaf(uy, R[1], ux * R[0]));
        o3[1] = fmaf(uz, R[6],  fmaf(uy, R[5], ux * R[4]));
        o3[2] = fmaf(uz, R[10], fmaf(uy, R[9], ux * R[8]));
    } else {
        o3[0] = ux; o3[1] = uy; o3[2] = uz;
    }
}


// Fold of the eight trilinear terms in the arithmetic of ATen's 5-D grid_sample on the CPU (the reference path the oracle runs; pinned bit for
// bit by scratch/keep/cpu_lookup_probe.py against the reference-generated fixtures): every term v * w is ROUNDED (no fma), w = (wx * wy) * wz,
// and the terms are added one after the other in the order (z0,y0,x0), (z0,y0,x1), (z0,y1,x0), (z0,y1,x1), (z1,...) starting from 0.
// Lane layout of the gather kernels: a lane holds the four (z, y) products of ONE x corner (k = 2 zc + yc), its partner two lanes further
// in the quad holds the other corner's.  The x0 lane adds its own product, then the partner's (one DPP quad swap each); the x1 lane runs the
// same instructions on swapped roles and its sum is discarded by the caller (it would be the x1-first order).
__device__ __forceinline__ f32x4 trilinear_fold_x0_lane(const f32x4 (&v)[4], const float (&w)[4])
{
#pragma clang fp contract(off)
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const f32x4 own = v[k] * w[k];
        f32x4 other;
#pragma unroll
        for (int c = 0; c < 4; ++c) other[c] = quad_swap2(own[c]);
        acc = acc + own;
        acc = acc + other;
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// Depth-fastest volumes (MVSNERF_VOL_HWDC, vol[y][x][d][8]).  The samples of a ray walk DEPTH: with depth as the fastest voxel index the
// sixteen consecutive samples a wave holds read, per (y, x) column, ONE contiguous run of ~17 voxels (544 B, five 128-byte lines) instead of
// 34 separate (z, y) rows 1.2 MB apart of which 64 B each are used.
// Lane q of a sample's quad owns (yc = q >> 1, channel half h = q & 1): for both x columns of its row y it loads its 16-byte half of the z0
// and the z0 + 1 voxel - the two lanes of a row cover the 64 contiguous bytes of each column.  The fold reproduces ATen's term order
// (z0,y0,x0), (z0,y0,x1), (z0,y1,x0), (z0,y1,x1), (z1,...) - every product rounded, added one after the other from 0 - in the y0 lanes: own
// (z, x0), own (z, x1), then the y1 partner's two (one DPP quad swap each), per depth plane.  Same instruction count as the DHWC fold
// (16 products, 16 swaps, 32 additions; the lookups are bound by the instruction stream of the eight waves a SIMD holds as much as by
// memory: a first version with one lane per column and quad broadcasts - 64 DPP additions - made the fused gather 14 % SLOWER).
// Bit-identical to the DHWC kernels (tests/test_gpu_layout.py).
// ---------------------------------------------------------------------------------------------
template <bool SMALL>
__device__ __forceinline__ int64_t vox_off8_zfast(int z, int y, int x, int D, int W)
{
    if constexpr (SMALL) return (int64_t)((__umul24(__umul24(y, W) + x, D) + z) << 3);     // launcher: H*W < 2^24, D < 2^24, D*H*W*8 < 2^31
    else return ((((int64_t)y * W + x) * D + z) << 3);
}

// taps of lane q in ATen's order within its row: k = 2 zc + xc -> v[k] = this lane's channel half of voxel (z0 + zc, y, x0 + xc), w[k] its weight
struct ZfastTaps { f32x4 v[4]; float w[4]; };

template <bool SMALL>
__device__ __forceinline__ ZfastTaps zfast_taps(const float* __restrict__ vol, int D, int H, int W, float nx, float ny, float nz, int q)
{
#pragma clang fp contract(off)
    ZfastTaps t;
    // same op order as the reference: grid = ndc*2-1 (utils.py:381); unnormalise ((g+1)/2)*(size-1)
    const float gx = nx * 2.0f - 1.0f, gy = ny * 2.0f - 1.0f, gz = nz * 2.0f - 1.0f;
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
    const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
    const float iz = ((gz + 1.0f) / 2.0f) * (float)(D - 1);
    const float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    const int yc = q >> 1, ch = (q & 1) * 4;
    // weights as ATen forms them: (x1-ix) for the low corner, (ix-x0) for the high one; w = (wx * wy) * wz
    const float wy = yc ? (iy - fy) : ((fy + 1.0f) - iy);
    const float cyf = fy + (float)yc;
    // NaN / huge coordinates: the float compares reject them before any int conversion is used
    const bool y_in = (cyf >= 0.0f) && (cyf <= (float)(H - 1));
    const float* zt = reinterpret_cast<const float*>(&g_zero_tap);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int zc = k >> 1, xc = k & 1;
        const float cxf = fx + (float)xc, czf = fz + (float)zc;
        const bool in = y_in && (cxf >= 0.0f) && (cxf <= (float)(W - 1)) && (czf >= 0.0f) && (czf <= (float)(D - 1));
        t.w[k] = ((xc ? (ix - fx) : ((fx + 1.0f) - ix)) * wy) * (zc ? (iz - fz) : ((fz + 1.0f) - iz));
        t.v[k] = ldg16(in ? vol + vox_off8_zfast<SMALL>((int)czf, (int)cyf, (int)cxf, D, W) + ch : zt);
    }
    return t;
}

// the four channel sums of this lane's half; valid in the y0 lanes (q < 2)
__device__ __forceinline__ f32x4 zfast_fold_y0_lane(const ZfastTaps& t)
{
#pragma clang fp contract(off)
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int z = 0; z < 2; ++z) {
        const f32x4 a = t.v[2 * z] * t.w[2 * z], b = t.v[2 * z + 1] * t.w[2 * z + 1];     // every term v * w rounded (no fma)
        f32x4 pa, pb;
#pragma unroll
        for (int c = 0; c < 4; ++c) { pa[c] = quad_swap2(a[c]); pb[c] = quad_swap2(b[c]); }
        acc = acc + a;                  // (z, y0, x0)
        acc = acc + b;                  // (z, y0, x1)
        acc = acc + pa;                 // (z, y1, x0): the partner two lanes further in the quad
        acc = acc + pb;                 // (z, y1, x1)
    }
    return acc;
}
