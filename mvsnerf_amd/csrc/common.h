// Shared helpers for the gfx950 kernels of libmvsnerf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvsnerf_hip_internal.h"      // (includes the stable tier, mvsnerf_hip.h)

#define MVS_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return (int)e__;             \
    } while (0)

static inline bool mvs_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline unsigned mvs_cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// Raising a kernel's dynamic-LDS cap is a per-DEVICE attribute: remember it per device (bit d of *done_mask), so that a process that
// touches a second GPU sets it there too.  Idempotent; the mask is the only state and it caches nothing but "already raised here".
static inline int mvs_raise_lds_cap(const void* fn, int bytes, unsigned long long* done_mask)
{
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return (int)e;
    const unsigned long long bit = 1ull << (d & 63);
    if (__atomic_load_n(done_mask, __ATOMIC_RELAXED) & bit) return 0;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELAXED);
    return 0;
}

// smallest multiple of 4 that is >= n and whose quarter is odd: an LDS row stride (in floats) that walks all 16 bank groups of 16 bytes
__host__ __device__ inline int mvs_odd_quad_stride(int n) { const int q = (n + 3) >> 2; return (q | 1) << 2; }

// InPlaceABN partial sums left by their producers (abn_partial_kernel, or a convolution that sums its own output tile):
// part[{sum, sum of squares}][channel][slot], slot < nslots (one slot per workgroup or M-tile).  Channel-major, so that the finalize
// workgroup of a channel reads two contiguous runs.  (Slot-major rows - part[slot][2][C] - made every wave-level load of the finalize
// touch 64 different lines: 11.6 us per finalize at 9 k tiles, 18 finalizes per scene encode = 8 % of it.)
__host__ __device__ inline int64_t abn_part_at(int which, int c, int C, int64_t slot, int64_t nslots) { return ((int64_t)which * C + c) * nslots + slot; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// Linear voxel / pixel index -> (x, y, z) of a [.][H][W] grid.  A 64-bit division is ~10x the instructions of a 32-bit one and a lane of the direct-load
// implicit-GEMM kernels did three of them before its first load (half of the VALU instructions of the deep CostRegNet layers): indices below 2^32 - every
// tensor of this network - take the 32-bit path.
__device__ __forceinline__ void mvs_unflatten3(int64_t v, int W, int H, int& x, int& y, int& z)
{
    if (((uint64_t)v >> 32) == 0) {
        const unsigned u = (unsigned)v, r = u / (unsigned)W;
        x = (int)(u - r * (unsigned)W);
        const unsigned q = r / (unsigned)H;
        y = (int)(r - q * (unsigned)H);
        z = (int)q;
    } else {
        x = (int)(v % W); y = (int)((v / W) % H); z = (int)(v / ((int64_t)W * H));
    }
}
// (x0 + off, y0, z0) carried into the grid for a SMALL off >= 0 (x0 + off < 2^20, W < 2^12): quotients by a float reciprocal, exact in that range
// ((n + 0.5) / W is never closer than 0.5 / W to an integer; the product's rounding error is below that while n < ~2.8e6)
__device__ __forceinline__ void mvs_carry3(int x0, int y0, int z0, int off, int W, int H, float rW, float rH, int& x, int& y, int& z)
{
    const int xs = x0 + off;
    const int qx = (int)(((float)xs + 0.5f) * rW);
    x = xs - qx * W;
    const int ys = y0 + qx;
    const int qy = (int)(((float)ys + 0.5f) * rH);
    y = ys - qy * H;
    z = z0 + qy;
}

// Workgroups are dealt to the 8 XCDs round-robin by their linear id (workgroup i runs on XCD i % 8) and every XCD has its own L2.
// Kernels whose neighbouring workgroups share input (convolution halos) renumber their tiles so that an XCD walks a CONTIGUOUS
// range of tile ids: tile = xcd_contiguous_tile(blockIdx.x, gridDim.x).
__device__ __forceinline__ int xcd_contiguous_tile(int bid, int nb)
{
    const int per = nb >> 3, rem = nb & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
}
