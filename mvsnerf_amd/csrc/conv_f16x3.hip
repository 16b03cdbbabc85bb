// conv0 of CostRegNet (models.py:756: 3x3x3, stride 1, Cin = 32 + 3V -> 8 channels; 74.5 % of the encoder's FLOPs) with fp32-GRADE results from
// the fp16 matrix cores: the two-piece operand split of mlp_f16x3.hip applied to the encoder's largest layer.  conv0 of every no-grad encode
// (`encoder.ENCODER_PRECISION = "auto"`) as a GUARDED sequence: a cost value or weight outside fp16's range sets the guard word and the fp32
// plane sweep + fp32-MFMA conv0 of conv_mfma.hip, enqueued behind this kernel and predicated on that word, recompute the layer
// (mvsnerf_sweep_conv0_guarded_fwd, encoder.hip).  `encoder_precision("fp16x3")` is the unguarded pair (saturating); gradients take fp32 kernels.
//
//     x = x0 + x1 (+ 2^-22 |x|),  x0 = fp16(x), x1 = fp16(x - x0);   w likewise;      x * w ~= x0*w0 + x0*w1 + x1*w0      (dropped: x1*w1 <= 2^-22 |x w|)
//
// Three v_mfma_f32_16x16x32_f16 per product, exact piece products, fp32 accumulation.  Evaluated on the CPU first, at config 2 with the shipped
// weights (scratch/keep/conv0_f16x3_numerics.py, piece convolutions in float64): the raw conv0 output is 1.1e-4 from the float64 convolution (of values up
// to 2215) where the fp32 reference path (oneDNN) is 1.3e-3; the neural volume built on it is 5.7e-6 from the fp32 oracle's - what a conv0 evaluated
// EXACTLY differs from it (5.3e-6): the split is below the noise of an fp32 summation order.
//
// Range: the plane sweep stores x * 2^-4 (mvsnerf_planesweep_costvar_f16x2_fwd) and the weights are packed * 2^4 - exact, products unchanged - so that
// the cost volume may reach 2^20 before an fp16 piece saturates (variance channels of the shipped FeatureNet: < 450).  Pieces below fp16's normal range
// are subnormals, which gfx950's matrix cores take as they are; what they lose is below 2^-21 absolute per operand.
//
// Kernel = conv_bf16.hip's forward scheme (wave = plane, M-tile = a row of 16 x, K = 32 = two (dz, dx) taps x 16 channels, one input-row fragment
// serving the three dy taps, LDS-DMA tiles of [voxel][32 B]) on 8 x 8 x 16 output voxels per workgroup of eight waves, run over TWO tiles per
// 16-channel block: the hi tile against both weight planes (34 MFMAs per fragment group), then the lo tile against the hi weights (17) - the dy = 0 and
// dy = 1 taps of an input row share one MFMA through the two halves of the 16 B columns (see the accumulator comment in the kernel).
#include "common.h"
#include "act.h"
#include "lds_dma.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int W_SCALE_LOG2 = 4;                                  // weights * 2^4 <-> cost volume * 2^-4 (encoder.hip, blocked mode 3)

__device__ const f32x4 g_zero16h = {0.0f, 0.0f, 0.0f, 0.0f};     // what a DMA lane reads for a voxel outside the volume (zero padding)

__device__ __forceinline__ void dma16_gather_h(const void* lane_ptr, unsigned lds_byte_uniform)
{
    // lanes read 16 B each at their own address; LDS receives them at lds_byte_uniform + lane * 16 (M0 carries the LDS base)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(lane_ptr), "s"(lds_byte_uniform) : "memory");
}

// Workgroup = EIGHT waves = 8 x 8 x 16 output voxels (z, y, x), wave w owns plane z0 + w.  The kernel moves 2.3 GB through the LDS-DMA path per launch
// at config 2 with the 4-plane tile of conv_bf16.hip (5.7 TB/s at 407 us - the chip's LDS-DMA fill rate, MI355X_MICROARCH.md "ldsdma-fill", is what bounds
// it, not the matrix pipes: the 17-MFMA scheme below took 29 % of the MFMA work out and 10 % of the time); the 8-plane tile's halo is 1.76 x its voxels
// instead of 2.11 x and one 15 KB weight fetch serves twice the voxels: 1.78 GB.  73 KB of LDS and <= 128 VGPRs: two workgroups = 16 waves per CU.
// The InPlaceABN statistics keep the 4-plane slots of mvsnerf_conv0_bf16_tiles (one slot per half of a workgroup).
constexpr int HTX = 16, HTY = 8, HTZ = 8, H_WAVES = 8;
constexpr int HPX = HTX + 2, HPY = HTY + 2, HPZ = HTZ + 2;
constexpr int HNV = HPX * HPY * HPZ;                            // 1800 voxels
constexpr int HT_PIECES = (HNV * 32 + 1023) / 1024;             // 57 DMA pieces per tile
constexpr int HT_BYTES = HT_PIECES * 1024;
constexpr int HT_SLOTS = (HT_PIECES + H_WAVES - 1) / H_WAVES;   // 8 per wave
constexpr int HW_BYTES = 5 * 3 * 4 * 8 * 16;                    // one weight plane of a chunk: [f][dy][kg][co 8][8 ci] fp16 = 7680 B
constexpr int HW_PIECES = (HW_BYTES + 1023) / 1024;             // 8 per plane
constexpr int HBUF = HT_BYTES + 2 * HW_PIECES * 1024;           // 74752 B: two workgroups per CU

__global__ __launch_bounds__(64 * H_WAVES, 4) void conv3d_k3s1_c8_f16x3_kernel(const _Float16* __restrict__ x16, int nblk16, int D, int H, int W,
                                                                     const _Float16* __restrict__ wq, float* __restrict__ out,
                                                                     float* __restrict__ stats, int* __restrict__ guard)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nbx = (W + HTX - 1) / HTX, nby = (H + HTY - 1) / HTY;
    const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, bz = tile_id / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = bx * HTX - 1, y0 = by * HTY - 1, z0 = bz * HTZ - 1;
    const int64_t nvox = (int64_t)D * H * W;
    if (guard && blockIdx.x == 0 && tid == 0 && (float)wq[(int64_t)nblk16 * 7680] != 0.0f) guard[0] = 1;     // status word behind the weights: one was clamped at pack time
    // DMA slots of this lane: piece p = wave + 8 j holds tile voxels 32 p .. 32 p + 31, lane -> (voxel 32 p + lane / 2, half lane & 1)
    int goff[HT_SLOTS];                                          // byte offset inside a channel block, -1: zeros
#pragma unroll
    for (int j = 0; j < HT_SLOTS; ++j) {
        const int v = (wave + H_WAVES * j) * 32 + (lane >> 1);
        const int vx = v % HPX, vy = (v / HPX) % HPY, vz = v / (HPX * HPY);
        const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        const bool in = v < HNV && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        goff[j] = in ? (((gz * H + gy) * W + gx) * 32 + (lane & 1) * 16) : -1;
    }
    const char* zero16 = reinterpret_cast<const char*>(&g_zero16h);
    const unsigned base = lds_byte_addr(lds);
    // plane 0 = hi pieces x16[0 .. nblk16), plane 1 = lo pieces right behind them
    auto issue_tile = [&](int plane, int c) {
        const char* xb = reinterpret_cast<const char*>(x16) + ((int64_t)plane * nblk16 + c) * nvox * 32;
#pragma unroll
        for (int j = 0; j < HT_SLOTS; ++j) {
            const int p = wave + H_WAVES * j;
            if (p < HT_PIECES) dma16_gather_h(goff[j] >= 0 ? xb + goff[j] : zero16, base + p * 1024);
        }
    };
    auto issue_weights = [&](int c) {                            // both planes of chunk c: 16 pieces, two per wave (the tail pieces are partly padding)
        const char* wb = reinterpret_cast<const char*>(wq) + (int64_t)c * 2 * HW_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = wave + H_WAVES * j, plane = p >> 3, off = (p & 7) * 1024 + lane * 16;
            dma16_gather_h(off < HW_BYTES ? wb + plane * HW_BYTES + off : zero16, base + HT_BYTES + p * 1024);
        }
    };
    // Accumulators P[0..8], one per input row j (y = y0 + j): B columns 0..7 carry the dy = 0 weights and columns 8..15 the dy = 1 weights, so ONE
    // MFMA of input row j yields [dy 0 -> output row j | dy 1 -> output row j - 1] in the left / right halves of P[j]; a second MFMA with
    // B = [dy 2 weights | 0] adds input row j's share of output row j - 2 to the left half of P[j - 2].  Output row t = left(P[t]) + right(P[t + 1]).
    // 9 + 8 = 17 MFMAs per (fragment, weight piece) instead of the 24 of one-dy-per-MFMA with a dead right half (round 4: 66.8 % of the kernel's
    // duration the matrix pipes were busy, half of that on columns nobody stored).
    f32x4 P[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) P[t] = f32x4{0, 0, 0, 0};
    const int m = lane & 15, kg = lane >> 4;
    const bool left = m < 8;
    const char* wt01 = lds + HT_BYTES + (m >> 3) * 512 + (kg * 8 + (m & 7)) * 16;      // columns 0..7: dy 0, columns 8..15: dy 1
    const char* wt2 = lds + HT_BYTES + 2 * 512 + (kg * 8 + (m & 7)) * 16;              // dy 2 (the right half of that fragment is zeroed)
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    // one (fragment group, weight piece): 17 MFMAs; rows are visited so that no accumulator is written twice in a row
    auto mma17 = [&](const f16x8 (&av)[10], const f16x8 b01, const f16x8 b2) {
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            if (j < 9) P[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[j], b01, P[j], 0, 0, 0);
            if (j >= 2) P[j - 2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(av[j], b2, P[j - 2], 0, 0, 0);
        }
    };
#pragma unroll 1
    for (int c = 0; c < nblk16; ++c) {
        // ---- hi tile x (lo weights, then hi weights): the small products first
        issue_tile(0, c);
        issue_weights(c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMA pieces have landed ...
        __syncthreads();                                          // ... everybody's have
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const int p = 2 * f + (kg >> 1) < 9 ? 2 * f + (kg >> 1) : 8;     // the missing tenth (dz, dx) pair re-reads the ninth; its weights are zero
            const int dz = p / 3, dx = p - 3 * dz;
            const char* al = lds + (((wave + dz) * HPY) * HPX + m + dx) * 32 + (kg & 1) * 16;
            f16x8 av[10];
            const f16x8 bh01 = *reinterpret_cast<const f16x8*>(wt01 + f * 3 * 512);
            const f16x8 bl01 = *reinterpret_cast<const f16x8*>(wt01 + HW_PIECES * 1024 + f * 3 * 512);
            const f16x8 bh2 = left ? *reinterpret_cast<const f16x8*>(wt2 + f * 3 * 512) : zero8;
            const f16x8 bl2 = left ? *reinterpret_cast<const f16x8*>(wt2 + HW_PIECES * 1024 + f * 3 * 512) : zero8;
#pragma unroll
            for (int j = 0; j < 10; ++j) av[j] = *reinterpret_cast<const f16x8*>(al + j * HPX * 32);
            mma17(av, bl01, bl2);
            mma17(av, bh01, bh2);
        }
        __syncthreads();                                          // everybody is done reading the tile
        // ---- lo tile x hi weights (the weight planes of this chunk stay where they are)
        issue_tile(1, c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const int p = 2 * f + (kg >> 1) < 9 ? 2 * f + (kg >> 1) : 8;
            const int dz = p / 3, dx = p - 3 * dz;
            const char* al = lds + (((wave + dz) * HPY) * HPX + m + dx) * 32 + (kg & 1) * 16;
            f16x8 av[10];
            const f16x8 bh01 = *reinterpret_cast<const f16x8*>(wt01 + f * 3 * 512);
            const f16x8 bh2 = left ? *reinterpret_cast<const f16x8*>(wt2 + f * 3 * 512) : zero8;
#pragma unroll
            for (int j = 0; j < 10; ++j) av[j] = *reinterpret_cast<const f16x8*>(al + j * HPX * 32);
            mma17(av, bh01, bh2);
        }
        __syncthreads();                                          // everybody is done reading the buffer
    }
    // D: lane (column n = lane & 15, g = lane >> 4): register r = voxel x 4 g + r of the M-tile.  Output row t, channel n < 8:
    // left(P[t]) sits in lane n, right(P[t + 1]) in lane n + 8 of the same 16-lane row: one DPP row rotation by 8 brings it over.
    // (everything the epilogue addresses with is derived HERE from an opaque copy of the thread index: computed from the kernel's `lane` / `wave` it is hoisted
    // above the main loop, which runs at the 128-register bound of four waves per SIMD - seven registers spilled)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6;
    const int n = lane_e & 15, g4 = lane_e >> 4;
    const int oz = bz * HTZ + wave_e;
    float ssum = 0.f, ssq = 0.f;
    // Stored from the D fragment a lane writes ONE float per instruction, half the lanes (n >= 8) are masked and an instruction touches four 32-byte pieces
    // (32 such instructions per wave for 150 MB of output).  A wave's plane is 8 rows of 16 voxels x 8 channels = 512 contiguous bytes each: the values go through
    // a wave-private stage in the (free: barrier above) tile buffer - padded by 8 floats per 4 voxels, so that the four voxel groups of a write and the quarter
    // rows of a read fall on different banks - and leave as four 1 KB global_store_dwordx4 (two rows each).  One pass over the rows: row t's values are formed,
    // counted for the InPlaceABN partial sums (in the order the direct stores took them) and staged, nothing else of the fragment stays live.
    constexpr int RS = 16 * 8 + 4 * 8;                            // floats per staged row
    float* stg = reinterpret_cast<float*>(lds) + wave_e * (8 * RS) + g4 * 40 + n;
    const bool mine = oz < D && n < 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int oy = by * HTY + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = P[t][r] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(P[t + 1][r]), 0x128, 0xf, 0xf, false));   // row_ror:8
            if (n < 8) stg[t * RS + r * 8] = v;
            if (mine && bx * HTX + g4 * 4 + r < W && oy < H) { ssum += v; ssq = fmaf(v, v, ssq); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // wave-private: no barrier
    {
        const int xr = (lane_e & 31) >> 1, hq = lane_e & 1;
        const float* rd = reinterpret_cast<const float*>(lds) + wave_e * (8 * RS) + (lane_e >> 5) * RS + xr * 8 + (xr >> 2) * 8 + hq * 4;
        const int ox = bx * HTX + xr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(rd + 2 * k * RS);
            const int oy = by * HTY + 2 * k + (lane_e >> 5);
            if (oz < D && oy < H && ox < W) *reinterpret_cast<f32x4*>(out + (((int64_t)oz * H + oy) * W + ox) * 8 + hq * 4) = v4;
        }
    }
    if (stats) {          // InPlaceABN partial sums in the 4-plane slots of mvsnerf_conv0_bf16_tiles (abn_finalize_kernel's layout): waves 0..3 -> slot of the
        __syncthreads();  // planes 8 bz .. 8 bz + 3, waves 4..7 -> the slot of the next four planes (absent when D ends inside the first half)
        float* red = reinterpret_cast<float*>(lds);
        ssum += __shfl_xor(ssum, 16); ssq += __shfl_xor(ssq, 16);
        ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);
        if (lane < 8) { red[wave * 16 + lane] = ssum; red[wave * 16 + 8 + lane] = ssq; }
        __syncthreads();
        const int nbz4 = (D + 3) / 4, h = wave >> 2, bz4 = 2 * bz + h;
        if ((wave & 3) == 0 && lane < 16 && bz4 < nbz4) {
            const float* r4 = red + h * 64;
            const float v = (r4[lane] + r4[16 + lane]) + (r4[32 + lane] + r4[48 + lane]);
            stats[abn_part_at(lane >> 3, lane & 7, 8, ((int64_t)bz4 * nby + by) * nbx + bx, (int64_t)nbz4 * nby * nbx)] = v;
        }
    }
}

// nn.Conv3d weight w[8][Cin][3][3][3] (fp32) -> wq[chunk][plane hi|lo][f][dy][kg][co][8] fp16 pieces of w * 2^4: the B fragments of the kernel above
__global__ __launch_bounds__(256) void conv0_pack_f16x3_kernel(const float* __restrict__ w, int Cin, int nblk16, _Float16* __restrict__ wq)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = nblk16 * 5 * 3 * 4 * 8 * 8;
    if (i >= total) return;
    const int e = i & 7, co = (i >> 3) & 7, kg = (i >> 6) & 3, dy = (i >> 8) % 3, f = (i / (256 * 3)) % 5, c = i / (256 * 15);
    const int p = 2 * f + (kg >> 1);
    const int ci = c * 16 + (kg & 1) * 8 + e;
    float v = 0.0f;
    if (p < 9 && ci < Cin) {
        const int dz = p / 3, dx = p - 3 * dz;
        v = w[((int64_t)co * Cin + ci) * 27 + (dz * 3 + dy) * 3 + dx] * (float)(1 << W_SCALE_LOG2);
    }
    if (!(fabsf(v) <= 65504.0f)) wq[(int64_t)nblk16 * 7680] = (_Float16)1.0f;      // status word (zeroed by the launcher): the guarded encode then always takes fp32
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);
    const _Float16 hi = (_Float16)v;
    const int local = i - c * 3840;
    wq[(int64_t)c * 7680 + local] = hi;
    wq[(int64_t)c * 7680 + 3840 + local] = (_Float16)(v - (float)hi);
}

}  // namespace

extern "C" size_t mvsnerf_conv0_f16x3_packed_elems(int Cin)
{
    if (Cin < 1) return 0;
    return (size_t)((Cin + 15) / 16) * 2 * 5 * 3 * 4 * 8 * 8 + 8;          // + 8 status elements: [0] != 0 when a weight left fp16's range
}

extern "C" int mvsnerf_conv0_f16x3_pack(const float* w, int Cin, void* packed, void* stream)
{
    if (!w || !packed || Cin < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(packed)) return MVSNERF_EALIGN;
    const int nblk = (Cin + 15) / 16;
    hipError_t e = hipMemsetAsync(reinterpret_cast<_Float16*>(packed) + (size_t)nblk * 7680, 0, 8 * sizeof(_Float16), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    conv0_pack_f16x3_kernel<<<mvs_cdiv((int64_t)nblk * 3840, 256), 256, 0, (hipStream_t)stream>>>(w, Cin, nblk, reinterpret_cast<_Float16*>(packed));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

int mvs_conv0_f16x3_fwd(const void* x16, int Cin, int D, int H, int W, const void* packed, float* out, float* stats_part, int* guard, hipStream_t st)
{
    if (!x16 || !packed || !out || Cin < 1 || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x16) || !mvs_aligned16(packed)) return MVSNERF_EALIGN;
    if ((int64_t)D * H * W * 32 >= ((int64_t)1 << 31)) return MVSNERF_EUNSUPPORTED;
    static unsigned long long cap_mask = 0;
    if (int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(conv3d_k3s1_c8_f16x3_kernel), HBUF, &cap_mask)) return rc;
    const int tiles = ((W + HTX - 1) / HTX) * ((H + HTY - 1) / HTY) * ((D + HTZ - 1) / HTZ);       // statistics: the 4-plane slots of mvsnerf_conv0_bf16_tiles(D, H, W)
    conv3d_k3s1_c8_f16x3_kernel<<<tiles, 64 * H_WAVES, HBUF, st>>>(
        reinterpret_cast<const _Float16*>(x16), (Cin + 15) / 16, D, H, W, reinterpret_cast<const _Float16*>(packed), out, stats_part, guard);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_conv0_f16x3_fwd(const void* x16, int Cin, int D, int H, int W, const void* packed, float* out, float* stats_part, void* stream)
{
    return mvs_conv0_f16x3_fwd(x16, Cin, D, H, W, packed, out, stats_part, nullptr, (hipStream_t)stream);
}
