// 3x3x3 stride-1 convolution with 8 output channels on the matrix cores (conv0 of CostRegNet, models.py:756: 74.5 % of the
// encoder's FLOPs; fp32 in, fp32 accumulate, exact fp32 products).
//
// Why v_mfma_f32_4x4x1_16B_f32.  With Cout = 8 the 32- and 16-wide MFMA shapes leave 75 % / 50 % of their columns empty.  The
// 16-block 4x4x1 form multiplies, per block, 4 rows x 1 k x 4 columns; 16 blocks = 8 groups of four voxels x 2 groups of four
// output channels, i.e. 32 voxels x 8 channels per instruction with no padding, at the same 64 FLOP/clk/SIMD as every other
// fp32 MFMA (scratch/r2/mfma4x4.hip: 125-137 TFLOP/s sustained, with or without one ds_read_b128 per four MFMAs).  Against the
// VALU kernel it replaces (one v_pk_fma per 2 MACs, weights through the scalar cache, 62 % of the wave cycles in s_waitcnt) the
// matrix instruction retires 256 MACs per issue slot, so the instruction stream stops being the bound.
//
// Mapping.  A workgroup owns 4 x 16 x 16 output voxels (z, y, x); wave w owns plane z0+w: eight M-tiles, M-tile t = rows (t, t+8) x 16 x.
//   lane l: block = l>>2 (mb = block>>1: voxel quad, nb = block&1: channel quad), i = l&3
//   A operand (k = one input channel of one tap): lane holds the input of voxel m = 4 mb + i      (both nb copies: LDS broadcast)
//   B operand: lane holds w[tap][ci][4 nb + i]
//   D: register r of lane l = output (voxel 4 mb + r, channel 4 nb + i)
// The (6 x 18 x 18)-voxel input halo is staged through LDS eight channels at a time, channel-last ([voxel][8], 32 B per voxel)
// with the two 16-byte halves of a voxel swapped on every other group of eight voxels: a ds_read_b128 of 16 consecutive voxels
// then touches every bank once (un-swizzled, voxels v and v+8 would collide: 2-way conflict on every operand read).
// One ds_read_b128 feeds four MFMAs (k = 4 channels).  Two workgroups per CU (69 KB LDS each): one stages its next channel chunk
// while the other one multiplies.  (Measured and dropped: ONE workgroup per CU with two LDS tiles, the next chunk written into the
// other tile in the middle of the multiply, one barrier per chunk: 1.69 ms against 0.94 - a single wave per SIMD issues this
// dependent-accumulator MFMA stream at 53 TFLOP/s; it takes the second wave to fill the matrix pipe.)
#include "common.h"
#include "act.h"
#include "lds_dma.h"

#ifdef MVS_CONV_DBG
__device__ int g_dbg;            // scratch/r2/conv_bench.hip only: bit 0 = no staging, bit 1 = no operand reads, bit 2 = no weight loads; wgrad: bit 3 = no DMA, bit 4 = no MFMA phase
#define MVS_DBG(bit) (g_dbg & (bit))
#else
#define MVS_DBG(bit) false
#endif

namespace {

__device__ const f32x4 g_zero16 = {0.0f, 0.0f, 0.0f, 0.0f};   // what a DMA lane reads for a voxel outside the volume (zero padding)

// Per-workgroup InPlaceABN statistics of an 8-channel output tile straight from the accumulators (the layer's raw output is not read
// again by abn_partial_kernel: 150 MB for the two full-resolution layers): lane (mb, nb, i) holds values of channel 4 nb + i only, so a
// lane sums its own values, the 8 lanes of a channel meet by xor-shuffle over the mb bits, the 4 waves in LDS; workgroup `slot` writes
// part[abn_part_at({sum, sum of squares}, channel, 8, slot, nslots)] - the layout abn_finalize_kernel reads (common.h).
__device__ __forceinline__ void c8_tile_stats(float s, float q, int lane, int wave, float* red /* [4][16] */, float* __restrict__ part, int slot, int nslots)
{
#pragma unroll
    for (int o = 8; o <= 32; o <<= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    if (lane < 8) { red[wave * 16 + lane] = s; red[wave * 16 + 8 + lane] = q; }
    __syncthreads();
    if (wave == 0 && lane < 16) {
        const float v = (red[lane] + red[16 + lane]) + (red[32 + lane] + red[48 + lane]);
        part[abn_part_at(lane >> 3, lane & 7, 8, slot, nslots)] = v;
    }
}

constexpr int TX = 16, TY = 16, TZ = 4;             // output tile
constexpr int PX = TX + 2, PY = TY + 2, PZ = TZ + 2; // staged input tile
constexpr int NVOX = PZ * PY * PX;                   // 1944
constexpr int CK = 8;                                // channels per staged chunk (LDS row = 32 B)

// float offset of (voxel v at x position vx, 16-byte half c4) in the staged tile: the halves are swapped for vx in [8, 16)
__device__ __forceinline__ int swz_off(int v, int vx, int c4) { return v * CK + ((c4 ^ ((vx >> 3) & 1)) << 2); }

constexpr int NSLOT = (NVOX + 127) / 128;            // staging slots per thread: thread t owns half (t&1) of voxels (t>>1) + 128 j

// Staging is split in two so that the global loads of chunk c+1 are in flight while chunk c is multiplied:
//   stage_load   16 independent 16-byte loads per thread into registers (addresses precomputed once per workgroup)
//   stage_store  (pending InPlaceABN of the producer, if any, applied here) -> ds_write_b128
template <int CIN, int C0, int CKC, bool BLOCKED>
__device__ __forceinline__ void stage_load(const float* __restrict__ x, int64_t block_stride, const int (&goff)[NSLOT], int c4t, f32x4 (&st)[NSLOT])
{
    if (CKC == 4 && c4t) return;                       // a 4-channel chunk has no second half
    // channel-last: channel C0 + 4 c4t of voxel row goff; blocked: channel block C0/8 is its own [voxel][8] array
    const float* base = BLOCKED ? x + (int64_t)(C0 / 8) * block_stride + c4t * 4 : x + C0 + c4t * 4;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        st[j] = f32x4{0, 0, 0, 0};
        if (goff[j] >= 0) st[j] = *reinterpret_cast<const f32x4*>(base + (int64_t)goff[j]);
    }
}

template <int C0, int CKC>
__device__ __forceinline__ void stage_store(const ActSrc& a, float* __restrict__ tile, const int (&goff)[NSLOT], unsigned swz_bits,
                                            int tid, f32x4 (&st)[NSLOT])
{
    const int c4t = tid & 1;
    if (CKC == 4 && c4t) return;
    f32x4 sc = {1, 1, 1, 1}, sh = {0, 0, 0, 0};
    if (a.scale) { sc = *reinterpret_cast<const f32x4*>(a.scale + C0 + c4t * 4); sh = *reinterpret_cast<const f32x4*>(a.shift + C0 + c4t * 4); }
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int v = (tid >> 1) + 128 * j;
        if (v >= NVOX) continue;
        f32x4 val = st[j];
        if (a.scale && goff[j] >= 0) {                  // zero padding is the padding of the ACTIVATED tensor
#pragma unroll
            for (int k = 0; k < 4; ++k) val[k] = act_apply(val[k], sc[k], sh[k]);
        }
        *reinterpret_cast<f32x4*>(tile + v * CK + ((c4t ^ (int)((swz_bits >> j) & 1u)) << 2)) = val;
    }
}

// The chunk's weights ride along: w[tap][C0 + ci][co] (global, the layout of mvsnerf_conv3d_pack_weights) -> LDS wt[tap][co][8 ci], so
// that a lane's four k-values are one ds_read_b128.  (Loading them from global inside the multiply loop cost 0.19 of 1.12 ms:
// the loop runs one group of 32 MFMAs = 256-512 cycles ahead of its operands, which covers an LDS round trip but not a vector-memory one.)
constexpr int WT_FLOATS = 27 * 8 * CK;               // 1728
constexpr int WSLOT = (27 * 8 * 2 + 255) / 256;      // 16-byte pieces (tap, co, half) per thread: 2

template <int CIN, int C0, int CKC>
__device__ __forceinline__ void wstage_load(const float* __restrict__ wp, int tid, f32x4 (&wr)[WSLOT])
{
#pragma unroll
    for (int j = 0; j < WSLOT; ++j) {
        const int idx = tid + 256 * j;
        const int tap = idx >> 4, co = (idx >> 1) & 7, c4 = idx & 1;
        wr[j] = f32x4{0, 0, 0, 0};
        if (idx < 27 * 16 && c4 * 4 < CKC) {
            const float* g = wp + ((int64_t)tap * CIN + C0 + c4 * 4) * 8 + co;
#pragma unroll
            for (int k = 0; k < 4; ++k) wr[j][k] = g[k * 8];
        }
    }
}

__device__ __forceinline__ void wstage_store(float* __restrict__ wt, int tid, const f32x4 (&wr)[WSLOT])
{
#pragma unroll
    for (int j = 0; j < WSLOT; ++j) {
        const int idx = tid + 256 * j;
        if (idx < 27 * 16) *reinterpret_cast<f32x4*>(wt + idx * 4) = wr[j];      // idx = (tap*8 + co)*2 + half
    }
}

template <int CIN, int C0, int CKC, int CREAL>          // CREAL: real input channels (<= CIN): k-values beyond them are zero padding and skipped
__device__ __forceinline__ void mfma_chunk(const float* __restrict__ wtile, const float* __restrict__ tile, int wave, int lane, f32x4 (&acc)[8])
{
    constexpr int K4 = CKC / 4;
    const int blk = lane >> 2, i = lane & 3, mb = blk >> 1, nb = blk & 1;
    const int m = mb * 4 + i;                          // A row of this lane: M-tile t = output rows (t, t+8): row t + 8 (m>>4), x = m&15
    const int xx = m & 15;
    // per-lane part of the operand address for each (dx, c4); the tile row j and the tap plane dz are wave-uniform offsets
    const float* alane[3][K4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int c4 = 0; c4 < K4; ++c4)
            alane[dx][c4] = tile + swz_off((wave * PY + (m >> 4) * 8) * PX + xx + dx, xx + dx, c4);
    const float* wl = wtile + (nb * 4 + i) * CK;           // staged weights [tap][co][8 ci]
    // One group = (dz, dx, c4): the ten row pairs P_j = (row j | row j+8), j = 0..9, of plane z+dz at x+dx serve all three dy taps
    // of all eight M-tiles (M-tile t at tap dy reads P_{t+dy}): 10 operand reads + 3 weight reads per 96 MFMAs.  (With M-tiles of
    // two ADJACENT rows every tap needed its own eight reads: 24 per 96 MFMAs, and the LDS pipe ran at > 50 %.)
    // The MFMAs of a group run in j order; as soon as P_j has been used, its registers are refilled with P_j of the NEXT group, which
    // is needed ~700 cycles later: one register set, no LDS round trip on the critical path.  The scheduling fences pin that order.
    f32x4 av[10], bw[2][3];
    auto fetch_b = [&](int dz, int dx, int c4, f32x4 (&bd)[3]) {
        if (MVS_DBG(4)) return;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) bd[dy] = *reinterpret_cast<const f32x4*>(wl + ((dz * 3 + dy) * 3 + dx) * 8 * CK + c4 * 4);
    };
    auto fetch_a = [&](int dz, int dx, int c4, int j) {
        if (MVS_DBG(2)) return;
        av[j] = *reinterpret_cast<const f32x4*>(alane[dx][c4] + (dz * PY + j) * PX * CK);
    };
    // group (dz, dx, c4) with its operands in av / bd; (ndz, ndx, nc4) = the group to refill for (has_next)
    auto group = [&](f32x4 (&bd)[3], int c4, bool has_next, int ndz, int ndx, int nc4) {
        // rows j and j+5 together: P_j feeds M-tiles j, j-1, j-2 and P_{j+5} feeds j+5, j+4, j+3 - six different accumulators, so that
        // consecutive MFMAs never wait for each other's result (three accumulators in turn ran the loop 12 % slower)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int jj = j + 5 * h, t = jj - dy;
                        if (t >= 0 && t < 8 && C0 + c4 * 4 + k < CREAL)        // (c4 is a constant at every call site: folded)
                            acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[jj][k], bd[dy][k], acc[t], 0, 0, 0);
                    }
            if (has_next) { fetch_a(ndz, ndx, nc4, j); fetch_a(ndz, ndx, nc4, j + 5); }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    constexpr int NG = 3 * K4;                         // groups per plane dz
    fetch_b(0, 0, 0, bw[0]);
#pragma unroll
    for (int j = 0; j < 10; ++j) fetch_a(0, 0, 0, j);
    if constexpr ((NG & 1) == 0) {                     // weight-register parity is static per plane: roll the dz loop
#pragma unroll 1
        for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const bool last = g + 1 == NG;
                const int ndz = last ? dz + 1 : dz, ndx = last ? 0 : (g + 1) / K4, nc4 = last ? 0 : (g + 1) % K4;
                const bool has_next = !last || dz < 2;
                if (has_next) fetch_b(ndz, ndx, nc4, bw[(g + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                group(bw[g & 1], g % K4, has_next, ndz, ndx, nc4);
            }
        }
    } else {                                           // 4-channel tail chunk: nine groups, straight line
#pragma unroll
        for (int g = 0; g < 3 * NG; ++g) {
            const bool has_next = g + 1 < 3 * NG;
            const int n = g + 1, ndz = n / NG, ndx = (n % NG) / K4, nc4 = n % K4;
            if (has_next) fetch_b(ndz, ndx, nc4, bw[(g + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            group(bw[g & 1], g % K4, has_next, ndz, ndx, nc4);
        }
    }
}

template <int CIN, int CREAL, bool BLOCKED>
__global__ __launch_bounds__(256, 2) void conv3d_k3s1_c8_mfma_kernel(ActSrc a, int ld, int D, int H, int W,
                                                                    const float* __restrict__ wp, float* __restrict__ out, int swz)
{
    static_assert(CIN % 4 == 0 && CIN <= 64, "channel chunks of 8 (+ one of 4)");
    __shared__ __attribute__((aligned(16))) float tile[NVOX * CK];
    __shared__ __attribute__((aligned(16))) float wtile[WT_FLOATS];
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    const int tile_id = swz ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, bz = tile_id / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = bx * TX - 1, y0 = by * TY - 1, z0 = bz * TZ - 1;
    // staging slots of this thread (the same for every channel chunk): element offset in the input (-1: outside the volume -> 0)
    // and float offset in the LDS tile (-1: no such voxel)
    const int c4t = tid & 1;
    int goff[NSLOT];
    unsigned swz_bits = 0;                            // bit j: the 16-byte halves of voxel slot j are swapped (vx in [8, 16))
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        const int v = (tid >> 1) + 128 * j;
        const int vx = v % PX, vy = (v / PX) % PY, vz = v / (PX * PY);
        const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        const bool in = v < NVOX && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        goff[j] = in ? (((gz * H + gy) * W + gx) * (BLOCKED ? 8 : ld)) : -1;
        swz_bits |= (unsigned)((vx >> 3) & 1) << j;
    }
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
    f32x4 st[NSLOT];
    constexpr int NCH = (CIN + 7) / 8;
    const int64_t block_stride = (int64_t)D * H * W * 8;
    f32x4 wr[WSLOT];
    stage_load<CIN, 0, (CIN >= 8 ? 8 : 4), BLOCKED>(a.x, block_stride, goff, c4t, st);
    wstage_load<CIN, 0, (CIN >= 8 ? 8 : 4)>(wp, tid, wr);
#define MVS_CH(IDX)                                                                                                   \
    if constexpr ((IDX) < NCH) {                                                                                      \
        constexpr int C0_ = (IDX) * 8, CKC_ = (CIN - C0_ >= 8) ? 8 : 4;                                             \
        if (!MVS_DBG(1)) {                                                                                            \
        __syncthreads();                                   /* everybody finished reading the previous chunk */        \
        stage_store<C0_, CKC_>(a, tile, goff, swz_bits, tid, st);                                                         \
        wstage_store(wtile, tid, wr);                                                                                 \
        __syncthreads();                                                                                              \
        if constexpr ((IDX) + 1 < NCH) {                                                                              \
            stage_load<CIN, C0_ + 8, (CIN - C0_ - 8 >= 8) ? 8 : 4, BLOCKED>(a.x, block_stride, goff, c4t, st);       \
            wstage_load<CIN, C0_ + 8, (CIN - C0_ - 8 >= 8) ? 8 : 4>(wp, tid, wr);                                    \
        }                                                                                                             \
        }                                                                                                             \
        mfma_chunk<CIN, C0_, CKC_, CREAL>(wtile, tile, wave, lane, acc);                                                        \
    }
    MVS_CH(0) MVS_CH(1) MVS_CH(2) MVS_CH(3) MVS_CH(4) MVS_CH(5) MVS_CH(6) MVS_CH(7)
#undef MVS_CH
    // D: register r = voxel 4 mb + r of the M-tile, this lane's channel 4 nb + i
    const int blk = lane >> 2, i = lane & 3, mb = blk >> 1, nb = blk & 1;
    const int oz = bz * TZ + wave;
    if (oz >= D) return;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = mb * 4 + r;
            const int ox = bx * TX + (m & 15), oy = by * TY + t + 8 * (m >> 4);
            if (ox < W && oy < H) out[(((int64_t)oz * H + oy) * W + ox) * 8 + nb * 4 + i] = acc[t][r];
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv0, inference form: input in channel blocks of FOUR (x4[Cin/4][D*H*W][4], written by the plane sweep), staged by LDS-DMA.
// Same mapping and MFMA stream as above with four channels per chunk; what changes is how a chunk reaches LDS:
//   * one `global_load_lds_dwordx4` moves 64 consecutive tile voxels x 16 bytes (1 KB) from global memory straight into the tile - no
//     staging registers, no ds_write pass; a lane whose voxel lies outside the volume reads a 16-byte block of zeros instead;
//   * two tiles (+ two weight tiles): the DMA of chunk c+1 is issued when chunk c starts and lands during its 864 MFMAs per wave, so
//     a chunk costs ONE barrier and the matrix pipes never wait for staging (in the register-staged kernel above staging was
//     0.10 of 0.94 ms).  Two workgroups per CU (70 KB each).
// Weights: wq[chunk][tap][co][4] (mvsnerf_conv3d_pack_weights_c8), one 3.4 KB slab per chunk by the same DMA.
constexpr int T4_FLOATS = ((NVOX * 4 + 255) / 256) * 256;       // tile of 4-channel voxels, rounded up to whole 1 KB DMA pieces (31)
constexpr int T4_PIECES = T4_FLOATS / 256;
constexpr int W4_FLOATS = 1024;                                 // 27 x 8 x 4 = 864 weights, four DMA pieces
constexpr int T4_SLOTS = (T4_PIECES + 3) / 4;                   // DMA pieces per wave: 8

__device__ __forceinline__ void dma16_gather(const void* lane_ptr, unsigned lds_byte_uniform)
{
    // lanes read 16 B each at their own address; LDS receives them at lds_byte_uniform + lane * 16 (M0 carries the LDS base)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(lane_ptr), "s"(lds_byte_uniform) : "memory");
}

template <int KREAL>      // k-values (input channels) of this chunk that exist; the rest is zero padding and skipped
__device__ __forceinline__ void mfma_chunk4(const float* __restrict__ wtile, const float* __restrict__ tile, int wave, int lane, f32x4 (&acc)[8])
{
    const int blk = lane >> 2, i = lane & 3, mb = blk >> 1, nb = blk & 1;
    const int m = mb * 4 + i, xx = m & 15;
    const float* alane = tile + ((wave * PY + (m >> 4) * 8) * PX + xx) * 4;      // 16 consecutive x = 256 contiguous bytes: conflict-free
    const float* wl = wtile + (nb * 4 + i) * 4;
    f32x4 av[10], bw[2][3];
    auto fetch_b = [&](int g, f32x4 (&bd)[3]) {                  // group g = (dz, dx)
        const int dz = g / 3, dx = g - dz * 3;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) bd[dy] = *reinterpret_cast<const f32x4*>(wl + ((dz * 3 + dy) * 3 + dx) * 32);
    };
    auto fetch_a = [&](int g, int j) {
        const int dz = g / 3, dx = g - dz * 3;
        av[j] = *reinterpret_cast<const f32x4*>(alane + ((dz * PY + j) * PX + dx) * 4);
    };
    fetch_b(0, bw[0]);
#pragma unroll
    for (int j = 0; j < 10; ++j) fetch_a(0, j);
#pragma unroll
    for (int g = 0; g < 9; ++g) {
        if (g + 1 < 9) fetch_b(g + 1, bw[(g + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int k = 0; k < KREAL; ++k)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int jj = j + 5 * h, t = jj - dy;
                        if (t >= 0 && t < 8) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[jj][k], bw[g & 1][dy][k], acc[t], 0, 0, 0);
                    }
            if (g + 1 < 9) { fetch_a(g + 1, j); fetch_a(g + 1, j + 5); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int CIN, int CREAL>
__global__ __launch_bounds__(256, 2) void conv3d_k3s1_c8_mfma4_kernel(const float* __restrict__ x4, int D, int H, int W, const float* __restrict__ wq,
                                                                     float* __restrict__ out, int swz, float* __restrict__ stats, const int* __restrict__ run_if)
{
    static_assert(CIN % 4 == 0 && CREAL <= CIN && CREAL > CIN - 4, "chunks of four channels; only the last one may be partly padding");
    if (run_if && *run_if == 0) return;                           // fp32 half of a guarded 16-bit sequence (include/mvsnerf_hip.h)
    constexpr int NCH = CIN / 4;
    __shared__ __attribute__((aligned(1024))) float lds4[2 * (T4_FLOATS + W4_FLOATS)];
    constexpr int BUF = T4_FLOATS + W4_FLOATS;                    // one (tile, weights) buffer; the two alternate
    const int nbx = (W + TX - 1) / TX, nby = (H + TY - 1) / TY;
    const int tile_id = swz ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, bz = tile_id / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = bx * TX - 1, y0 = by * TY - 1, z0 = bz * TZ - 1;
    // this lane's DMA slots: piece p = wave + 4 j covers tile voxels 64 p .. 64 p + 63; element offset of the voxel in a channel block, -1 = zeros
    int goff[T4_SLOTS];
#pragma unroll
    for (int j = 0; j < T4_SLOTS; ++j) {
        const int v = (wave + 4 * j) * 64 + lane;
        const int vx = v % PX, vy = (v / PX) % PY, vz = v / (PX * PY);
        const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        const bool in = v < NVOX && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        goff[j] = in ? ((gz * H + gy) * W + gx) * 4 : -1;
    }
    const int64_t block_stride = (int64_t)D * H * W * 4;
    const float* zero16 = reinterpret_cast<const float*>(&g_zero16);
    auto issue = [&](int c, float* dst) {                         // DMA of chunk c (input tile + weight slab) into `dst`
        const float* xb = x4 + (int64_t)c * block_stride;
        const unsigned base = lds_byte_addr(dst);
#pragma unroll
        for (int j = 0; j < T4_SLOTS; ++j) {
            const int p = wave + 4 * j;
            if (p < T4_PIECES) dma16_gather(goff[j] >= 0 ? xb + goff[j] : zero16, base + p * 1024);
        }
        const int wi = wave * 64 + lane;                          // weight slab: 216 x 16 bytes, one piece per wave
        dma16_gather(wi < 216 ? wq + ((int64_t)c * 216 + wi) * 4 : zero16, base + T4_FLOATS * 4 + wave * 1024);
    };
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
    issue(0, lds4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) {
        float* cur = lds4 + (c & 1) * BUF;
        if (c + 1 < NCH) issue(c + 1, lds4 + ((c + 1) & 1) * BUF);       // lands while this chunk is multiplied (the other tile is free: barrier below)
        if (c + 1 < NCH || CREAL == CIN) mfma_chunk4<4>(cur + T4_FLOATS, cur, wave, lane, acc);
        else mfma_chunk4<CREAL - (CIN - 4)>(cur + T4_FLOATS, cur, wave, lane, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMA pieces have landed ...
        __syncthreads();                                          // ... everybody's have, and everybody is done reading `cur`
    }
    // D: lane (mb, nb, i) holds, in register r of accumulator t, channel 4 nb + i of voxel (x = 4 (mb & 3) + r, row t + 8 (mb >> 2)) of the wave's plane.  Stored from
    // there a lane writes one float per instruction (32 instructions per wave, eight 32-byte pieces each); a row of the plane is 512 contiguous bytes: the values go
    // through a wave-private stage in the (free: barrier above) tile buffers - 8 floats of padding per 4 voxels and 4 per row keep writes and reads off each
    // other's banks - and leave as eight 1 KB global_store_dwordx4 (two rows each).  Indices from an opaque copy of the thread index (nothing is hoisted above
    // the chunk loop); the InPlaceABN partial sums are taken in the order the direct stores took them.
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6;
    const int blk = lane_e >> 2, i = lane_e & 3, mb = blk >> 1, nb = blk & 1;
    const int oz = bz * TZ + wave_e;
    float ssum = 0.f, ssq = 0.f;
    constexpr int RS = 16 * 8 + 4 * 8 + 4;                        // floats per staged row
    float* stg = lds4 + wave_e * (16 * RS);
    {
        float* wr = stg + (mb >> 2) * (8 * RS) + (mb & 3) * 40 + nb * 4 + i;
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wr[t * RS + r * 8] = acc[t][r];
                const int m = mb * 4 + r;
                const int ox = bx * TX + (m & 15), oy = by * TY + t + 8 * (m >> 4);
                if (oz < D && ox < W && oy < H) { ssum += acc[t][r]; ssq = fmaf(acc[t][r], acc[t][r], ssq); }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // wave-private: no barrier
        const int xr = (lane_e & 31) >> 1, hq = lane_e & 1;
        const float* rd = stg + (lane_e >> 5) * RS + xr * 8 + (xr >> 2) * 8 + hq * 4;
        const int ox = bx * TX + xr;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(rd + 2 * k * RS);
            const int oy = by * TY + 2 * k + (lane_e >> 5);
            if (oz < D && oy < H && ox < W) *reinterpret_cast<f32x4*>(out + (((int64_t)oz * H + oy) * W + ox) * 8 + hq * 4) = v4;
        }
    }
    if (stats) {
        __syncthreads();                                          // the output stages are read: reuse the tiles' first floats
        c8_tile_stats(ssum, ssq, lane, wave, lds4, stats, tile_id, gridDim.x);
    }
}

// packed[tap][ci][8] (mvsnerf_conv3d_pack_weights, Cout = 8) -> wq[ci/4][tap][co][4]
__global__ __launch_bounds__(256) void conv_w4_repack_kernel(const float* __restrict__ wp, float* __restrict__ wq, int CIN)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 27 * CIN * 8) return;
    const int c4 = i & 3, co = (i >> 2) & 7, tap = (i >> 5) % 27, ch = i / (27 * 32);
    wq[i] = wp[((int64_t)tap * CIN + ch * 4 + c4) * 8 + co];
}

// ---------------------------------------------------------------------------------------------------------------------------
// The deep layers of CostRegNet (32 / 64 output channels, 9 152 .. 73 216 output voxels; models.py:758-761) on
// v_mfma_f32_32x32x2_f32: implicit GEMM, M = 32 consecutive output voxels, N = output channels, K = 27 taps x Cin.
// The VALU kernels they replace gave one thread one voxel x 16 channels, i.e. 144 workgroups for conv5/conv6: 10 TFLOP/s, bound by
// the length of a single thread's FMA chain.  Here a workgroup owns one M-tile and its four waves split the 27 taps (7/7/7/6); the
// partial accumulators meet in LDS and wave 0 adds them in a fixed order (deterministic) and stores.
//   lane (m = lane & 31, kh = lane >> 5):  A operand = 4 consecutive input channels of voxel m (one 16-byte load, channel-last input,
//   pending InPlaceABN applied on the fly), channels 8 cb + 4 kh + j; MFMA j of the quartet contracts channels (8cb + j, 8cb + 4 + j);
//   B operand = w[tap][ci][32 nb + m] (the layout of mvsnerf_conv3d_pack_weights: consecutive lanes -> consecutive floats).
// These volumes fit the L2 (conv6 input 2.3 MB), so the 27-fold re-read of the input comes from cache and needs no LDS tile.
template <int CIN, int COUT, int S>
__global__ __launch_bounds__(512) void conv3d_k3_mfma32_kernel(ActSrc a, int ld, int Di, int Hi, int Wi, const float* __restrict__ w32,
                                                              float* __restrict__ out, int Do, int Ho, int Wo, float* __restrict__ stats)
{
    constexpr int NB = COUT / 32, CB = CIN / 8;
    constexpr int WAVES = 8, NR = WAVES / NB;                    // wave w: output block w % NB, tap range w / NB of NR
    static_assert(COUT % 32 == 0 && CIN % 8 == 0 && WAVES % NB == 0, "32-wide output blocks, 8-channel k groups");
    __shared__ __attribute__((aligned(16))) float red[(NR - 1) * NB * 16 * 64];
    __shared__ __attribute__((aligned(16))) float act_sc[CIN], act_sh[CIN];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int nb = wave % NB, rg = wave / NB;
    const bool lazy = a.scale != nullptr;
    if (lazy) {
        for (int c = tid; c < CIN; c += 512) { act_sc[c] = a.scale[c]; act_sh[c] = a.shift[c]; }
        __syncthreads();
    }
    const int64_t nvox = (int64_t)Do * Ho * Wo;
    const int64_t vox = (int64_t)blockIdx.x * 32 + m;
    const bool live = vox < nvox;
    const int64_t vc = live ? vox : nvox - 1;
    int x, y, z;
    mvs_unflatten3(vc, Wo, Ho, x, y, z);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int t0 = (27 * rg) / NR, t1 = (27 * (rg + 1)) / NR;    // 27 taps dealt to NR ranges (3 or 4 each for NR = 8)
#pragma unroll 1
    for (int tap = t0; tap < t1; ++tap) {
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        const int zi = z * S - 1 + dz, yi = y * S - 1 + dy, xi = x * S - 1 + dx;
        const bool in = live && zi >= 0 && zi < Di && yi >= 0 && yi < Hi && xi >= 0 && xi < Wi;
        const float* src = a.x + (in ? ((int64_t)zi * Hi + yi) * Wi + xi : 0) * ld + kh * 4;
        const float* wt = w32 + ((int64_t)tap * CB * COUT + nb * 32 + m) * 8 + kh * 4;      // w32[tap][cb][co][8]
        f32x4 av[CB], bw[CB];
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {                        // all loads of the tap first: 2 CB independent 16-byte requests in flight
            av[cb] = *reinterpret_cast<const f32x4*>(src + cb * 8);
            bw[cb] = *reinterpret_cast<const f32x4*>(wt + (int64_t)cb * COUT * 8);
        }
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f32x4 v = av[cb];
            if (lazy) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(act_sc + cb * 8 + kh * 4), sh = *reinterpret_cast<const f32x4*>(act_sh + cb * 8 + kh * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], sc[j], sh[j]);
            }
            if (!in) v = f32x4{0, 0, 0, 0};                      // zero padding of the ACTIVATED input
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], bw[cb][j], acc, 0, 0, 0);
        }
    }
    // combine the tap ranges: ranges 1.. park their accumulators in LDS, range 0 adds them in order (deterministic) and stores
    if (rg > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(((rg - 1) * NB + nb) * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (rg == 0) {
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
#pragma unroll
            for (int w = 0; w < NR - 1; ++w) v += red[((w * NB + nb) * 16 + r) * 64 + lane];
            // D: register r of lane (n = m, half = kh) = output voxel (r&3) + 8 (r>>2) + 4 half of the tile, channel 32 nb + n
            const int64_t ov = (int64_t)blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (ov < nvox) { out[ov * COUT + nb * 32 + m] = v; ssum += v; ssq = fmaf(v, v, ssq); }
        }
        if (stats) {                                              // InPlaceABN partial sums of this M-tile (abn_finalize_kernel's layout)
            ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);
            if (kh == 0) {
                stats[abn_part_at(0, nb * 32 + m, COUT, blockIdx.x, gridDim.x)] = ssum;
                stats[abn_part_at(1, nb * 32 + m, COUT, blockIdx.x, gridDim.x)] = ssq;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Transposed 3x3x3 convolution, stride 2, padding 1, output_padding 1 (models.py:739-752: conv7 64->32, conv9 32->16, conv11 16->8) on
// v_mfma_f32_32x32x2_f32.  out[o] = sum over taps k with o = 2 i - 1 + k.  Per dimension an output of parity 0 (o = 2i) has one tap
// (k = 1, input i) and one of parity 1 (o = 2i + 1) two (k = 0 from input i + 1, k = 2 from input i), so the eight parity classes of
// o are eight small gather-form convolutions over the SAME input positions i.  M = 32 consecutive input positions.  The 32 columns of
// the MFMA are COUT output channels x NCLS = 32 / COUT parity classes of the innermost dimensions (COUT = 32: one class; 16: the two x
// parities; 8: the four (y, x) parities), so that no column is padding; the remaining ("outer") parity bits come from blockIdx.y.  The
// contraction enumerates, for merged dimensions, both input offsets {0, 1} (a column whose parity has no tap at an offset gets a zero
// weight: 25 % / 44 % of the products for COUT = 16 / 8), for outer dimensions the one or two taps of the parity, times Cin.
// Workgroup = (M-tile, outer class); its 8 waves deal the (offset combination, 8-channel group) units round-robin and meet in LDS.
// WAVES waves share an M-tile (they deal the contraction units and meet in LDS); TPW M-tiles per workgroup.  conv7 has 8..64 units per
// tile (8 waves); conv11 has 8 or 16 - with 8 waves the LDS reduction cost more than the products (281 us against 199 us for the VALU
// kernel), one wave per tile and no reduction at all is the right shape there.
template <int CIN, int COUT, int WAVES, int TPW>
__global__ __launch_bounds__(64 * WAVES * TPW) void convT3d_k3s2_mfma32_kernel(const float* __restrict__ x, int Di, int Hi, int Wi,
                                                                              const float* __restrict__ w32, float* __restrict__ out,
                                                                              float* __restrict__ stats)
{
    constexpr int NCLS = 32 / COUT, MD = NCLS == 1 ? 0 : NCLS == 2 ? 1 : 2;      // merged dimensions: none | x | y and x
    constexpr int CB = CIN / 8;
    static_assert(COUT == 8 || COUT == 16 || COUT == 32, "32 columns = COUT channels x 32/COUT parity classes");
    __shared__ __attribute__((aligned(16))) float red_all[TPW * (WAVES > 1 ? WAVES - 1 : 1) * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wv % WAVES, tsel = wv / WAVES;                // wave within its tile, tile within the workgroup
    float* red = red_all + tsel * (WAVES > 1 ? WAVES - 1 : 1) * 16 * 64;
    const int64_t tile = (int64_t)blockIdx.x * TPW + tsel;
    const int m = lane & 31, kh = lane >> 5;
    const int oc = blockIdx.y;                                     // outer parity bits: MD = 0: (pz,py,px); 1: (pz,py); 2: (pz)
    const int pz = MD == 0 ? (oc >> 2) & 1 : MD == 1 ? (oc >> 1) & 1 : oc & 1;
    const int py_o = MD == 0 ? (oc >> 1) & 1 : oc & 1;             // used when y is an outer dimension (MD <= 1)
    const int px_o = oc & 1;                                       // used when x is an outer dimension (MD == 0)
    // this lane as a B/D column: channel and merged-dimension parities
    const int co = m % COUT, icls = m / COUT;
    const int py_c = MD == 2 ? (icls >> 1) & 1 : py_o, px_c = MD >= 1 ? icls & 1 : px_o;
    // this lane as an A row: input position of the M-tile
    const int64_t nin = (int64_t)Di * Hi * Wi;
    const int64_t pos = tile * 32 + m;
    const bool live = pos < nin;
    const int64_t pc = live ? pos : nin - 1;
    int ix, iy, iz;
    mvs_unflatten3(pc, Wi, Hi, ix, iy, iz);
    // contraction entries per dimension: (input offset, kernel tap or -1) - uniform per workgroup for outer dims, per column for merged
    const int nz = pz ? 2 : 1;
    const int ny = MD == 2 ? 2 : (py_o ? 2 : 1);
    const int nx = MD >= 1 ? 2 : (px_o ? 2 : 1);
    auto tap_of = [](int parity, int off, bool merged, int entry) -> int {          // kernel index along one dimension, -1 = no tap
        if (merged) return parity ? (off ? 0 : 2) : (off ? -1 : 1);
        return parity ? (entry ? 2 : 0) : 1;
    };
    auto off_of = [](int parity, bool merged, int entry) -> int { return merged ? entry : (parity ? (entry ? 0 : 1) : 0); };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int n_units = nz * ny * nx * CB;
#pragma unroll 1
    for (int u = wave; u < n_units; u += WAVES) {
        const int cb = u % CB, e = u / CB;
        const int ex = e % nx, ey = (e / nx) % ny, ez = e / (nx * ny);
        const int oz = off_of(pz, false, ez), oy = off_of(py_o, MD == 2, ey), ox = off_of(px_o, MD >= 1, ex);     // input offsets (uniform)
        const int kz = tap_of(pz, oz, false, ez), ky = tap_of(py_c, oy, MD == 2, ey), kx = tap_of(px_c, ox, MD >= 1, ex);   // taps (per column)
        const int zi = iz + oz, yi = iy + oy, xi = ix + ox;
        const bool in = live && zi < Di && yi < Hi && xi < Wi;
        f32x4 av = {0, 0, 0, 0}, bw = {0, 0, 0, 0};
        if (in) av = *reinterpret_cast<const f32x4*>(x + (((int64_t)zi * Hi + yi) * Wi + xi) * CIN + cb * 8 + kh * 4);
        if (ky >= 0 && kx >= 0)
            bw = *reinterpret_cast<const f32x4*>(w32 + ((int64_t)(((kz * 3 + ky) * 3 + kx) * CB + cb) * COUT + co) * 8 + kh * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bw[j], acc, 0, 0, 0);
    }
    if constexpr (WAVES > 1) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
    }
    if (wave == 0) {
        const int Ho = 2 * Hi, Wo = 2 * Wi;
        float ssum = 0.f, ssq = 0.f;
        // the 16 input positions of this lane's results lie within 28 of tile * 32 + 4 kh: one decomposition, then small carries (16 x three 64-bit
        // divisions per lane were most of this kernel's VALU instructions)
        int bx, by, bz;
        mvs_unflatten3(tile * 32 + 4 * kh, Wi, Hi, bx, by, bz);
        const float rW = 1.0f / (float)Wi, rH = 1.0f / (float)Hi;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = acc[r];
            if constexpr (WAVES > 1) {
#pragma unroll
                for (int w = 0; w < WAVES - 1; ++w) v += red[(w * 16 + r) * 64 + lane];
            }
            // D: register r of lane (column m, half kh) = input position (r&3) + 8 (r>>2) + 4 kh of the tile
            const int64_t ip = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (ip < nin) {
                int jx, jy, jz;
                mvs_carry3(bx, by, bz, (r & 3) + 8 * (r >> 2), Wi, Hi, rW, rH, jx, jy, jz);
                out[((((int64_t)(2 * jz + pz)) * Ho + 2 * jy + py_c) * Wo + 2 * jx + px_c) * COUT + co] = v;
                ssum += v; ssq = fmaf(v, v, ssq);
            }
        }
        if (stats) {
            // InPlaceABN partial sums of this (M-tile, outer parity class): the lanes of a channel are its two halves (kh) and its NCLS merged
            // parity classes (columns co + COUT icls); slot = tile * gridDim.y + class, gridDim.x * TPW * gridDim.y slots (abn_part_at)
            ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);
#pragma unroll
            for (int o = COUT; o < 32; o <<= 1) { ssum += __shfl_xor(ssum, o); ssq += __shfl_xor(ssq, o); }
            if (lane < COUT) {
                const int64_t nslots = (int64_t)gridDim.x * TPW * gridDim.y, slot = tile * gridDim.y + blockIdx.y;
                stats[abn_part_at(0, co, COUT, slot, nslots)] = ssum;
                stats[abn_part_at(1, co, COUT, slot, nslots)] = ssq;
            }
        }
    }
}

// weights [tap][ci][co] (mvsnerf_conv3d_pack_weights) -> [tap][ci/8][co][8]: a lane's four k-values become one 16-byte load
__global__ __launch_bounds__(256) void conv_w32_repack_kernel(const float* __restrict__ wp, float* __restrict__ w32, int CIN, int COUT)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 27 * CIN * COUT) return;
    const int c8 = i & 7, co = (i >> 3) % COUT, cb = (i / (8 * COUT)) % (CIN / 8), tap = i / (CIN * COUT);
    w32[i] = wp[((int64_t)tap * CIN + cb * 8 + c8) * COUT + co];
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv0's weight gradient (8 x Cin x 27 sums over all voxels; 41 % of the CostRegNet backward's FLOPs) on v_mfma_f32_4x4x1_16B_f32:
//   gW[co][ci][tap] = sum_u X[u][ci] * G[u - (tap - 1)][co]      X: cost volume in channel blocks of four, G: gradient of conv0's raw output
// Here the voxels are the k dimension.  One instruction takes eight voxels (block pair mb) and multiplies, per voxel, four input
// channels (A operand, the wave's channel block) by the eight output channels (B operand, nb = which four): the 4 x 4 results of
// the 16 blocks are (voxel slot mb, co quad nb) partial sums of one tap, so 27 accumulators (108 registers) hold every tap of a
// channel block and nothing is padded.  A wave owns one channel block; the four waves of a workgroup share the G halo tile.
//   A: lane (mb, nb, i) = X[u0 + mb][4 cg + i]          32 contiguous floats of the blocked tile (both nb read the same: broadcast)
//   B: lane (mb, nb, j) = G[u0 + mb - tap + 1][4 nb + j] 64 contiguous floats of the channel-last halo tile: conflict-free
// Marching along y, the three dy taps slide over the same G rows: a step reads one new halo row (9 operands: 3 dz x 3 dx) and one
// A operand for 27 MFMAs.  The accumulators stay in registers across all tiles a workgroup visits (grid-stride over 2 x 16 x 16
// tiles); at the end the eight voxel slots are folded across lanes and the workgroup leaves ONE partial result (deterministic; the
// partials are summed by mvs_partial_sum).  grid.y splits the channel blocks four at a time (G is re-read per split: it is the
// small operand, 32 B per voxel against 16 B per voxel and channel block).
constexpr int GW_X = 16, GW_Y = 16, GW_Z = 2;                        // X tile
constexpr int GH_X = GW_X + 2, GH_Y = GW_Y + 2, GH_Z = GW_Z + 2;     // G halo tile
constexpr int GW_NV = GW_X * GW_Y * GW_Z;                            // 512 voxels: 8 DMA pieces per channel block
constexpr int GH_NV = GH_X * GH_Y * GH_Z;                            // 1296 voxels of 32 B: 40.5 DMA pieces
constexpr int GW_XF = 4 * GW_NV * 4;                                 // floats of the X tile: [4 channel blocks][512][4]
constexpr int GW_GP = (GH_NV * 2 + 63) / 64;                         // 41
constexpr int GW_GF = GW_GP * 256;
constexpr int GW_GSLOTS = (GW_GP + 3) / 4;                           // G pieces per wave: 11

__device__ __forceinline__ void wgrad4_tile(const float* __restrict__ xw, const float* __restrict__ gt, int lane, f32x4 (&acc)[27])
{
    const int mb = lane >> 3, i = lane & 3;
    const float* xl = xw + mb * 4 + i;
    const float* gl = gt + (mb + 2) * 8 + (lane & 7);
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        const int tz = q >> 1, xg = q & 1;
        const float* xa = xl + (tz * (GW_X * GW_Y) + xg * 8) * 4;
        const float* gb = gl + xg * 64 + tz * (GH_Y * GH_X * 8);
        float win[4][3][3];                                           // [halo row % 4][dz][dx]: the row of the NEXT step is in flight
        auto load_row = [&](int hy) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) win[hy % 4][a][c] = gb[(((2 - a) * GH_Y + hy) * GH_X - c) * 8];
        };
        load_row(0);
        load_row(1);
        load_row(2);
        float a_cur = xa[0];
#pragma unroll
        for (int y = 0; y < GW_Y; ++y) {
            if (y + 3 < GH_Y) load_row(y + 3);                        // consumed by the dy = 0 taps of the next step
            const float a_nxt = y + 1 < GW_Y ? xa[(y + 1) * GW_X * 4] : 0.0f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 2; b >= 0; --b) {
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        acc[(a * 3 + b) * 3 + c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a_cur, win[(y + 2 - b) % 4][a][c], acc[(a * 3 + b) * 3 + c], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
        }
    }
}

__global__ __launch_bounds__(256, 2) void conv3d_k3s1_c8_wgrad4_kernel(const float* __restrict__ x4, const float* __restrict__ g, int D, int H, int W,
                                                                      int NCG, int B, float* __restrict__ partial)
{
    __shared__ __attribute__((aligned(1024))) float lds[GW_XF + GW_GF];
    float* xt = lds;
    float* gt = lds + GW_XF;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.y, cg = split * 4 + wave;
    const bool active = cg < NCG;
    const int nbx = (W + GW_X - 1) / GW_X, nby = (H + GW_Y - 1) / GW_Y, nbz = (D + GW_Z - 1) / GW_Z;
    const int ntiles = nbx * nby * nbz;
    const int64_t nvox = (int64_t)D * H * W;
    // halo voxel of this lane in its G pieces (piece = wave + 4 jj covers 32 voxels, two lanes per voxel) and its byte offset from the
    // halo's first voxel: tile-invariant, so a tile inside the volume costs one scalar base address and one DMA instruction per piece
    int hpk[GW_GSLOTS];
    unsigned grel[GW_GSLOTS];
#pragma unroll
    for (int jj = 0; jj < GW_GSLOTS; ++jj) {
        const int gp = wave + 4 * jj, hv = gp * 32 + (lane >> 1);
        const int hx = hv % GH_X, hy = (hv / GH_X) % GH_Y, hz = hv / (GH_X * GH_Y);
        const bool in = gp < GW_GP && hv < GH_NV;
        hpk[jj] = in ? (hx | (hy << 8) | (hz << 16)) : -1;
        grel[jj] = in ? (unsigned)(((hz * H + hy) * W + hx) * 8 + (lane & 1) * 4) * 4u : 0u;
    }
    const int tx = lane & 15, ty = wave * 4 + (lane >> 4);           // this lane's voxel in an X piece (plane tz = piece parity)
    const unsigned xbase = lds_byte_addr(xt), gbase = lds_byte_addr(gt);
    const unsigned xrel[2] = {(unsigned)(ty * W + tx) * 16u, (unsigned)((H + ty) * W + tx) * 16u};
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = f32x4{0, 0, 0, 0};
    // Tile order: XCD x (= blockIdx.x & 7; gridDim.x is a multiple of 8, so the channel splits of a tile sequence share the XCD) walks
    // its own contiguous eighth of the tiles, its workgroups side by side: neighbouring halos and the other splits' reads of the same
    // G tile come out of that XCD's L2 instead of HBM.
    const int per_xcd = (ntiles + 7) >> 3, wg_per_xcd = gridDim.x >> 3;
    const int t_end = min(((int)(blockIdx.x & 7) + 1) * per_xcd, ntiles);
#pragma unroll 1
    for (int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3); tile < t_end; tile += wg_per_xcd) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int x0 = bx * GW_X, y0 = by * GW_Y, z0 = bz * GW_Z;
        __syncthreads();                                              // everybody finished reading the previous tile
        // wave-uniform bases: first voxel of the X tile in channel block 0, first voxel of the G halo (outside the array for a border
        // tile: only lanes whose own voxel is inside dereference it)
        const char* xorg = reinterpret_cast<const char*>(x4) + (((int64_t)z0 * H + y0) * W + x0) * 16;
        const char* gorg = reinterpret_cast<const char*>(g) + ((((int64_t)z0 - 1) * H + (y0 - 1)) * W + (x0 - 1)) * 32;
        const bool interior = x0 >= 1 && x0 + GW_X + 1 <= W && y0 >= 1 && y0 + GW_Y + 1 <= H && z0 >= 1 && z0 + GW_Z + 1 <= D;
        if (MVS_DBG(8)) {
        } else if (interior) {
#pragma unroll
            for (int j = 0; j < 8; ++j)                               // X pieces: channel block j >> 1 of the split, plane j & 1
                if (split * 4 + (j >> 1) < NCG) lds_dma_1k(xorg + (int64_t)(split * 4 + (j >> 1)) * nvox * 16, xbase + (wave + 4 * j) * 1024, xrel[j & 1]);
#pragma unroll
            for (int jj = 0; jj < GW_GSLOTS; ++jj)
                if (wave + 4 * jj < GW_GP) lds_dma_1k(gorg, gbase + (wave + 4 * jj) * 1024, grel[jj]);
        } else {
            // border tile: a lane whose voxel is outside the volume writes the zero padding itself (the DMA skips inactive lanes)
            const bool xy_ok = x0 + tx < W && y0 + ty < H;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (split * 4 + (j >> 1) >= NCG) continue;
                float* dst = xt + (wave + 4 * j) * 256 + lane * 4;
                if (xy_ok && z0 + (j & 1) < D) lds_dma_1k(xorg + (int64_t)(split * 4 + (j >> 1)) * nvox * 16, xbase + (wave + 4 * j) * 1024, xrel[j & 1]);
                else *reinterpret_cast<f32x4*>(dst) = f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int jj = 0; jj < GW_GSLOTS; ++jj) {
                const int gp = wave + 4 * jj;
                if (gp >= GW_GP) continue;
                const int gx = x0 - 1 + (hpk[jj] & 255), gy = y0 - 1 + ((hpk[jj] >> 8) & 255), gz = z0 - 1 + (hpk[jj] >> 16);
                float* dst = gt + gp * 256 + lane * 4;
                if (hpk[jj] >= 0 && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D) lds_dma_1k(gorg, gbase + gp * 1024, grel[jj]);
                else *reinterpret_cast<f32x4*>(dst) = f32x4{0, 0, 0, 0};
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (active && !MVS_DBG(16)) wgrad4_tile(xt + wave * (GW_NV * 4), gt, lane, acc);
    }
    if (!active) return;
    // fold the eight voxel slots (lane bits 3..5); lanes 0..7 then hold gW[co = lane][ci = 4 cg + r][tap]
    const int co = lane & 7;
    float* po = partial + ((int64_t)blockIdx.x * 8 + co) * B * 27;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[t][r];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (lane < 8 && cg * 4 + r < B) po[(int64_t)(cg * 4 + r) * 27 + t] = v;
        }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Transposed 3x3x3 convolution, stride 2, 16 -> 8 channels (conv11, models.py:751; the data gradient of conv1 has the same shape) on
// v_mfma_f32_4x4x1_16B_f32.  The parity-class kernel above spends 44 % of its products on zero weights at 8 output channels (170 us for
// 4 GFLOP and 188 MB).  Here the 16 blocks are 8 input-voxel quads x 2 output-channel quads and a class's taps are enumerated exactly:
// an output of parity p in one dimension takes (k = 1, input j) for p = 0 and (k = 0, input j + 1), (k = 2, input j) for p = 1, so input
// offset (dz, dy, dx) in {0,1}^3 serves 2^(number of zeros) (class, tap) pairs, 27 in all.  A wave owns two M-tiles (two input rows x 16 x
// each) and all 8 classes (64 accumulator registers); per (offset, channel quad) it reads the A operands once (one ds_read_b128 per
// M-tile: four k-steps) and, per pair, one b128 of weights for 8 MFMAs.  Input tile [ci/4][voxel][4] (conflict-free operand reads),
// weights [ci/4][tap][co][4] (mvsnerf_pack_weights_multi kind 1 / mvsnerf_conv3d_pack_weights_c8 of the layer's packed weights).
constexpr int CT8_TX = 16, CT8_TY = 8, CT8_TZ = 2;
constexpr int CT8_PX = CT8_TX + 1, CT8_PY = CT8_TY + 1, CT8_PZ = CT8_TZ + 1;
constexpr int CT8_NVH = CT8_PX * CT8_PY * CT8_PZ;                     // 459

__global__ __launch_bounds__(256) void convT3d_k3s2_c16to8_mfma4_kernel(ActSrc xa, ActSrc xb, int Di, int Hi, int Wi,
                                                                       const float* __restrict__ wq, float* __restrict__ out, int swz,
                                                                       float* __restrict__ stats)
{
    __shared__ __attribute__((aligned(16))) float xt[4 * CT8_NVH * 4];      // [ci quad][halo voxel][4]
    __shared__ __attribute__((aligned(16))) float wt[4 * 27 * 8 * 4];       // [ci quad][tap][co][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbx = (Wi + CT8_TX - 1) / CT8_TX, nby = (Hi + CT8_TY - 1) / CT8_TY;
    const int tile_id = swz ? xcd_contiguous_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, bz = tile_id / (nbx * nby);
    const int x0 = bx * CT8_TX, y0 = by * CT8_TY, z0 = bz * CT8_TZ;
    for (int i = tid; i < 4 * 27 * 8; i += 256) *reinterpret_cast<f32x4*>(wt + i * 4) = *reinterpret_cast<const f32x4*>(wq + i * 4);
    for (int it = tid; it < 4 * CT8_NVH; it += 256) {
        const int cq = it & 3, v = it >> 2;
        const int vx = v % CT8_PX, vy = (v / CT8_PX) % CT8_PY, vz = v / (CT8_PX * CT8_PY);
        const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        // the pending InPlaceABN of the producer(s) and the U-Net skip sum are applied here, once per staged element (the input used to
        // be materialised by a separate abn_apply_add pass: 113 MB of traffic for conv11)
        if (gx < Wi && gy < Hi && gz < Di) load_act4<16>(xa, xb, ((int64_t)gz * Hi + gy) * Wi + gx, 16, cq * 4, val);
        *reinterpret_cast<f32x4*>(xt + (cq * CT8_NVH + v) * 4) = val;
    }
    __syncthreads();
    const int mb = lane >> 3, nb = (lane >> 2) & 1, li = lane & 3;
    const int m = mb * 4 + li;                                        // A operand: this lane's voxel of the M-tile (row m >> 4, x m & 15)
    // the wave's two M-tiles: tile rows (2 r2, 2 r2 + 1) of plane tz, M-tile index mt = wave * 2 + r -> tz = mt >> 2, r2 = mt & 3
    f32x4 acc[8][2];
#pragma unroll
    for (int c = 0; c < 8; ++c) { acc[c][0] = f32x4{0, 0, 0, 0}; acc[c][1] = f32x4{0, 0, 0, 0}; }
    int abase[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int mt = wave * 2 + r, tz = mt >> 2, r2 = mt & 3;
        abase[r] = ((tz * CT8_PY + r2 * 2 + (m >> 4)) * CT8_PX + (m & 15)) * 4;
    }
    const float* wl = wt + (nb * 4 + li) * 4;                         // B operand: this lane's output channel 4 nb + li
#pragma unroll
    for (int off = 0; off < 8; ++off) {
        const int dz = off >> 2, dy = (off >> 1) & 1, dx = off & 1;
#pragma unroll
        for (int cq = 0; cq < 4; ++cq) {
            f32x4 a[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) a[r] = *reinterpret_cast<const f32x4*>(xt + cq * (CT8_NVH * 4) + abase[r] + ((dz * CT8_PY + dy) * CT8_PX + dx) * 4);
            // the (class, tap) pairs of this offset: per dimension d = 0 -> (p, k) in {(0, 1), (1, 2)}, d = 1 -> (1, 0)
#pragma unroll
            for (int sz = 0; sz < (dz ? 1 : 2); ++sz)
#pragma unroll
                for (int sy = 0; sy < (dy ? 1 : 2); ++sy)
#pragma unroll
                    for (int sx = 0; sx < (dx ? 1 : 2); ++sx) {
                        const int pz = dz ? 1 : sz, kz = dz ? 0 : 1 + sz;
                        const int py = dy ? 1 : sy, ky = dy ? 0 : 1 + sy;
                        const int px = dx ? 1 : sx, kx = dx ? 0 : 1 + sx;
                        const int cls = pz * 4 + py * 2 + px, tap = (kz * 3 + ky) * 3 + kx;
                        const f32x4 b = *reinterpret_cast<const f32x4*>(wl + ((cq * 27 + tap) * 8) * 4);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            acc[cls][0] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[0][k], b[k], acc[cls][0], 0, 0, 0);
                            acc[cls][1] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[1][k], b[k], acc[cls][1], 0, 0, 0);
                        }
                    }
        }
    }
    // D: register q of lane (mb, nb, li) = (voxel 4 mb + q of the M-tile, channel 4 nb + li).  Stored from there, a lane writes ONE float per instruction and a
    // wave-instruction touches eight 32-byte pieces 256 bytes apart (64 such instructions per wave: the kernel was bound by them, 98 us for 190 MB).  An input row
    // of 16 voxels and one output-row parity (pz, py) make 32 CONSECUTIVE output voxels = 1 KB of the channel-last output: the accumulators go through the (now
    // free) input tile - a wave-private stage of four such rows at a time, padded so that writes and reads are conflict-free - and leave as one 1 KB
    // global_store_dwordx4 per row (16 per wave).
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    float ssum = 0.f, ssq = 0.f;
    if (stats) {                                                      // the InPlaceABN partial sums, in the order the direct stores took them (same bits as before)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int mt = wave * 2 + r, jz = z0 + (mt >> 2);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mm = mb * 4 + q, jy = y0 + (mt & 3) * 2 + (mm >> 4), jx = x0 + (mm & 15);
                if (jz >= Di || jy >= Hi || jx >= Wi) continue;
#pragma unroll
                for (int cls = 0; cls < 8; ++cls) { ssum += acc[cls][r][q]; ssq = fmaf(acc[cls][r][q], acc[cls][r][q], ssq); }
            }
        }
    }
    __syncthreads();                                                  // every wave has read its A operands: the input tile is free
    constexpr int CT8_RS = 304;                                       // floats per staged row: 32 voxels x 8 + 4 x 8 padding, 2 rows = 32 banks apart
    float* stg = xt + wave * (4 * CT8_RS);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int mt = wave * 2 + r, jz = z0 + (mt >> 2);
#pragma unroll
        for (int pz = 0; pz < 2; ++pz) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mm = mb * 4 + q, row = mm >> 4, jxl = mm & 15;
#pragma unroll
                for (int pyx = 0; pyx < 4; ++pyx) {
                    const int py = pyx >> 1, px = pyx & 1, cls = pz * 4 + pyx;
                    const int ox = 2 * jxl + px;
                    stg[(row * 2 + py) * CT8_RS + ox * 8 + (ox >> 3) * 8 + nb * 4 + li] = acc[cls][r][q];
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // wave-private stage: no barrier
            const int ox = lane >> 1, chq = lane & 1;
#pragma unroll
            for (int ri = 0; ri < 4; ++ri) {
                const int row = ri >> 1, py = ri & 1;
                const f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + ri * CT8_RS + ox * 8 + (ox >> 3) * 8 + chq * 4);
                const int jy = y0 + (mt & 3) * 2 + row;
                const int oz = 2 * jz + pz, oy = 2 * jy + py, oxg = 2 * x0 + ox;
                if (jz < Di && jy < Hi && oxg < Wo) *reinterpret_cast<f32x4*>(out + (((int64_t)oz * Ho + oy) * Wo + oxg) * 8 + chq * 4) = v4;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the rows are read before the next pass overwrites them
        }
    }
    if (stats) {
        __syncthreads();                                              // the output stages are free
        c8_tile_stats(ssum, ssq, lane, wave, xt, stats, tile_id, gridDim.x);
    }
}

}  // namespace

// Called by mvsnerf_conv3d_fwd (encoder.hip) for the layers with 32 / 64 output channels; MVSNERF_EUNSUPPORTED = not instantiated.
// ---------------------------------------------------------------------------------------------------------------------------
// conv1 (8 -> 16, stride 2; and the data gradient of conv11, the same shape) of CostRegNet (models.py:757) on v_mfma_f32_16x16x4_f32.
// Measured (profiles/r03_*): 110 us against 109 us for the VALU kernel it replaces - the layer reads 150 MB and is bound by that - but the
// InPlaceABN statistics now come out of the same launch.  The SAME kernel for conv2 (16 -> 16, stride 1; template <16, 1>) ran 159 us against
// 116 us for the LDS-tiled VALU kernel (one wave per 32 voxels re-reads every input voxel 27 times through L1: 2.3 GB of L1 traffic for a
// 37 MB tensor, where the tiled kernel stages a halo once) and is NOT dispatched: conv2 keeps the VALU kernel.  One WAVE per
// 32 consecutive output voxels = two 16-voxel M-tiles that share every B operand; lane (m = lane & 15, kh = lane >> 4): A = CPL consecutive
// input channels of voxel m at the tap (channels CPL kh + j, one 8- or 16-byte load, pending InPlaceABN applied on the fly), B = the same
// channels of w32[tap][ci / 8][co][ci % 8] (the layout of the 32/64-channel kernels above) for output channel m.  No LDS: the 27-fold
// re-read of the input is served by L1/L2.  Leaves the InPlaceABN partial sums of its 32-voxel tiles when asked.
template <int CIN, int S>
__global__ __launch_bounds__(256) void conv3d_k3_mfma16_kernel(ActSrc a, int ld, int Di, int Hi, int Wi, const float* __restrict__ w32,
                                                              float* __restrict__ out, int Do, int Ho, int Wo, float* __restrict__ stats,
                                                              const int* __restrict__ run_if = nullptr)
{
    if (run_if && *run_if == 0) return;                          // fp32 half of a guarded sequence (mvsnerf_conv3d_f16x3_guarded_fwd)
    constexpr int COUT = 16, CPL = CIN / 4, CB8 = CIN / 8;
    static_assert(CIN == 8 || CIN == 16, "8 or 16 input channels");
    typedef float fcpl __attribute__((ext_vector_type(CPL)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, kh = lane >> 4;
    const int64_t nvox = (int64_t)Do * Ho * Wo;
    const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
    if (tile * 32 >= nvox) return;
    int xs[2], ys[2], zs[2];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int64_t v = tile * 32 + h * 16 + m;
        live[h] = v < nvox;
        const int64_t vc = live[h] ? v : nvox - 1;
        xs[h] = (int)(vc % Wo); ys[h] = (int)((vc / Wo) % Ho); zs[h] = (int)(vc / ((int64_t)Wo * Ho));
    }
    const bool lazy = a.scale != nullptr;
    float sc[CPL], sh[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) { sc[j] = lazy ? a.scale[kh * CPL + j] : 1.0f; sh[j] = lazy ? a.shift[kh * CPL + j] : 0.0f; }
    f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    // this lane's weights of a tap: channels CPL kh .. CPL kh + CPL - 1 of output channel m
    const float* wl = w32 + ((int64_t)((CIN == 16 ? (kh >> 1) : 0) * COUT + m)) * 8 + (CIN == 16 ? (kh & 1) * 4 : kh * 2);
#pragma unroll 3
    for (int tap = 0; tap < 27; ++tap) {
        const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
        const fcpl bw = *reinterpret_cast<const fcpl*>(wl + (int64_t)tap * CB8 * COUT * 8);
        fcpl av[2];
        bool in[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int zi = zs[h] * S - 1 + dz, yi = ys[h] * S - 1 + dy, xi = xs[h] * S - 1 + dx;
            in[h] = live[h] && zi >= 0 && zi < Di && yi >= 0 && yi < Hi && xi >= 0 && xi < Wi;
            av[h] = *reinterpret_cast<const fcpl*>(a.x + (in[h] ? ((int64_t)zi * Hi + yi) * Wi + xi : 0) * ld + kh * CPL);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            fcpl v = av[h];
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                if (lazy) v[j] = act_apply(v[j], sc[j], sh[j]);
                if (!in[h]) v[j] = 0.0f;                           // zero padding of the ACTIVATED input
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[h] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[j], bw[j], acc[h], 0, 0, 0);
        }
    }
    // D: register r of lane (col m, group kh) = voxel 4 kh + r of the 16-voxel tile
    float ssum = 0.f, ssq = 0.f;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t ov = tile * 32 + h * 16 + 4 * kh + r;
            if (ov < nvox) { out[ov * COUT + m] = acc[h][r]; ssum += acc[h][r]; ssq = fmaf(acc[h][r], acc[h][r], ssq); }
        }
    if (stats) {
        ssum += __shfl_xor(ssum, 16); ssq += __shfl_xor(ssq, 16);
        ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);
        const int64_t ntiles = (nvox + 31) / 32;
        if (kh == 0) { stats[abn_part_at(0, m, COUT, tile, ntiles)] = ssum; stats[abn_part_at(1, m, COUT, tile, ntiles)] = ssq; }
    }
}

int mvs_conv3d_mfma32_tiles(int D, int H, int W, int stride)
{
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    return (int)mvs_cdiv((int64_t)Do * Ho * Wo, 32);
}

int mvs_conv3d_mfma32(const ActSrc& a, const ActSrc& b, int Cin, int cin_ld, int D, int H, int W, const float* w32, int Cout, int stride,
                      float* out, float* stats, hipStream_t st, const int* run_if)
{
    if (run_if && !(Cin == 8 && Cout == 16 && stride == 2)) return MVSNERF_EUNSUPPORTED;      // predication: conv1's kernel only
    if (b.x || (cin_ld & 3)) return MVSNERF_EUNSUPPORTED;
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const unsigned grid = mvs_cdiv((int64_t)Do * Ho * Wo, 32);
#define MVS_M32(CIN, COUT, S) conv3d_k3_mfma32_kernel<CIN, COUT, S><<<grid, 512, 0, st>>>(a, cin_ld, D, H, W, w32, out, Do, Ho, Wo, stats)
    const unsigned grid16 = (grid + 3) / 4;
    switch (Cin * 1000 + Cout * 10 + stride) {
        case 8 * 1000 + 16 * 10 + 2:  conv3d_k3_mfma16_kernel<8, 2><<<grid16, 256, 0, st>>>(a, cin_ld, D, H, W, w32, out, Do, Ho, Wo, stats, run_if); break;    // conv1
        case 16 * 1000 + 32 * 10 + 2: MVS_M32(16, 32, 2); break;     // conv3 (and the data gradient of conv9)
        case 32 * 1000 + 32 * 10 + 1: MVS_M32(32, 32, 1); break;     // conv4 (and its data gradient)
        case 32 * 1000 + 64 * 10 + 2: MVS_M32(32, 64, 2); break;     // conv5 (and the data gradient of conv7)
        case 64 * 1000 + 64 * 10 + 1: MVS_M32(64, 64, 1); break;     // conv6 (and its data gradient)
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_M32
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// conv11-shaped transposed convolution (16 -> 8) without padded products (mvsnerf_conv_transpose3d_c8_fwd); wq: [ci/4][tap][co][4]
int mvs_convT3d_c16to8_tiles(int D, int H, int W) { return ((W + CT8_TX - 1) / CT8_TX) * ((H + CT8_TY - 1) / CT8_TY) * ((D + CT8_TZ - 1) / CT8_TZ); }

int mvs_convT3d_c16to8_mfma4(const ActSrc& xa, const ActSrc& xb, int D, int H, int W, const float* wq, float* out, int xcd, float* stats, hipStream_t st)
{
    const unsigned grid = (unsigned)mvs_convT3d_c16to8_tiles(D, H, W);
    convT3d_k3s2_c16to8_mfma4_kernel<<<grid, 256, 0, st>>>(xa, xb, D, H, W, wq, out, xcd, stats);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// transposed convolution; plain (materialised) input x[D][H][W][Cin]
// rows of InPlaceABN partial sums the kernel leaves (stats != NULL): (M-tiles rounded up to the workgroup's TPW) x outer parity classes
int mvs_convT3d_mfma32_tiles(int Cin, int Cout, int D, int H, int W)
{
    const int64_t tiles = ((int64_t)D * H * W + 31) / 32;
    const int tpw = Cin == 64 ? 1 : (Cin == 32 ? 2 : 4);
    return (int)(mvs_cdiv(tiles, tpw) * tpw * (8 / (32 / Cout)));
}

int mvs_convT3d_mfma32(const float* x, int Cin, int D, int H, int W, const float* w32, int Cout, float* out, float* stats, hipStream_t st)
{
    const int64_t tiles = ((int64_t)D * H * W + 31) / 32;
#define MVS_T32(CIN, COUT, WV, TPW) convT3d_k3s2_mfma32_kernel<CIN, COUT, WV, TPW><<<dim3(mvs_cdiv(tiles, TPW), 8 / (32 / COUT)), 64 * WV * TPW, 0, st>>>(x, D, H, W, w32, out, stats)
    switch (Cin * 100 + Cout) {
        case 64 * 100 + 32: MVS_T32(64, 32, 8, 1); break;      // conv7:  8..64 units per tile
        case 32 * 100 + 16: MVS_T32(32, 16, 2, 2); break;      // conv9:  8..32
        case 16 * 100 + 8:  MVS_T32(16, 8, 1, 4); break;       // conv11: 8 | 16 -> one wave per tile, no reduction
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_T32
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

bool mvs_conv3d_mfma32_supported(int Cin, int Cout, int stride)
{
    const int k = Cin * 1000 + Cout * 10 + stride;
    return k == 8162 || k == 16322 || k == 32321 || k == 32642 || k == 64641;
}

int mvs_conv_w32_repack(const float* wpacked, float* w32, int Cin, int Cout, hipStream_t st)
{
    conv_w32_repack_kernel<<<mvs_cdiv((int64_t)27 * Cin * Cout, 256), 256, 0, st>>>(wpacked, w32, Cin, Cout);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// conv0's weight gradient from the blocked cost volume (mvsnerf_conv3d_c8_blocked_wgrad).  workspace: cap_parts + MVS_RED_SLICES rows
// of 8 * cin_real * 27 floats.
static int wgrad4_nx(int Cin, int D, int H, int W, int cap_parts)
{
    const int ncg = Cin / 4, nsplit = (ncg + 3) / 4;
    const int ntiles = ((W + GW_X - 1) / GW_X) * ((H + GW_Y - 1) / GW_Y) * ((D + GW_Z - 1) / GW_Z);
    int nx = (512 / nsplit) & ~7;                                     // two workgroups per CU; a multiple of 8 (see the kernel's tile order)
    while (nx > 8 && (nx > cap_parts || nx / 8 > (ntiles + 7) / 8)) nx -= 8;
    return nx;
}

int mvs_conv3d_c8_wgrad4_parts(int Cin, int D, int H, int W, int cap_parts) { return wgrad4_nx(Cin, D, H, W, cap_parts); }

int mvs_conv3d_c8_wgrad4(const float* x4, int Cin, int cin_real, int D, int H, int W, const float* g, float* gw, float* workspace, int cap_parts,
                         hipStream_t st)
{
    if ((Cin & 3) || cin_real > Cin || cin_real <= Cin - 4) return MVSNERF_EINVAL;
    const int ncg = Cin / 4, nsplit = (ncg + 3) / 4;
    const int nx = wgrad4_nx(Cin, D, H, W, cap_parts);
    if (nx > cap_parts) return MVSNERF_EINVAL;
    conv3d_k3s1_c8_wgrad4_kernel<<<dim3(nx, nsplit), 256, 0, st>>>(x4, g, D, H, W, ncg, cin_real, workspace);
    MVS_LAUNCH_CHECK();
    if (!gw) return MVSNERF_OK;                                       // partials left for mvsnerf_partial_sum_multi
    const int64_t n_out = (int64_t)8 * cin_real * 27;
    mvs_partial_sum(workspace, nx, n_out, workspace + (size_t)cap_parts * n_out, gw, st);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// conv0 on a cost volume in channel blocks of four (mvsnerf_conv3d_c8_blocked_fwd)
int mvs_conv3d_c8_mfma4_tiles(int D, int H, int W) { return ((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ); }

int mvs_conv3d_c8_mfma4(const float* x4, int Cin, int cin_real, int D, int H, int W, const float* wq, float* out, int xcd, float* stats, hipStream_t st,
                        const int* run_if)
{
    if ((int64_t)D * H * W * 4 >= (int64_t)1 << 31) return MVSNERF_EUNSUPPORTED;
    const unsigned grid = (unsigned)(((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ));
#define MVS_L4(CIN, CREAL) case CIN * 100 + CREAL: conv3d_k3s1_c8_mfma4_kernel<CIN, CREAL><<<grid, 256, 0, st>>>(x4, D, H, W, wq, out, xcd, stats, run_if); break
    switch (Cin * 100 + cin_real) {
        MVS_L4(32, 32); MVS_L4(36, 35); MVS_L4(40, 38); MVS_L4(44, 41); MVS_L4(44, 44); MVS_L4(48, 47); MVS_L4(52, 50); MVS_L4(56, 53); MVS_L4(56, 56);
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_L4
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

int mvs_conv_w4_repack(const float* wpacked, float* wq, int Cin, hipStream_t st)
{
    conv_w4_repack_kernel<<<mvs_cdiv((int64_t)27 * Cin * 8, 256), 256, 0, st>>>(wpacked, wq, Cin);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// Called by mvsnerf_conv3d_fwd (encoder.hip) for stride-1 layers with 8 output channels and a channel-last input.  Cin: padded
// channel count (multiple of 4), cin_real: how many of them exist (products with the zero padding are skipped).  Returns
// MVSNERF_EUNSUPPORTED for a channel count it is not instantiated for (the caller then takes the VALU kernel).
int mvs_conv3d_c8_mfma(const ActSrc& a, const ActSrc& b, int Cin, int cin_real, int cin_ld, int D, int H, int W, const float* wpacked, float* out,
                       int xcd, hipStream_t st)
{
    if (b.x) return MVSNERF_EUNSUPPORTED;                         // one lazily-activated source only (conv0's input is the raw cost volume)
    const bool blocked = cin_ld == -8;                            // input in channel blocks of 8: x[(Cin+7)/8][D*H*W][8]
    if ((int64_t)D * H * W * (blocked ? 8 : cin_ld) >= (int64_t)1 << 31) return MVSNERF_EUNSUPPORTED;      // 32-bit element offsets in the staging slots
    const unsigned grid = (unsigned)(((W + TX - 1) / TX) * ((H + TY - 1) / TY) * ((D + TZ - 1) / TZ));
#define MVS_L(CIN, CREAL) case CIN * 100 + CREAL:                                                                                             \
        if (blocked) conv3d_k3s1_c8_mfma_kernel<CIN, CREAL, true><<<grid, 256, 0, st>>>(a, cin_ld, D, H, W, wpacked, out, xcd);                \
        else conv3d_k3s1_c8_mfma_kernel<CIN, CREAL, false><<<grid, 256, 0, st>>>(a, cin_ld, D, H, W, wpacked, out, xcd);                       \
        break
    if (cin_real <= 0 || cin_real > Cin) cin_real = Cin;
    switch (Cin * 100 + cin_real) {
        // 32 + 3V real channels for V = 0..8 source views (padded to a multiple of 4), and the padded counts themselves
        MVS_L(32, 32); MVS_L(36, 35); MVS_L(36, 36); MVS_L(40, 38); MVS_L(40, 40); MVS_L(44, 41); MVS_L(44, 44);
        MVS_L(48, 47); MVS_L(48, 48); MVS_L(52, 50); MVS_L(52, 52); MVS_L(56, 53); MVS_L(56, 56);
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_L
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
