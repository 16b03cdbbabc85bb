// conv0 of CostRegNet (models.py:756: 3x3x3, stride 1, Cin = 32 + 3V -> 8 channels; 74.5 % of the encoder's FLOPs) on the bf16 matrix
// cores: the encoder side of the reference's `precision=16 if args.use_amp` (train_mvs_nerf_pl.py:317-318; BASELINE config 3 "bf16").
// Operands (cost volume, weights, output gradient) are rounded to bf16, products accumulate in fp32, everything around the
// convolutions (plane-sweep arithmetic, InPlaceABN statistics / apply, master weights, gradients) stays fp32 - the recipe of the
// bf16 MLP kernels (mlp_bf16.hip).
//
// Data layout.  The plane sweep writes the cost volume in channel BLOCKS of 16 bf16 values: x16[Cin16/16][D*H*W][16] (32 bytes per voxel
// and block; mvsnerf_planesweep_costvar_bf16_fwd).  One v_mfma_f32_16x16x32_bf16 multiplies 16 voxels (A rows) x 32 k-values by
// 32 k x 16 columns; a lane (m = lane & 15, kg = lane >> 4) feeds 8 consecutive k-values = one 16-byte half of a voxel's block, i.e. one
// ds_read_b128 / one 16-byte piece of a DMA'd tile, whatever the tap.  Only 8 of the 16 columns are real output channels: at 16x the fp32
// matrix rate the kernels are bound by moving the 0.45 GB volume, not by the matrix pipes (45 MFMAs per 16 voxels = 0.1 ms chip-wide).
#include "common.h"
#include "act.h"
#include "lds_dma.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ const f32x4 g_zero16b = {0.0f, 0.0f, 0.0f, 0.0f};      // what a DMA lane reads for a voxel outside the volume (zero padding)

__device__ __forceinline__ void dma16_gather_b(const void* lane_ptr, unsigned lds_byte_uniform)
{
    // lanes read 16 B each at their own address; LDS receives them at lds_byte_uniform + lane * 16 (M0 carries the LDS base)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(lane_ptr), "s"(lds_byte_uniform) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------ forward
// Workgroup = 4 waves = 4 x 8 x 16 output voxels (z, y, x); wave w owns plane z0 + w, M-tile t = row y0 + t (16 x).
// Per 16-channel block (chunk) the (6 x 10 x 18)-voxel halo is DMA'd into LDS as [voxel][32 B] (34 pieces of 1 KB) together with the
// chunk's 7.5 KB of weight fragments.  ONE 43 KB buffer per workgroup and three workgroups per CU: a chunk is issue -> wait -> barrier ->
// 120 MFMAs per wave -> barrier, and it is the other two workgroups' MFMAs that cover the DMA round trip.  (First version: two buffers per
// workgroup with chunk c+1 in flight during chunk c - 86 KB, i.e. ONE workgroup of four waves per CU: 0.42 ms, every chunk waiting ~4 us for
// its DMA with nothing else resident to run.)  K = 32 of one MFMA = two (dz, dx) taps x 16 channels: fragment f = 0..4 pairs the (dz, dx)
// combinations p = 2f, 2f+1 (p = 3 dz + dx; p = 9 does not exist: zero weights).  For a fixed f the A fragment of INPUT row j serves
// the three dy taps of the M-tiles t = j - dy: 10 operand reads + 2 weight reads per 17 MFMAs (dy 0 and dy 1 share one MFMA through the two halves
// of the 16 B columns).
constexpr int BTX = 16, BTY = 8, BTZ = 4;
constexpr int BPX = BTX + 2, BPY = BTY + 2, BPZ = BTZ + 2;
constexpr int BNV = BPX * BPY * BPZ;                            // 1080 voxels
constexpr int BT_PIECES = (BNV * 32 + 1023) / 1024;             // 34 DMA pieces per chunk
constexpr int BT_BYTES = BT_PIECES * 1024;
constexpr int BT_SLOTS = (BT_PIECES + 3) / 4;                   // 9 per wave
constexpr int BW_BYTES = 5 * 3 * 4 * 8 * 16;                    // weights of a chunk: [f][dy][kg][co 8][8 ci] bf16 = 7680 B
constexpr int BW_PIECES = (BW_BYTES + 1023) / 1024;             // 8
constexpr int BBUF = BT_BYTES + BW_PIECES * 1024;               // 43008 B

__global__ __launch_bounds__(256, 3) void conv3d_k3s1_c8_bf16_kernel(const __bf16* __restrict__ x16, int nblk16, int D, int H, int W,
                                                                    const __bf16* __restrict__ wq, float* __restrict__ out,
                                                                    float* __restrict__ stats)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int nbx = (W + BTX - 1) / BTX, nby = (H + BTY - 1) / BTY;
    const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, bz = tile_id / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = bx * BTX - 1, y0 = by * BTY - 1, z0 = bz * BTZ - 1;
    const int64_t nvox = (int64_t)D * H * W;
    // DMA slots of this lane: piece p = wave + 4 j holds tile voxels 32 p .. 32 p + 31, lane -> (voxel 32 p + lane / 2, half lane & 1)
    int goff[BT_SLOTS];                                          // byte offset inside a channel block, -1: zeros
#pragma unroll
    for (int j = 0; j < BT_SLOTS; ++j) {
        const int v = (wave + 4 * j) * 32 + (lane >> 1);
        const int vx = v % BPX, vy = (v / BPX) % BPY, vz = v / (BPX * BPY);
        const int gx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        const bool in = v < BNV && gx >= 0 && gx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        goff[j] = in ? (((gz * H + gy) * W + gx) * 32 + (lane & 1) * 16) : -1;
    }
    const char* zero16 = reinterpret_cast<const char*>(&g_zero16b);
    auto issue = [&](int c, char* dst) {
        const char* xb = reinterpret_cast<const char*>(x16) + (int64_t)c * nvox * 32;
        const unsigned base = lds_byte_addr(dst);
#pragma unroll
        for (int j = 0; j < BT_SLOTS; ++j) {
            const int p = wave + 4 * j;
            if (p < BT_PIECES) dma16_gather_b(goff[j] >= 0 ? xb + goff[j] : zero16, base + p * 1024);
        }
        const char* wb = reinterpret_cast<const char*>(wq) + (int64_t)c * BW_BYTES;
#pragma unroll
        for (int j = 0; j < 2; ++j) {                            // 8 weight pieces: two per wave (the tail piece is partly padding)
            const int p = wave + 4 * j, off = p * 1024 + lane * 16;
            dma16_gather_b(off < BW_BYTES ? wb + off : zero16, base + BT_BYTES + p * 1024);
        }
    };
    // Accumulators P[0..8], one per INPUT row j: B columns 0..7 carry the dy = 0 weights and columns 8..15 the dy = 1 weights, so one MFMA of input
    // row j yields [dy 0 -> output row j | dy 1 -> output row j - 1]; a second MFMA with B = [dy 2 | 0] adds row j's share of output row j - 2 to the
    // left half of P[j - 2].  Output row t = left(P[t]) + right(P[t + 1]): 17 MFMAs per fragment group instead of 24 (conv_f16x3.hip: same scheme).
    f32x4 P[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) P[t] = f32x4{0, 0, 0, 0};
    const int m = lane & 15, kg = lane >> 4;
    const bool left = m < 8;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int c = 0; c < nblk16; ++c) {
        char* cur = lds;
        issue(c, cur);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's DMA pieces have landed ...
        __syncthreads();                                          // ... everybody's have
        const char* wt01 = cur + BT_BYTES + (m >> 3) * 512 + (kg * 8 + (m & 7)) * 16;      // columns 0..7: dy 0, columns 8..15: dy 1
        const char* wt2 = cur + BT_BYTES + 2 * 512 + (kg * 8 + (m & 7)) * 16;              // dy 2 (right half zeroed)
#pragma unroll
        for (int f = 0; f < 5; ++f) {
            const int p = 2 * f + (kg >> 1) < 9 ? 2 * f + (kg >> 1) : 8;     // the missing tenth (dz, dx) pair re-reads the ninth; its weights are zero
            const int dz = p / 3, dx = p - 3 * dz;
            const char* al = cur + (((wave + dz) * BPY) * BPX + m + dx) * 32 + (kg & 1) * 16;
            bf16x8 av[10];
            const bf16x8 b01 = *reinterpret_cast<const bf16x8*>(wt01 + f * 3 * 512);
            const bf16x8 b2 = left ? *reinterpret_cast<const bf16x8*>(wt2 + f * 3 * 512) : zero8;
#pragma unroll
            for (int j = 0; j < 10; ++j) av[j] = *reinterpret_cast<const bf16x8*>(al + j * BPX * 32);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                if (j < 9) P[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[j], b01, P[j], 0, 0, 0);
                if (j >= 2) P[j - 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[j], b2, P[j - 2], 0, 0, 0);
            }
        }
        __syncthreads();                                          // everybody is done reading the buffer
    }
    // D: lane (column n = lane & 15, g = lane >> 4): register r = voxel x 4 g + r of the M-tile; left(P[t]) sits in lane n, right(P[t + 1]) in
    // lane n + 8 of the same 16-lane row (DPP row rotation by 8).  As in conv_f16x3.hip the values leave through a wave-private LDS stage (the tile buffer is
    // free: barrier above) as 1 KB row stores instead of 32 half-masked dword stores per wave; the epilogue's indices come from an opaque copy of the thread
    // index (nothing to hoist above the main loop); the InPlaceABN partial sums are taken in the order the direct stores took them.
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6;
    const int n = lane_e & 15, g4 = lane_e >> 4;
    const int oz = bz * BTZ + wave_e;
    float ssum = 0.f, ssq = 0.f;
    constexpr int RS = 16 * 8 + 4 * 8;                            // floats per staged row
    float* stg = reinterpret_cast<float*>(lds) + wave_e * (8 * RS) + g4 * 40 + n;
    const bool mine = oz < D && n < 8;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int oy = by * BTY + t;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = P[t][r] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(P[t + 1][r]), 0x128, 0xf, 0xf, false));   // row_ror:8
            if (n < 8) stg[t * RS + r * 8] = v;
            if (mine && bx * BTX + g4 * 4 + r < W && oy < H) { ssum += v; ssq = fmaf(v, v, ssq); }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // wave-private: no barrier
    {
        const int xr = (lane_e & 31) >> 1, hq = lane_e & 1;
        const float* rd = reinterpret_cast<const float*>(lds) + wave_e * (8 * RS) + (lane_e >> 5) * RS + xr * 8 + (xr >> 2) * 8 + hq * 4;
        const int ox = bx * BTX + xr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(rd + 2 * k * RS);
            const int oy = by * BTY + 2 * k + (lane_e >> 5);
            if (oz < D && oy < H && ox < W) *reinterpret_cast<f32x4*>(out + (((int64_t)oz * H + oy) * W + ox) * 8 + hq * 4) = v4;
        }
    }
    if (stats) {          // InPlaceABN partial sums of this tile: abn_part_at(...) of common.h, slot = tile (abn_finalize_kernel's layout)
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
        ssum += __shfl_xor(ssum, 16); ssq += __shfl_xor(ssq, 16);
        ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);
        if (lane < 8) { red[wave * 16 + lane] = ssum; red[wave * 16 + 8 + lane] = ssq; }
        __syncthreads();
        if (wave == 0 && lane < 16) {
            const float v = (red[lane] + red[16 + lane]) + (red[32 + lane] + red[48 + lane]);
            stats[abn_part_at(lane >> 3, lane & 7, 8, tile_id, gridDim.x)] = v;
        }
    }
}

// nn.Conv3d weight w[8][Cin][3][3][3] (fp32) -> wq[chunk][f][dy][kg][co][8] bf16, the B fragments of the kernel above
__global__ __launch_bounds__(256) void conv0_pack_bf16_kernel(const float* __restrict__ w, int Cin, int nblk16, __bf16* __restrict__ wq)
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
        v = w[((int64_t)co * Cin + ci) * 27 + (dz * 3 + dy) * 3 + dx];
    }
    wq[i] = (__bf16)v;
}

}  // namespace

extern "C" size_t mvsnerf_conv0_bf16_packed_elems(int Cin)
{
    if (Cin < 1) return 0;
    return (size_t)((Cin + 15) / 16) * 5 * 3 * 4 * 8 * 8;
}

extern "C" int mvsnerf_conv0_bf16_pack(const float* w, int Cin, void* packed, void* stream)
{
    if (!w || !packed || Cin < 1) return MVSNERF_EINVAL;
    const int nblk = (Cin + 15) / 16;
    conv0_pack_bf16_kernel<<<mvs_cdiv((int64_t)nblk * 3840, 256), 256, 0, (hipStream_t)stream>>>(w, Cin, nblk, reinterpret_cast<__bf16*>(packed));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_conv0_bf16_tiles(int D, int H, int W) { return ((W + BTX - 1) / BTX) * ((H + BTY - 1) / BTY) * ((D + BTZ - 1) / BTZ); }

extern "C" int mvsnerf_conv0_bf16_fwd(const void* x16, int Cin, int D, int H, int W, const void* packed, float* out, float* stats_part, void* stream)
{
    if (!x16 || !packed || !out || Cin < 1 || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x16) || !mvs_aligned16(packed)) return MVSNERF_EALIGN;
    if ((int64_t)D * H * W * 32 >= ((int64_t)1 << 31)) return MVSNERF_EUNSUPPORTED;
    static unsigned long long cap_mask = 0;
    if (int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(conv3d_k3s1_c8_bf16_kernel), BBUF, &cap_mask)) return rc;
    conv3d_k3s1_c8_bf16_kernel<<<mvsnerf_conv0_bf16_tiles(D, H, W), 256, BBUF, (hipStream_t)stream>>>(
        reinterpret_cast<const __bf16*>(x16), (Cin + 15) / 16, D, H, W, reinterpret_cast<const __bf16*>(packed), out, stats_part);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ------------------------------------------------------------------------------------------------------------------ data gradient
// d cost[v][ci] = sum_{tap, co} g[v - tap + 1][co] * w[co][ci][tap] for the `nci` (multiple of 16, <= 32) input channels starting at c_first -
// the plane sweep's backward needs the 32 variance channels only (the warped thumbnails have no parameters upstream).
// g: gradient of conv0's RAW output, fp32 channel-last [D][H][W][8] (what the InPlaceABN backward writes); it is DMA'd as fp32
// ([voxel][32 B] tiles, one buffer pair) and rounded to bf16 when a lane reads its operand: no separate conversion pass.
// MFMA: A rows = 16 voxels (x), K = 32 = four (dz, dx) taps x 8 output channels, B columns = 16 input channels; fragment f = 0..2 holds
// the (dz, dx) combinations p = 4f..4f+3 (p >= 9: zero weights).  As in the forward kernel the A fragment of g-row j serves the dy taps
// of the output rows t = j + dy - 2 ... (mirrored: the data gradient is the correlation with the flipped kernel); all 18 weight fragments
// (3 dy x 3 f x 2 column blocks) stay in registers for the whole kernel.
namespace {

constexpr int GT_PIECES = (BNV * 32 + 1023) / 1024;              // fp32 g tile: 1080 voxels x 32 B = 34 pieces
constexpr int GT_BYTES = GT_PIECES * 1024;

template <int NCB>       // column blocks of 16 input channels (1 or 2)
__global__ __launch_bounds__(256, 2) void conv3d_k3s1_c8_bf16_dgrad_kernel(const float* __restrict__ g, int D, int H, int W,
                                                                          const __bf16* __restrict__ wd, float* __restrict__ gx)
{
    __shared__ __attribute__((aligned(1024))) char lds[GT_BYTES];
    const int nbx = (W + BTX - 1) / BTX, nby = (H + BTY - 1) / BTY;
    const int tile_id = xcd_contiguous_tile(blockIdx.x, gridDim.x);
    const int bx = tile_id % nbx, by = (tile_id / nbx) % nby, bz = tile_id / (nbx * nby);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int x0 = bx * BTX - 1, y0 = by * BTY - 1, z0 = bz * BTZ - 1;
    const char* zero16 = reinterpret_cast<const char*>(&g_zero16b);
    const unsigned base = lds_byte_addr(lds);
#pragma unroll
    for (int j = 0; j < BT_SLOTS; ++j) {
        const int p = wave + 4 * j;
        const int v = p * 32 + (lane >> 1);
        const int vx = v % BPX, vy = (v / BPX) % BPY, vz = v / (BPX * BPY);
        const int gxx = x0 + vx, gy = y0 + vy, gz = z0 + vz;
        const bool in = v < BNV && gxx >= 0 && gxx < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
        if (p < GT_PIECES)
            dma16_gather_b(in ? reinterpret_cast<const char*>(g) + ((int64_t)((gz * H + gy) * W + gxx) * 32 + (lane & 1) * 16) : zero16, base + p * 1024);
    }
    // weights: wd[f][dy][cb][kg][ci 16][8 co] bf16, B fragment of lane (n = lane & 15, kg): 16 bytes
    const int n = lane & 15, kg = lane >> 4;
    bf16x8 bw[3][3][NCB];
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
                bw[f][dy][cb] = *reinterpret_cast<const bf16x8*>(wd + ((((f * 3 + dy) * NCB + cb) * 4 + kg) * 16 + n) * 8);
    f32x4 acc[8][NCB];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[t][cb] = f32x4{0, 0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        const int p = 4 * f + kg < 9 ? 4 * f + kg : 8;
        const int dz = p / 3, dx = p - 3 * dz;
        // output voxel (z, y, x) at tap (dz, dy, dx) reads g at (z - dz + 1, y - dy + 1, x - dx + 1): halo index (wave + 2 - dz, t + 2 - dy, m + 2 - dx)
        const char* al = lds + ((((wave + 2 - dz) * BPY) * BPX) + (lane & 15) + 2 - dx) * 32;
        bf16x8 av[10];
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const f32x4 lo = *reinterpret_cast<const f32x4*>(al + j * BPX * 32), hi = *reinterpret_cast<const f32x4*>(al + j * BPX * 32 + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { av[j][e] = (__bf16)lo[e]; av[j][4 + e] = (__bf16)hi[e]; }
        }
#pragma unroll
        for (int j = 0; j < 10; ++j)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int t = j + dy - 2;                          // g-row j = t + 2 - dy
                if (t >= 0 && t < 8) {
#pragma unroll
                    for (int cb = 0; cb < NCB; ++cb) acc[t][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[j], bw[f][dy][cb], acc[t][cb], 0, 0, 0);
                }
            }
    }
    // Stored from the D fragment a lane writes one float per instruction (64 instructions per wave, each four 64-byte pieces a row apart) for the 600 MB this
    // kernel writes.  A row of the wave's plane is 16 voxels x 16 NCB channels = 1 or 2 KB of CONTIGUOUS output: two rows at a time go through a wave-private stage
    // in the (free: barrier) g tile - padded by 16 floats per 4 voxels, so that the four voxel groups of a write fall on different banks - and leave as 1 KB
    // global_store_dwordx4.  (The epilogue's indices come from an opaque copy of the thread index: nothing of it can be hoisted above the MFMA loops.)
    __syncthreads();                                              // every wave has read its A operands: the g tile is free
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, wave_e = tid_e >> 6;
    const int n_e = lane_e & 15, g4 = lane_e >> 4, oz = bz * BTZ + wave_e;
    constexpr int CW = 16 * NCB;                                  // channels per voxel
    constexpr int RS = 16 * CW + 4 * 16;                          // floats per staged row
    constexpr int QV = CW / 4;                                    // 16-byte chunks per voxel
    float* stg = reinterpret_cast<float*>(lds) + wave_e * (2 * RS);
#pragma unroll
    for (int t2 = 0; t2 < 8; t2 += 2) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb) stg[tt * RS + (g4 * 4 + r) * CW + g4 * 16 + cb * 16 + n_e] = acc[t2 + tt][cb][r];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // wave-private: no barrier
#pragma unroll
        for (int k = 0; k < (2 * 16 * QV) / 64; ++k) {
            const int c = k * 64 + lane_e;                        // 16-byte chunk of the two rows
            const int tt = c / (16 * QV), cc = c - tt * (16 * QV), x = cc / QV, q = cc - x * QV;
            const f32x4 v4 = *reinterpret_cast<const f32x4*>(stg + tt * RS + x * CW + (x >> 2) * 16 + q * 4);
            const int oy = by * BTY + t2 + tt, ox = bx * BTX + x;
            if (oz < D && oy < H && ox < W) *reinterpret_cast<f32x4*>(gx + (((int64_t)oz * H + oy) * W + ox) * CW + q * 4) = v4;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the rows are read before the next pair overwrites them
    }
}

// w[8][Cin][3][3][3] -> wd[f][dy][cb][kg][ci 16][8 co] bf16 for input channels c_first + (cb*16 + ci)
__global__ __launch_bounds__(256) void conv0_pack_bf16_dgrad_kernel(const float* __restrict__ w, int Cin, int c_first, int ncb, __bf16* __restrict__ wd)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int total = 3 * 3 * ncb * 4 * 16 * 8;
    if (i >= total) return;
    const int co = i & 7, ci = (i >> 3) & 15, kg = (i >> 7) & 3;
    const int rest = i >> 9, cb = rest % ncb, dy = (rest / ncb) % 3, f = rest / (ncb * 3);
    const int p = 4 * f + kg, c = c_first + cb * 16 + ci;
    float v = 0.0f;
    if (p < 9 && c < Cin) {
        const int dz = p / 3, dx = p - 3 * dz;
        v = w[((int64_t)co * Cin + c) * 27 + (dz * 3 + dy) * 3 + dx];
    }
    wd[i] = (__bf16)v;
}

}  // namespace

extern "C" size_t mvsnerf_conv0_bf16_dgrad_packed_elems(int n_ci) { return (n_ci == 16 || n_ci == 32) ? (size_t)9 * (n_ci / 16) * 4 * 16 * 8 : 0; }

extern "C" int mvsnerf_conv0_bf16_dgrad_pack(const float* w, int Cin, int c_first, int n_ci, void* packed, void* stream)
{
    if (!w || !packed || Cin < 1 || c_first < 0 || c_first + n_ci > Cin) return MVSNERF_EINVAL;
    if (n_ci != 16 && n_ci != 32) return MVSNERF_EUNSUPPORTED;
    const int ncb = n_ci / 16;
    conv0_pack_bf16_dgrad_kernel<<<mvs_cdiv(9 * ncb * 512, 256), 256, 0, (hipStream_t)stream>>>(w, Cin, c_first, ncb, reinterpret_cast<__bf16*>(packed));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_conv0_bf16_dgrad(const float* g, int D, int H, int W, const void* packed, int n_ci, float* gx, void* stream)
{
    if (!g || !packed || !gx || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (n_ci != 16 && n_ci != 32) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(g) || !mvs_aligned16(packed)) return MVSNERF_EALIGN;
    if ((int64_t)D * H * W * 32 >= ((int64_t)1 << 31)) return MVSNERF_EUNSUPPORTED;
    const int tiles = mvsnerf_conv0_bf16_tiles(D, H, W);
    hipStream_t st = (hipStream_t)stream;
    if (n_ci == 32) conv3d_k3s1_c8_bf16_dgrad_kernel<2><<<tiles, 256, 0, st>>>(g, D, H, W, reinterpret_cast<const __bf16*>(packed), gx);
    else conv3d_k3s1_c8_bf16_dgrad_kernel<1><<<tiles, 256, 0, st>>>(g, D, H, W, reinterpret_cast<const __bf16*>(packed), gx);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ------------------------------------------------------------------------------------------------------------------ weight gradient
// gw[co][ci][tap] = sum_v g[v][co] * x[v + tap - 1][ci]: the reduction runs over VOXELS, so both MFMA operands want eight consecutive
// voxels of one channel per lane while the data is channel-last in HBM and LDS.  gfx950's transposing LDS read (ds_read_b64_tr_b16: the
// 16 lanes of a group fetch a [4 rows][16 columns] block of 16-bit values, 8 bytes per lane, and lane i gets COLUMN i) turns
// [voxel][16 channels] rows into exactly that without touching the layout:
//   A = x^T  rows: 16 input channels of block cb, k: 32 consecutive voxels along x (two transposing reads per lane)
//   B = g    k: the same 32 voxels,               columns: 8 output channels (+ 8 columns fetched from a block of zeros)
// A workgroup (4 waves) owns ONE channel block cb and walks a range of (4 x 6 x 32)-voxel tiles; wave w owns the g-plane z0 + w and keeps
// its six g-row fragments in registers; an x-row fragment (plane zz = w + dz, halo row yy, shift dx) is read once and feeds the up to
// three taps dy whose g-row yy - dy lies in the tile.  27 accumulators per wave persist over the whole range; at the end the four waves
// are summed in a fixed order and the workgroup writes ITS channel block of partial result `range` (gw's own [co][ci][tap] layout):
// deterministic, reduced by mvs_partial_sum like every other weight-gradient kernel.
namespace {

constexpr int WTX = 32, WTY = 6, WTZ = 4;
constexpr int WPX = WTX + 2, WPY = WTY + 2, WPZ = WTZ + 2;
constexpr int WNV = WPX * WPY * WPZ;                             // 1632 halo voxels of 32 B
constexpr int WX_PIECES = (WNV * 32 + 1023) / 1024;              // 51
constexpr int WX_BYTES = WX_PIECES * 1024;
constexpr int WX_SLOTS = (WX_PIECES + 3) / 4;                    // 13 per wave
constexpr int WG_VOX = WTX * WTY * WTZ;                          // 768 g voxels of 16 B (bf16 x 8)
constexpr int WG_BYTES = WG_VOX * 16;
constexpr int W_LDS = WX_BYTES + WG_BYTES + 64;                  // + a block of zeros for the eight padding columns of g

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// lane i of a 16-lane group passes the address of row (i >> 2), columns 4 (i & 3) .. + 3 of a [4][16] block and receives column i
__device__ __forceinline__ bf16x4 tr_read(const char* p)
{
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const void*)p);
    return __builtin_bit_cast(bf16x4, v);
}

__global__ __launch_bounds__(256, 2) void conv3d_k3s1_c8_bf16_wgrad_kernel(const __bf16* __restrict__ x16, const float* __restrict__ g,
                                                                          int Cin, int D, int H, int W, int n_ranges, float* __restrict__ partial)
{
    __shared__ __attribute__((aligned(1024))) char lds[W_LDS];
    char* xt = lds;
    char* gt = lds + WX_BYTES;
    char* zt = gt + WG_BYTES;                                     // 64 bytes of zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int range = blockIdx.x % n_ranges, cb = blockIdx.x / n_ranges;
    const int nbx = (W + WTX - 1) / WTX, nby = (H + WTY - 1) / WTY, nbz = (D + WTZ - 1) / WTZ;
    const int n_tiles = nbx * nby * nbz;
    const int t_begin = (int)((int64_t)n_tiles * range / n_ranges), t_end = (int)((int64_t)n_tiles * (range + 1) / n_ranges);
    const int64_t nvox = (int64_t)D * H * W;
    const char* xb = reinterpret_cast<const char*>(x16) + (int64_t)cb * nvox * 32;
    const char* zero16 = reinterpret_cast<const char*>(&g_zero16b);
    if (tid < 16) reinterpret_cast<float*>(zt)[tid] = 0.0f;
    f32x4 acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = f32x4{0, 0, 0, 0};
    const int i16 = lane & 15, kg = lane >> 4;
    // per-lane parts of the transposing reads: this lane passes row (8 kg + (i16 >> 2) [+ 4]) of the fragment, column chunk (i16 & 3).
    // The two reads of a fragment take the rows 8 kg + 0..3 and 8 kg + 4..7; ODD lane groups take them in the opposite order (for A and B alike, so the
    // k-index of an element is the same on both sides): one read instruction then covers 128-byte blocks at 0, 384, 512, 896 instead of 0, 256, 512, 768 -
    // both halves of the 64 banks instead of one (round 4 PMC: LDS bank conflict rate 0.47 = every x-fragment read took four cycles instead of two).
    const int first4 = (kg & 1) * 4;
    const int row_in_frag = 8 * kg + (i16 >> 2) + first4, row_step = (kg & 1) ? -4 : 4, chunk = i16 & 3;
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int x0 = bx * WTX - 1, y0 = by * WTY - 1, z0 = bz * WTZ - 1;
        __syncthreads();                                          // everybody is done with the previous tile
        {   // x halo: 51 DMA pieces of 32 voxels
            const unsigned base = lds_byte_addr(xt);
#pragma unroll
            for (int j = 0; j < WX_SLOTS; ++j) {
                const int p = wave + 4 * j;
                const int v = p * 32 + (lane >> 1);
                const int vx = v % WPX, vy = (v / WPX) % WPY, vz = v / (WPX * WPY);
                const int gx_ = x0 + vx, gy = y0 + vy, gz = z0 + vz;
                const bool in = v < WNV && gx_ >= 0 && gx_ < W && gy >= 0 && gy < H && gz >= 0 && gz < D;
                if (p < WX_PIECES) dma16_gather_b(in ? xb + ((int64_t)((gz * H + gy) * W + gx_) * 32 + (lane & 1) * 16) : zero16, base + p * 1024);
            }
        }
        // g tile: fp32 [voxel][8] -> bf16 [voxel][8] (16 B rows), 3 voxels per thread
#pragma unroll
        for (int j = 0; j < WG_VOX / 256; ++j) {
            const int v = tid + 256 * j;
            const int vx = v % WTX, vy = (v / WTX) % WTY, vz = v / (WTX * WTY);
            const int gx_ = x0 + 1 + vx, gy = y0 + 1 + vy, gz = z0 + 1 + vz;
            bf16x8 h;
            if (gx_ < W && gy < H && gz < D) {
                const float* src = g + ((int64_t)(gz * H + gy) * W + gx_) * 8;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(src), hi = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { h[e] = (__bf16)lo[e]; h[4 + e] = (__bf16)hi[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) h[e] = (__bf16)0.0f;
            }
            *reinterpret_cast<bf16x8*>(gt + v * 16) = h;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // B fragments: the six g rows of this wave's plane
        bf16x8 bg[WTY];
#pragma unroll
        for (int y = 0; y < WTY; ++y) {
            const char* rowp = gt + ((wave * WTY + y) * WTX + row_in_frag) * 16 + chunk * 8;
            const char* p0 = chunk < 2 ? rowp : zt, * p1 = chunk < 2 ? rowp + row_step * 16 : zt;
            const bf16x4 lo = tr_read(p0), hi = tr_read(p1);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bg[y][e] = lo[e]; bg[y][4 + e] = hi[e]; }
        }
#pragma unroll
        for (int dz = 0; dz < 3; ++dz)
#pragma unroll
            for (int yy = 0; yy < WPY; ++yy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const char* rowp = xt + ((((wave + dz) * WPY + yy) * WPX) + dx + row_in_frag) * 32 + chunk * 8;
                    const bf16x4 lo = tr_read(rowp), hi = tr_read(rowp + row_step * 32);
                    bf16x8 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = lo[e]; a[4 + e] = hi[e]; }
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int y = yy - dy;
                        if (y >= 0 && y < WTY) acc[(dz * 3 + dy) * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bg[y], acc[(dz * 3 + dy) * 3 + dx], 0, 0, 0);
                    }
                }
    }
    // fixed-order sum of the four waves (red[tap][lane][r]), then this channel block of partial `range`
    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll 1
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int k = 0; k < 27; ++k) {
                f32x4* dst = reinterpret_cast<f32x4*>(red + (k * 64 + lane) * 4);
                *dst = w == 0 ? acc[k] : (*dst + acc[k]);
            }
        }
        __syncthreads();
    }
    float* pr = partial + (int64_t)range * 8 * Cin * 27;
    for (int idx = tid; idx < 27 * 16 * 8; idx += 256) {
        const int co = idx & 7, ci = (idx >> 3) & 15, tap = idx >> 7;
        const int c = cb * 16 + ci;
        if (c < Cin) pr[((int64_t)co * Cin + c) * 27 + tap] = red[(tap * 64 + ((ci >> 2) * 16 + co)) * 4 + (ci & 3)];     // D: lane (col co, group ci / 4), register ci % 4
    }
}

}  // namespace

extern "C" int mvsnerf_conv0_bf16_wgrad_parts(int D, int H, int W)
{
    const int n_tiles = ((W + WTX - 1) / WTX) * ((H + WTY - 1) / WTY) * ((D + WTZ - 1) / WTZ);
    return n_tiles < 170 ? n_tiles : 170;                         // x 3 channel blocks = 510 workgroups = two per CU
}

// workspace: mvsnerf_conv3d_wgrad_workspace_floats(8, Cin) floats (the partial results lie at its start, rows of 8 * Cin * 27 floats);
// gw NULL: leave them there for mvsnerf_partial_sum_multi
extern "C" int mvsnerf_conv0_bf16_wgrad(const void* x16, int Cin, int D, int H, int W, const float* g, float* gw, float* workspace, void* stream)
{
    if (!x16 || !g || !workspace || Cin < 1 || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (!mvs_aligned16(x16) || !mvs_aligned16(g)) return MVSNERF_EALIGN;
    if ((int64_t)D * H * W * 32 >= ((int64_t)1 << 31)) return MVSNERF_EUNSUPPORTED;
    const int n_ranges = mvsnerf_conv0_bf16_wgrad_parts(D, H, W), nb16 = (Cin + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    conv3d_k3s1_c8_bf16_wgrad_kernel<<<n_ranges * nb16, 256, 0, st>>>(reinterpret_cast<const __bf16*>(x16), g, Cin, D, H, W, n_ranges, workspace);
    MVS_LAUNCH_CHECK();
    if (!gw) return MVSNERF_OK;
    const int64_t n_out = (int64_t)8 * Cin * 27;
    mvs_partial_sum(workspace, n_ranges, n_out, workspace + (size_t)2048 * n_out, gw, st);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
