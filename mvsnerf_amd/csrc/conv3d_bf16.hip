// conv1 ... conv11 of CostRegNet (models.py:725-769: 3x3x3, stride 1 / 2, and the three stride-2 transposed layers) on the bf16 matrix cores:
// the rest of the encoder side of the reference's `precision=16 if args.use_amp` (train_mvs_nerf_pl.py:317-318; BASELINE config 3 "bf16") next
// to conv0 (conv_bf16.hip).  Forward AND data gradients: a data gradient is the same convolution with re-packed weights (mirrored taps for
// stride 1; stride-2 conv <-> transposed conv), exactly as on the fp32 path.  Operands are rounded to bf16 (round to nearest even) on their way
// into the A / B fragments, products accumulate in fp32, activations stay fp32 in HBM with their pending InPlaceABN applied on load, the
// InPlaceABN partial sums of the output leave with the same launch.
//
// One v_mfma_f32_16x16x32_bf16 multiplies 16 output voxels (A rows) x 32 k-values by 32 k x 16 output channels.  k enumerates (tap, input
// channel) with the channel fastest; a lane (m = lane & 15, kg = lane >> 4) feeds the 8 consecutive k-values 32 ks + 8 kg .. + 7, i.e. eight
// consecutive channels of ONE tap of its voxel (Cin is 8, 16, 32 or 64, so a group of eight never straddles taps): two 16-byte loads of the
// channel-last input, straight from L1 / L2 (these volumes are at most 37 MB; the 27-fold re-read never reaches HBM), activated, rounded, used.
// Weights are packed once per weight change into fragment order [k-step][16-column block][lane][8] (16 bytes per lane and MFMA).
// A wave owns an M-tile of 16 voxels and ALL output channels (1, 2 or 4 column blocks); four waves per workgroup share the statistics slot.
//
// Transposed layers: out[o] = sum over taps k with o = 2 i - 1 + k; per dimension an output of parity 0 (o = 2 i) has one tap (kernel index
// 1, input i), one of parity 1 two (kernel index 0 from input i + 1, kernel index 2 from input i).  An M-tile = 16 consecutive positions i of
// the INPUT lattice; the columns of the MFMA are the output channels of BOTH x parities (column = px * Cout + channel: the two output voxels
// 2 ix, 2 ix + 1 of a position are 2 Cout contiguous floats and consecutive positions are contiguous, so the stores are full lines; a
// column whose parity has no tap at an x offset carries a zero weight: 25 % of the products), k enumerates (z sub-tap, y sub-tap, x offset,
// channel), and the wave walks the four (z, y) parity classes itself - 1 + 2 + 2 + 4 tap pairs - with the index arithmetic done once.  (First
// version: one launch class per (pz, py, px), 8 output channels in 16 columns, 4-byte stores at a two-voxel stride: conv11 272 us against
// 112 us on the fp32 matrix-core kernel.)
#include "common.h"
#include "act.h"
#include "conv3d_bf16_layout.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int ilog2c(int v) { return v <= 1 ? 0 : 1 + ilog2c(v >> 1); }

// lazily-activated operand tables in LDS: [0] scale a, [1] shift a, [2] scale b, [3] shift b
template <int CIN>
__device__ __forceinline__ void stage_act(const ActSrc& a, const ActSrc& b, float (*act)[CIN], int tid)
{
    for (int c = tid; c < CIN; c += 256) {
        act[0][c] = a.scale ? a.scale[c] : 1.0f; act[1][c] = a.scale ? a.shift[c] : 0.0f;
        act[2][c] = (b.x && b.scale) ? b.scale[c] : 1.0f; act[3][c] = (b.x && b.scale) ? b.shift[c] : 0.0f;
    }
    __syncthreads();
}

// eight consecutive channels c0 .. c0 + 7 at float offset `off` of the (first + second) source, activated, rounded to bf16; zeros when !in
template <int CIN>
__device__ __forceinline__ bf16x8 load_a8(const ActSrc& a, const ActSrc& b, const float (*act)[CIN], int64_t off, int c0, bool in)
{
    f32x4 v0 = *reinterpret_cast<const f32x4*>(a.x + off), v1 = *reinterpret_cast<const f32x4*>(a.x + off + 4);
    if (a.scale) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { v0[j] = act_apply(v0[j], act[0][c0 + j], act[1][c0 + j]); v1[j] = act_apply(v1[j], act[0][c0 + 4 + j], act[1][c0 + 4 + j]); }
    }
    if (b.x) {
        f32x4 t0 = *reinterpret_cast<const f32x4*>(b.x + off), t1 = *reinterpret_cast<const f32x4*>(b.x + off + 4);
        if (b.scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { t0[j] = act_apply(t0[j], act[2][c0 + j], act[3][c0 + j]); t1[j] = act_apply(t1[j], act[2][c0 + 4 + j], act[3][c0 + 4 + j]); }
        }
        v0 += t0; v1 += t1;
    }
    if (!in) { v0 = f32x4{0, 0, 0, 0}; v1 = v0; }                 // zero padding of the ACTIVATED input
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = (__bf16)v0[j]; r[4 + j] = (__bf16)v1[j]; }
    return r;
}

// D fragment -> memory + InPlaceABN partial sums.  Lane (n = lane & 15, g = lane >> 4): acc[nt][r] = voxel `vox_of(4 g + r)`, channel 16 nt + n.
template <int NT, typename VOXFN>
__device__ __forceinline__ void store_tile(const f32x4 (&acc)[NT], int Cout, float* __restrict__ out, float* __restrict__ stats, int64_t slot, int64_t nslots,
                                           float (*red)[2][NT * 16], int tid, VOXFN vox_of)
{
    const int lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    int64_t ov[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ov[r] = vox_of(4 * g + r);        // -1: no such voxel
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ch = nt * 16 + n;
        float s = 0.f, q = 0.f;
        if (ch < Cout) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (ov[r] >= 0) { const float v = acc[nt][r]; out[ov[r] * Cout + ch] = v; s += v; q = fmaf(v, v, q); }
        }
        if (stats) {
            s += __shfl_xor(s, 16); q += __shfl_xor(q, 16);
            s += __shfl_xor(s, 32); q += __shfl_xor(q, 32);
            if (g == 0) { red[wave][0][ch] = s; red[wave][1][ch] = q; }
        }
    }
    if (stats) {          // one slot per workgroup: abn_part_at(...) of common.h (abn_finalize_kernel's layout)
        __syncthreads();
        if (tid < 2 * NT * 16) {
            const int which = tid / (NT * 16), c = tid - which * (NT * 16);
            if (c < Cout) stats[abn_part_at(which, c, Cout, slot, nslots)] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ conv, stride S
// KSPLIT = 1: the four waves of a workgroup own four M-tiles.  KSPLIT = 4 (layers with few voxels: conv4 ... conv6 have 73 k / 9 k outputs,
// i.e. less than one M-tile per SIMD, and a wave's k-loop would be one long chain of exposed load latencies): the four waves share ONE
// M-tile, take every fourth k-step and meet in LDS; wave 0 adds the partial accumulators in a fixed order and stores.
// KZ x K x K taps, padding K / 2 (KZ / 2 along z): 3, 3 for the 3-D layers; 1, K for FeatureNet's 2-D layers (models.py:688-722) with the images as
// z (not strided).  Cin = 4 (the image layer, 3 real channels): a lane's eight k-values are two taps x four channels.  bias: the 1x1 toplayer.
template <int CIN, int NT, int S, int KSPLIT, int KZ, int K>
__global__ __launch_bounds__(256) void conv_bf16_kernel(ActSrc a, ActSrc b, int ld, int Di, int Hi, int Wi, const __bf16* __restrict__ wq, int Cout,
                                                       float* __restrict__ out, int Do, int Ho, int Wo, float* __restrict__ stats, const float* __restrict__ bias)
{
    constexpr int NTAP = KZ * K * K, PZ = KZ / 2, P = K / 2, SZ = KZ == 1 ? 1 : S;
    constexpr int KS = (NTAP * CIN + 31) / 32, LOG = ilog2c(CIN);
    static_assert((1 << LOG) == CIN && CIN >= 4, "Cin: a power of two >= 4 (groups of eight channels never straddle taps; Cin 4: two taps per group)");
    static_assert(KSPLIT == 1 || KSPLIT == 4, "one M-tile per wave, or one per workgroup");
    __shared__ float act[4][CIN];
    __shared__ float red[4][2][NT * 16];
    __shared__ __attribute__((aligned(16))) float part[KSPLIT == 4 ? 3 * NT * 64 * 4 : 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kg = lane >> 4;
    stage_act<CIN>(a, b, act, tid);
    const int64_t nvox = (int64_t)Do * Ho * Wo;
    const int64_t tile0 = (KSPLIT == 1 ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x) * 16;
    const int64_t vox = tile0 + m;
    const bool live = vox < nvox;
    const int64_t vc = live ? vox : nvox - 1;
    int x, y, z;
    mvs_unflatten3(vc, Wo, Ho, x, y, z);
    const int zb = z * SZ - PZ, yb = y * S - P, xb = x * S - P;
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0, 0, 0, 0};
    const bf16x8* __restrict__ wl = reinterpret_cast<const bf16x8*>(wq) + lane;
#pragma unroll 2
    for (int ks = (KSPLIT == 1 ? 0 : wave); ks < KS; ks += KSPLIT) {
        const int kb = ks * 32 + kg * 8;
        bf16x8 av;
        if constexpr (CIN == 4) {                                 // taps kb / 4 and kb / 4 + 1, four channels each
            f32x4 v[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const int tap = (kb >> 2) + hh;
                const int dz = tap / (K * K), rr = tap - dz * (K * K), dy = rr / K, dx = rr - dy * K;
                const int zi = zb + dz, yi = yb + dy, xi = xb + dx;
                const bool in = live && tap < NTAP && zi >= 0 && zi < Di && yi >= 0 && yi < Hi && xi >= 0 && xi < Wi;
                v[hh] = *reinterpret_cast<const f32x4*>(a.x + (in ? (((int64_t)zi * Hi + yi) * Wi + xi) * ld : 0));
                if (a.scale) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[hh][j] = act_apply(v[hh][j], act[0][j], act[1][j]);
                }
                if (!in) v[hh] = f32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { av[j] = (__bf16)v[0][j]; av[4 + j] = (__bf16)v[1][j]; }
        } else {
            const int tap = kb >> LOG, c0 = kb & (CIN - 1);
            const int dz = tap / (K * K), rr = tap - dz * (K * K), dy = rr / K, dx = rr - dy * K;
            const int zi = zb + dz, yi = yb + dy, xi = xb + dx;
            const bool in = live && tap < NTAP && zi >= 0 && zi < Di && yi >= 0 && yi < Hi && xi >= 0 && xi < Wi;
            const int64_t off = in ? (((int64_t)zi * Hi + yi) * Wi + xi) * ld + c0 : 0;
            av = load_a8<CIN>(a, b, act, off, c0, in);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wl[(ks * NT + nt) * 64], acc[nt], 0, 0, 0);
    }
    if constexpr (KSPLIT == 4) {
        if (wave > 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(part + (((wave - 1) * NT + nt) * 64 + lane) * 4) = acc[nt];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int w = 0; w < 3; ++w)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] += *reinterpret_cast<const f32x4*>(part + ((w * NT + nt) * 64 + lane) * 4);
        }
    }
    if (bias) {                                                   // D: lane (n = lane & 15, .): column 16 nt + n
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const float bv = (nt * 16 + (lane & 15)) < Cout ? bias[nt * 16 + (lane & 15)] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[nt][r] += bv;
        }
    }
    store_tile<NT>(acc, Cout, out, stats, blockIdx.x, gridDim.x, red, tid, [&](int r) {
        const int64_t o = tile0 + r;
        return (o < nvox && (KSPLIT == 1 || wave == 0)) ? o : (int64_t)-1;
    });
}

// ------------------------------------------------------------------------------------------------------------------ conv, LDS-tiled
// The big layers (conv1: 8 -> 16 stride 2 on the full-resolution volume, conv2: 16 -> 16 at half resolution, and the data gradients that are the
// same convolutions: conv2's own and conv11's) through LDS instead of 27 re-reads of every input voxel from L1 / L2 (each with its activation
// and rounding): the direct-load kernel above spends 85 / 65 us per launch on 37 / 150 MB of input, i.e. it is bound by the 27-fold load +
// activate + convert instruction stream, not by memory.  A workgroup walks a range of output tiles (TOZ x TOY x 32 voxels = 2 TOZ TOY M-tiles
// of 16 along x, four or two per wave); the input halo of a tile is activated and rounded ONCE on its way into LDS as [voxel][CIN] bf16 rows,
// and the A fragment of an MFMA is one 16-byte LDS read (eight channels of one tap of the lane's voxel).  The next tile's halo is requested
// (buffer loads, 32-bit offsets, zeros past the end) before the current tile's MFMAs and lands in registers under them.  The weights of all
// k-steps stay in registers (7 or 14 x 16 bytes per lane).  Same packed weights, same output, same InPlaceABN partial sums (slot = workgroup;
// the slots the direct-load grid would have had beyond that are written as zeros, so the caller's slot count does not depend on the kernel).
// KZ = 1: FeatureNet's full- and half-resolution 3 x 3 layers (models.py:688-722; the images are z, without halo or stride); Cout = 8 or 16 (one column block).
template <int CIN, int S, int TOZ, int TOY, int KZ = 3>
struct TiledCfg {
    static constexpr int TX = 32, HX = (TX - 1) * S + 3, HY = (TOY - 1) * S + 3, HZ = KZ == 3 ? (TOZ - 1) * S + 3 : TOZ, SZ = KZ == 3 ? S : 1;
    static constexpr int NVH = HX * HY * HZ, ROWB = CIN * 2, XQ = CIN / 4, NX = (NVH * XQ + 255) / 256;
    static constexpr int LDS_BYTES = ((NVH * ROWB + 63) & ~63) + 64;
    static constexpr int NTAP = KZ * 9, KS = (NTAP * CIN + 31) / 32;
};

template <int CIN, int S, int TOZ, int TOY, int KZ = 3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_bf16_tiled_kernel(
    ActSrc a, int ld, int Di, int Hi, int Wi, const __bf16* __restrict__ wq, int Cout, float* __restrict__ out, int Do, int Ho, int Wo,
    float* __restrict__ stats, int nslots)
{
    using C = TiledCfg<CIN, S, TOZ, TOY, KZ>;
    constexpr int TX = C::TX, HX = C::HX, HY = C::HY, NVH = C::NVH, ROWB = C::ROWB, XQ = C::XQ, NX = C::NX, KS = C::KS, NTAP = C::NTAP, SZ = C::SZ, COUT = 16;
    constexpr int MT_PER_WAVE = TOZ * TOY * 2 / 4;
    static_assert(CIN == 8 || CIN == 16, "the layers with >= 0.5 M output voxels");
    static_assert((TOZ * TOY * 2) % 4 == 0 && NX <= 32, "M-tiles divide among the four waves; one mask bit per prefetched quad");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* xt = lds;                                               // [NVH][CIN] bf16
    __shared__ float red[4][2][COUT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, kg = lane >> 4;
    const int nbx = (Wo + TX - 1) / TX, nby = (Ho + TOY - 1) / TOY, nbz = (Do + TOZ - 1) / TOZ;
    const int n_tiles = nbx * nby * nbz, n_ranges = gridDim.x;
    const int t_begin = (int)((int64_t)n_tiles * blockIdx.x / n_ranges), t_end = (int)((int64_t)n_tiles * (blockIdx.x + 1) / n_ranges);
    // weights of every k-step, and the LDS offset of this lane's tap in each
    bf16x8 wreg[KS];
    int toff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        wreg[ks] = reinterpret_cast<const bf16x8*>(wq)[ks * 64 + lane];
        const int kb = ks * 32 + kg * 8, tap = CIN == 16 ? (kb >> 4) : (kb >> 3), c0 = CIN == 16 ? (kb & 15) : 0;
        const int t = tap < NTAP ? tap : 0;                       // the padding k-values multiply zero weights: any address will do
        const int dz = KZ == 3 ? t / 9 : 0, dy = (t / 3) % 3, dx = t % 3;
        toff[ks] = ((dz * HY + dy) * HX + dx) * ROWB + c0 * 2;
    }
    // this thread's channel quad of every staged item, its activation
    const int xq = (tid & (XQ - 1)) * 4;
    f32x4 sc{1.f, 1.f, 1.f, 1.f}, sh{0.f, 0.f, 0.f, 0.f};
    const bool act_on = a.scale != nullptr;
    if (act_on) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[j] = a.scale[xq + j]; sh[j] = a.shift[xq + j]; }
    }
    typedef unsigned u32x4v __attribute__((__vector_size__(16)));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((int64_t)Di * Hi * Wi * ld * 4), 0x00020000);
    f32x4 px[NX];
    unsigned mx = 0;
    auto prefetch = [&](int tile) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int ix0 = bx * TX * S - 1, iy0 = by * TOY * S - 1, iz0 = KZ == 3 ? bz * TOZ * S - 1 : bz * TOZ;
        mx = 0;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int v = (tid + 256 * u) / XQ;
            const int ix = ix0 + v % HX, iy = iy0 + (v / HX) % HY, iz = iz0 + v / (HX * HY);
            const bool in = v < NVH && ix >= 0 && ix < Wi && iy >= 0 && iy < Hi && iz >= 0 && iz < Di;
            const unsigned off = in ? (unsigned)(((iz * Hi + iy) * Wi + ix) * ld + xq) * 4u : 0xffffffffu;
            px[u] = __builtin_bit_cast(f32x4, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            mx |= (unsigned)in << u;
        }
    };
    float s_sum = 0.f, q_sum = 0.f;                               // InPlaceABN partial sums of channel (lane & 15), rows 4 (lane >> 4) .. + 3 of every M-tile
    if (t_begin < t_end) prefetch(t_begin);
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int ox0 = bx * TX, oy0 = by * TOY, oz0 = bz * TOZ;
        __syncthreads();                                          // everybody is done with the previous tile
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int it = tid + 256 * u;
            f32x4 v = px[u];
            if (act_on && ((mx >> u) & 1)) {                     // the zero padding is padding of the ACTIVATED input
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], sc[j], sh[j]);
            }
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
            if (it < NVH * XQ) *reinterpret_cast<bf16x4*>(xt + it * 8) = o;
        }
        __syncthreads();
        if (tile + 1 < t_end) prefetch(tile + 1);
#pragma unroll
        for (int q = 0; q < MT_PER_WAVE; ++q) {
            const int mt = wave * MT_PER_WAVE + q, xh = mt & 1, row = mt >> 1, oy_l = row % TOY, oz_l = row / TOY;
            const char* base = xt + ((oz_l * SZ * HY + oy_l * S) * HX + (xh * 16 + m) * S) * ROWB;
            f32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(base + toff[ks]), wreg[ks], acc, 0, 0, 0);
            // D: lane (n = lane & 15, g = lane >> 4): acc[r] = voxel 4 g + r of the M-tile, channel n
            const int oz = oz0 + oz_l, oy = oy0 + oy_l;
            if (oz < Do && oy < Ho && m < Cout) {
                float* orow = out + (((int64_t)oz * Ho + oy) * Wo) * Cout + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ox = ox0 + xh * 16 + 4 * kg + r;
                    if (ox < Wo) { const float v = acc[r]; orow[(int64_t)ox * Cout] = v; s_sum += v; q_sum = fmaf(v, v, q_sum); }
                }
            }
        }
    }
    if (stats) {
        s_sum += __shfl_xor(s_sum, 16); q_sum += __shfl_xor(q_sum, 16);
        s_sum += __shfl_xor(s_sum, 32); q_sum += __shfl_xor(q_sum, 32);
        if (kg == 0) { red[wave][0][m] = s_sum; red[wave][1][m] = q_sum; }
        __syncthreads();
        if (tid < 2 * Cout) {
            const int which = tid / Cout, c = tid - which * Cout;
            stats[abn_part_at(which, c, Cout, blockIdx.x, nslots)] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
        }
        // the slots this grid does not own
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < (int64_t)(nslots - (int)gridDim.x) * 2 * Cout; i += (int64_t)gridDim.x * 256) {
            const int64_t slot = gridDim.x + i / (2 * Cout);
            const int r = (int)(i % (2 * Cout));
            stats[abn_part_at(r / Cout, r % Cout, Cout, slot, nslots)] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ transposed conv, stride 2
template <int CIN, int NT>          // NT = 2 * Cout / 16 column blocks (both x parities)
__global__ __launch_bounds__(256) void convT3d_k3s2_bf16_kernel(ActSrc a, ActSrc b, int Di, int Hi, int Wi, const __bf16* __restrict__ wq, int Cout,
                                                               float* __restrict__ out, float* __restrict__ stats)
{
    constexpr int LOG = ilog2c(CIN), KSC = (8 * CIN) / 32;        // k-steps reserved per (pz, py) class in wq (the class with four (z, y) tap pairs)
    static_assert(CIN >= 16, "two x offsets x Cin must fill whole k-steps");
    __shared__ float act[4][CIN];
    __shared__ float red[4][2][NT * 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, kg = lane >> 4, n = lane & 15, g = lane >> 4;
    stage_act<CIN>(a, b, act, tid);
    const int64_t nvox = (int64_t)Di * Hi * Wi;                   // input lattice positions
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
    const int64_t vox = tile0 + m;
    const bool live = vox < nvox;
    const int64_t vc = live ? vox : nvox - 1;
    int x, y, z;
    mvs_unflatten3(vc, Wi, Hi, x, y, z);
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    // rows of the D fragment this lane stores: positions tile0 + 4 g + r
    int64_t obase[4];                                             // float offset of output voxel (2 iz, 2 iy, 2 ix), -1: no such position
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = tile0 + 4 * g + r;
        int ix, iy, iz;
        mvs_unflatten3(i, Wi, Hi, ix, iy, iz);
        obase[r] = i < nvox ? (((int64_t)(2 * iz) * Ho + 2 * iy) * Wo + 2 * ix) * Cout : (int64_t)-1;
    }
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { ssum[nt] = 0.f; ssq[nt] = 0.f; }
    const bf16x8* __restrict__ wl = reinterpret_cast<const bf16x8*>(wq) + lane;
#pragma unroll 1
    for (int cls = 0; cls < 4; ++cls) {
        const int pz = cls >> 1, py = cls & 1;
        const int ks_n = ((1 + pz) * (1 + py) * 2 * CIN) / 32;
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0, 0, 0, 0};
#pragma unroll 2
        for (int ks = 0; ks < ks_n; ++ks) {
            const int kb = ks * 32 + kg * 8;
            const int t = kb >> LOG, c0 = kb & (CIN - 1);
            // t = ((sz) * (1 + py) + sy) * 2 + sx with the sub-tap bits of the parity-1 dimensions: bit 0 -> kernel index 0, input i + 1; bit 1 ->
            // kernel index 2, input i (x: both always - the column's parity picks its weight)
            int bits = t;
            const int sx = bits & 1; bits >>= 1;
            const int sy = py ? (bits & 1) : 1; bits >>= py;
            const int sz = pz ? (bits & 1) : 1;
            const int zi = z + (sz ? 0 : 1), yi = y + (sy ? 0 : 1), xi = x + (sx ? 0 : 1);
            const bool in = live && zi < Di && yi < Hi && xi < Wi;
            const int64_t off = in ? (((int64_t)zi * Hi + yi) * Wi + xi) * CIN + c0 : 0;
            const bf16x8 av = load_a8<CIN>(a, b, act, off, c0, in);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, wl[((cls * KSC + ks) * NT + nt) * 64], acc[nt], 0, 0, 0);
        }
        // D: lane (n, g): acc[nt][r] = position tile0 + 4 g + r, column 16 nt + n = px * Cout + channel -> 2 Cout contiguous floats per position
        const int64_t coff = ((int64_t)pz * Ho + py) * Wo * Cout;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (obase[r] >= 0) { const float v = acc[nt][r]; out[obase[r] + coff + nt * 16 + n] = v; ssum[nt] += v; ssq[nt] = fmaf(v, v, ssq[nt]); }
    }
    if (stats) {
        // channel of column 16 nt + n: (16 nt + n) % Cout; the two x parities of a channel: NT = 1 (Cout 8) lanes n, n ^ 8; NT = 2 blocks 0, 1; NT = 4 blocks (0, 2), (1, 3)
        constexpr int NCH = NT == 1 ? 1 : NT / 2;                 // column blocks per parity
        float cs[NCH], cq[NCH];
        if constexpr (NT == 1) { cs[0] = ssum[0] + __shfl_xor(ssum[0], 8); cq[0] = ssq[0] + __shfl_xor(ssq[0], 8); }
        else {
#pragma unroll
            for (int h = 0; h < NCH; ++h) { cs[h] = ssum[h] + ssum[h + NCH]; cq[h] = ssq[h] + ssq[h + NCH]; }
        }
#pragma unroll
        for (int h = 0; h < NCH; ++h) {
            cs[h] += __shfl_xor(cs[h], 16); cq[h] += __shfl_xor(cq[h], 16);
            cs[h] += __shfl_xor(cs[h], 32); cq[h] += __shfl_xor(cq[h], 32);
            if (g == 0 && (NT > 1 || n < 8)) { red[wave][0][h * 16 + n] = cs[h]; red[wave][1][h * 16 + n] = cq[h]; }
        }
        __syncthreads();
        if (tid < 2 * Cout) {
            const int which = tid / Cout, c = tid - which * Cout;
            stats[abn_part_at(which, c, Cout, blockIdx.x, gridDim.x)] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ transposed conv 16 -> 8, LDS-tiled
// conv11 (16 channels at half resolution -> 8 at full resolution, the skip sum conv2 + conv9 as a two-source input) and conv1's data gradient:
// 0.59 M input positions, 150 MB of output.  Same M-tile / column / k enumeration and the same packed weights as convT3d_k3s2_bf16_kernel<16, 1>;
// the input tile (TIZ x TIY x 32 positions + one more along each axis) is activated, summed and rounded once into LDS, an A fragment is one
// 16-byte LDS read, the nine k-steps' weights stay in registers, the next tile's loads run under the MFMAs (94 -> see profiles/ us per launch).
template <bool TWO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void convT3d_16to8_bf16_tiled_kernel(
    ActSrc a, ActSrc b, int Di, int Hi, int Wi, const __bf16* __restrict__ wq, float* __restrict__ out, float* __restrict__ stats, int nslots)
{
    constexpr int CIN = 16, COUT = 8, TIZ = 2, TIY = 4, TX = 32, HX = TX + 1, HY = TIY + 1, HZ = TIZ + 1, NVH = HX * HY * HZ, ROWB = CIN * 2, XQ = 4;
    constexpr int NX = (NVH * XQ + 255) / 256, MT_PER_WAVE = TIZ * TIY * 2 / 4, KSC = (8 * CIN) / 32, NKS = 9;
    __shared__ __attribute__((aligned(1024))) char xt[(NVH * ROWB + 63) & ~63];
    __shared__ float red[4][2][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, kg = lane >> 4;
    const int Ho = 2 * Hi, Wo = 2 * Wi;
    const int nbx = (Wi + TX - 1) / TX, nby = (Hi + TIY - 1) / TIY, nbz = (Di + TIZ - 1) / TIZ;
    const int n_tiles = nbx * nby * nbz, n_ranges = gridDim.x;
    const int t_begin = (int)((int64_t)n_tiles * blockIdx.x / n_ranges), t_end = (int)((int64_t)n_tiles * (blockIdx.x + 1) / n_ranges);
    // the nine k-steps (classes (pz, py) = (0,0): 1, (0,1): 2, (1,0): 2, (1,1): 4): weights and this lane's tap offset in the LDS tile
    bf16x8 wreg[NKS];
    int toff[NKS];
    {
        int i = 0;
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            const int pz = cls >> 1, py = cls & 1, ks_n = ((1 + pz) * (1 + py) * 2 * CIN) / 32;
#pragma unroll
            for (int ks = 0; ks < ks_n; ++ks, ++i) {
                wreg[i] = reinterpret_cast<const bf16x8*>(wq)[(cls * KSC + ks) * 64 + lane];
                const int kb = ks * 32 + kg * 8, t = kb >> 4, c0 = kb & 15;
                int bits = t;
                const int sx = bits & 1; bits >>= 1;
                const int sy = py ? (bits & 1) : 1; bits >>= py;
                const int sz = pz ? (bits & 1) : 1;
                toff[i] = (((sz ? 0 : 1) * HY + (sy ? 0 : 1)) * HX + (sx ? 0 : 1)) * ROWB + c0 * 2;
            }
        }
    }
    const int xq = (tid & (XQ - 1)) * 4;
    f32x4 sc1{1.f, 1.f, 1.f, 1.f}, sh1{0.f, 0.f, 0.f, 0.f}, sc2 = sc1, sh2 = sh1;
    const bool on1 = a.scale != nullptr, on2 = TWO && b.scale != nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (on1) { sc1[j] = a.scale[xq + j]; sh1[j] = a.shift[xq + j]; }
        if (on2) { sc2[j] = b.scale[xq + j]; sh2[j] = b.shift[xq + j]; }
    }
    typedef unsigned u32x4v __attribute__((__vector_size__(16)));
    const int bytes = (int)((int64_t)Di * Hi * Wi * CIN * 4);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, bytes, 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(TWO ? b.x : a.x), 0, bytes, 0x00020000);
    f32x4 px[NX], py2[TWO ? NX : 1];
    unsigned mx = 0;
    auto prefetch = [&](int tile) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int ix0 = bx * TX, iy0 = by * TIY, iz0 = bz * TIZ;
        mx = 0;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int v = (tid + 256 * u) / XQ;
            const int ix = ix0 + v % HX, iy = iy0 + (v / HX) % HY, iz = iz0 + v / (HX * HY);
            const bool in = v < NVH && ix < Wi && iy < Hi && iz < Di;
            const unsigned off = in ? (unsigned)(((iz * Hi + iy) * Wi + ix) * CIN + xq) * 4u : 0xffffffffu;
            px[u] = __builtin_bit_cast(f32x4, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0));
            if (TWO) py2[TWO ? u : 0] = __builtin_bit_cast(f32x4, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs2, off, 0, 0));
            mx |= (unsigned)in << u;
        }
    };
    float s_sum = 0.f, q_sum = 0.f;                               // column n = px * 8 + channel
    if (t_begin < t_end) prefetch(t_begin);
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int ix0 = bx * TX, iy0 = by * TIY, iz0 = bz * TIZ;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const int it = tid + 256 * u;
            f32x4 v = px[u];
            if ((mx >> u) & 1) {
                if (on1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], sc1[j], sh1[j]);
                }
                if (TWO) {
                    f32x4 t = py2[TWO ? u : 0];
                    if (on2) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) t[j] = act_apply(t[j], sc2[j], sh2[j]);
                    }
                    v += t;
                }
            }
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
            if (it < NVH * XQ) *reinterpret_cast<bf16x4*>(xt + it * 8) = o;
        }
        __syncthreads();
        if (tile + 1 < t_end) prefetch(tile + 1);
#pragma unroll
        for (int q = 0; q < MT_PER_WAVE; ++q) {
            const int mt = wave * MT_PER_WAVE + q, xh = mt & 1, row = mt >> 1, iy_l = row % TIY, iz_l = row / TIY;
            const char* base = xt + ((iz_l * HY + iy_l) * HX + xh * 16 + m) * ROWB;
            const int iz = iz0 + iz_l, iy = iy0 + iy_l;
            const bool row_ok = iz < Di && iy < Hi;
            float* orow = out + ((((int64_t)(2 * iz) * Ho + 2 * iy) * Wo) * COUT) + m;
            int i = 0;
#pragma unroll
            for (int cls = 0; cls < 4; ++cls) {
                const int pz = cls >> 1, py = cls & 1, ks_n = ((1 + pz) * (1 + py) * 2 * CIN) / 32;
                f32x4 acc{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < ks_n; ++ks, ++i)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(base + toff[i]), wreg[i], acc, 0, 0, 0);
                // D: lane (n = lane & 15, g = lane >> 4): acc[r] = position 4 g + r of the M-tile, column n = (x parity, channel): 16 contiguous floats
                if (row_ok) {
                    float* oc = orow + ((int64_t)pz * Ho + py) * Wo * COUT;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ix = ix0 + xh * 16 + 4 * kg + r;
                        if (ix < Wi) { const float v = acc[r]; oc[(int64_t)ix * 2 * COUT] = v; s_sum += v; q_sum = fmaf(v, v, q_sum); }
                    }
                }
            }
        }
    }
    if (stats) {
        s_sum += __shfl_xor(s_sum, 8); q_sum += __shfl_xor(q_sum, 8);          // the two x parities of a channel
        s_sum += __shfl_xor(s_sum, 16); q_sum += __shfl_xor(q_sum, 16);
        s_sum += __shfl_xor(s_sum, 32); q_sum += __shfl_xor(q_sum, 32);
        if (kg == 0 && m < 8) { red[wave][0][m] = s_sum; red[wave][1][m] = q_sum; }
        __syncthreads();
        if (tid < 2 * COUT) {
            const int which = tid / COUT, c = tid - which * COUT;
            stats[abn_part_at(which, c, COUT, blockIdx.x, nslots)] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
        }
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < (int64_t)(nslots - (int)gridDim.x) * 2 * COUT; i += (int64_t)gridDim.x * 256) {
            const int64_t slot = gridDim.x + i / (2 * COUT);
            const int r = (int)(i % (2 * COUT));
            stats[abn_part_at(r / COUT, r % COUT, COUT, slot, nslots)] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ weight fragments
// wp[27][Cin][Cout] fp32 (mvsnerf_conv3d_pack_weights: the generic layout of the layer or of its data gradient) -> bf16 B fragments.
// conv:        wq[ks][nt][lane][8],  k = 32 ks + 8 (lane >> 4) + j = tap * Cin + ci, column 16 nt + (lane & 15) = output channel
// transposed:  wq[class pz, py][ks][nt][lane][8] (Cin / 4 k-steps reserved per class), k = t * Cin + ci with t the class's sub-tap number,
//              column = px * Cout + output channel (kernel above)
__global__ __launch_bounds__(256) void conv3d_bf16_pack_kernel(const float* __restrict__ wp, int Cin, int Cout, int transposed, int64_t total, __bf16* __restrict__ wq)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int tap, ci, co;
    wq[i] = (__bf16)(mvs_conv3d_bf16_coords(i, Cin, Cout, transposed, tap, ci, co) ? wp[((int64_t)tap * Cin + ci) * Cout + co] : 0.0f);
}

bool shape_ok(int Cin, int Cout) { return mvs_conv3d_bf16_shape_ok(Cin, Cout); }
int n_col_blocks(int Cout) { return (Cout + 15) / 16; }

}  // namespace

extern "C" size_t mvsnerf_conv3d_bf16_packed_elems(int Cin, int Cout, int transposed)
{
    return mvs_conv3d_bf16_elems(Cin, Cout, transposed);
}

extern "C" int mvsnerf_conv3d_bf16_pack(const float* wpacked, int Cin, int Cout, int transposed, void* wq, void* stream)
{
    if (!wpacked || !wq) return MVSNERF_EINVAL;
    if (!shape_ok(Cin, Cout)) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(wq)) return MVSNERF_EALIGN;
    const size_t total = mvs_conv3d_bf16_elems(Cin, Cout, transposed);
    if (total == 0) return MVSNERF_EUNSUPPORTED;
    conv3d_bf16_pack_kernel<<<mvs_cdiv((int64_t)total, 256), 256, 0, (hipStream_t)stream>>>(wpacked, Cin, Cout, transposed ? 1 : 0, (int64_t)total, reinterpret_cast<__bf16*>(wq));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// InPlaceABN statistics slots of the launches below (input dims) = workgroups: 64 output voxels each, or 16 when the layer is small enough for
// the k-split form (fewer than 8192 M-tiles of 16 voxels); x 8 parity classes when transposed
static bool ksplit_of(int64_t nvox_out) { return (nvox_out + 15) / 16 < 8192; }

extern "C" int mvsnerf_conv3d_bf16_tiles(int D, int H, int W, int stride)
{
    if (stride != 1 && stride != 2) return 0;
    const int64_t nvox = (int64_t)((D - 1) / stride + 1) * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
    return (int)(ksplit_of(nvox) ? (nvox + 15) / 16 : (nvox + 63) / 64);
}

extern "C" int mvsnerf_conv_transpose3d_bf16_tiles(int D, int H, int W) { return (int)(((int64_t)D * H * W + 63) / 64); }

static bool act_ok16(const float* x, const float* sc, const float* sh) { return x && ((sc == nullptr) == (sh == nullptr)) && mvs_aligned16(x); }

extern "C" int mvsnerf_conv3d_bf16_fwd(const float* x1, const float* scale1, const float* shift1,
                                       const float* x2, const float* scale2, const float* shift2,
                                       int Cin, int cin_ld, int D, int H, int W, const void* wq, int Cout, int stride,
                                       float* out, float* stats_part, void* stream)
{
    if (!act_ok16(x1, scale1, shift1) || !wq || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (x2 && !act_ok16(x2, scale2, shift2)) return MVSNERF_EINVAL;
    if ((stride != 1 && stride != 2) || !shape_ok(Cin, Cout)) return MVSNERF_EUNSUPPORTED;
    if ((cin_ld & 3) || cin_ld < Cin || !mvs_aligned16(wq)) return MVSNERF_EALIGN;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;   // k3 p1
    const unsigned grid = (unsigned)mvsnerf_conv3d_bf16_tiles(D, H, W, stride);
    hipStream_t st = (hipStream_t)stream;
    const __bf16* w = reinterpret_cast<const __bf16*>(wq);
    const bool split = ksplit_of((int64_t)Do * Ho * Wo);
    // the LDS-tiled kernel: single-source 8 -> 16 stride 2 / 16 -> 16 stride 1 layers with at least 256 K output voxels, tensors below 2 GB
    if (!x2 && Cout == 16 && !split && (int64_t)Do * Ho * Wo >= (1 << 18) && (int64_t)D * H * W * cin_ld * 4 < (1ll << 31) && (int)grid >= 512 &&
        ((Cin == 16 && stride == 1) || (Cin == 8 && stride == 2))) {
        if (Cin == 16) conv_bf16_tiled_kernel<16, 1, 2, 4><<<512, 256, TiledCfg<16, 1, 2, 4>::LDS_BYTES, st>>>(a, cin_ld, D, H, W, w, Cout, out, Do, Ho, Wo, stats_part, (int)grid);
        else conv_bf16_tiled_kernel<8, 2, 2, 2><<<512, 256, TiledCfg<8, 2, 2, 2>::LDS_BYTES, st>>>(a, cin_ld, D, H, W, w, Cout, out, Do, Ho, Wo, stats_part, (int)grid);
        MVS_LAUNCH_CHECK();
        return MVSNERF_OK;
    }
#define MVS_C16(CIN, NT, S) do { if (split) conv_bf16_kernel<CIN, NT, S, 4, 3, 3><<<grid, 256, 0, st>>>(a, b, cin_ld, D, H, W, w, Cout, out, Do, Ho, Wo, stats_part, nullptr); \
                                 else conv_bf16_kernel<CIN, NT, S, 1, 3, 3><<<grid, 256, 0, st>>>(a, b, cin_ld, D, H, W, w, Cout, out, Do, Ho, Wo, stats_part, nullptr); } while (0)
    switch ((Cin * 100 + n_col_blocks(Cout)) * 10 + stride) {
        case (8 * 100 + 1) * 10 + 1: MVS_C16(8, 1, 1); break;    case (8 * 100 + 1) * 10 + 2: MVS_C16(8, 1, 2); break;
        case (16 * 100 + 1) * 10 + 1: MVS_C16(16, 1, 1); break;  case (16 * 100 + 1) * 10 + 2: MVS_C16(16, 1, 2); break;
        case (16 * 100 + 2) * 10 + 1: MVS_C16(16, 2, 1); break;  case (16 * 100 + 2) * 10 + 2: MVS_C16(16, 2, 2); break;
        case (32 * 100 + 1) * 10 + 2: MVS_C16(32, 1, 2); break;
        case (32 * 100 + 2) * 10 + 1: MVS_C16(32, 2, 1); break;  case (32 * 100 + 2) * 10 + 2: MVS_C16(32, 2, 2); break;
        case (32 * 100 + 4) * 10 + 1: MVS_C16(32, 4, 1); break;  case (32 * 100 + 4) * 10 + 2: MVS_C16(32, 4, 2); break;
        case (64 * 100 + 2) * 10 + 2: MVS_C16(64, 2, 2); break;
        case (64 * 100 + 4) * 10 + 1: MVS_C16(64, 4, 1); break;  case (64 * 100 + 4) * 10 + 2: MVS_C16(64, 4, 2); break;
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_C16
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_conv_transpose3d_bf16_fwd(const float* x1, const float* scale1, const float* shift1,
                                                 const float* x2, const float* scale2, const float* shift2,
                                                 int Cin, int D, int H, int W, const void* wq, int Cout, float* out, float* stats_part, void* stream)
{
    if (!act_ok16(x1, scale1, shift1) || !wq || !out || D < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if (x2 && !act_ok16(x2, scale2, shift2)) return MVSNERF_EINVAL;
    if (!shape_ok(Cin, Cout) || Cin < 16) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(wq)) return MVSNERF_EALIGN;
    const ActSrc a{x1, scale1, shift1}, b{x2, scale2, shift2};
    const unsigned grid = (unsigned)(((int64_t)D * H * W + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    const __bf16* w = reinterpret_cast<const __bf16*>(wq);
    if (Cin == 16 && Cout == 8 && (int)grid >= 512 && (int64_t)D * H * W >= (1 << 18) && (int64_t)D * H * W * 16 * 4 < (1ll << 31)) {   // the LDS-tiled kernel
        if (x2) convT3d_16to8_bf16_tiled_kernel<true><<<512, 256, 0, st>>>(a, b, D, H, W, w, out, stats_part, (int)grid);
        else convT3d_16to8_bf16_tiled_kernel<false><<<512, 256, 0, st>>>(a, b, D, H, W, w, out, stats_part, (int)grid);
        MVS_LAUNCH_CHECK();
        return MVSNERF_OK;
    }
#define MVS_T16(CIN, NT) convT3d_k3s2_bf16_kernel<CIN, NT><<<grid, 256, 0, st>>>(a, b, D, H, W, w, Cout, out, stats_part)
    switch (Cin * 100 + Cout) {
        case 16 * 100 + 8: MVS_T16(16, 1); break;           // conv11, data gradient of conv1
        case 32 * 100 + 16: MVS_T16(32, 2); break;          // conv9, data gradient of conv3
        case 64 * 100 + 32: MVS_T16(64, 4); break;          // conv7, data gradient of conv5
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_T16
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ------------------------------------------------------------------------------------------------------------------ FeatureNet (2-D)
// The same kernel with 1 x k x k taps over [N][H][W][C] images (z = image, never strided): models.py:688-722 under use_amp.  ksize 1 | 3 | 5,
// stride 1 | 2 (padding ksize / 2), one lazily-activated source, optional bias (the 1x1 toplayer).  Weights: mvsnerf_pack_weights_multi kind 3
// with ntaps = ksize^2 from the Conv2d weight (or its data-gradient view).  Shapes: the eight layers + toplayer and their stride-1 data gradients.
extern "C" size_t mvsnerf_conv2d_bf16_packed_elems(int Cin, int Cout, int ksize)
{
    if (ksize != 1 && ksize != 3 && ksize != 5) return 0;
    return mvs_conv3d_bf16_elems(Cin, Cout, 0, ksize * ksize);
}

static int64_t conv2d_out_pixels(int N, int H, int W, int ksize, int stride)
{
    const int P = ksize / 2;
    return (int64_t)N * ((H + 2 * P - ksize) / stride + 1) * ((W + 2 * P - ksize) / stride + 1);
}

extern "C" int mvsnerf_conv2d_bf16_tiles(int N, int H, int W, int ksize, int stride)
{
    if ((stride != 1 && stride != 2) || (ksize != 1 && ksize != 3 && ksize != 5)) return 0;
    const int64_t npix = conv2d_out_pixels(N, H, W, ksize, stride);
    return (int)(ksplit_of(npix) ? (npix + 15) / 16 : (npix + 63) / 64);
}

extern "C" int mvsnerf_conv2d_bf16_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int N, int H, int W,
                                       const void* wq, const float* bias, int Cout, int ksize, int stride, float* out, float* stats_part, void* stream)
{
    if (!act_ok16(x, scale, shift) || !wq || !out || N < 1 || H < 1 || W < 1) return MVSNERF_EINVAL;
    if ((stride != 1 && stride != 2) || mvsnerf_conv2d_bf16_packed_elems(Cin, Cout, ksize) == 0) return MVSNERF_EUNSUPPORTED;
    if ((cin_ld & 3) || cin_ld < Cin || !mvs_aligned16(wq)) return MVSNERF_EALIGN;
    const ActSrc a{x, scale, shift}, b{nullptr, nullptr, nullptr};
    const int P = ksize / 2, Ho = (H + 2 * P - ksize) / stride + 1, Wo = (W + 2 * P - ksize) / stride + 1;
    const unsigned grid = (unsigned)mvsnerf_conv2d_bf16_tiles(N, H, W, ksize, stride);
    const bool split = ksplit_of((int64_t)N * Ho * Wo);
    hipStream_t st = (hipStream_t)stream;
    const __bf16* w = reinterpret_cast<const __bf16*>(wq);
    // the LDS-tiled kernel: 3 x 3 stride-1 layers with 8 or 16 input channels, one column block, no bias, at least 128 K pixels, tensors below 2 GB
    if (ksize == 3 && stride == 1 && !bias && (Cin == 8 || Cin == 16) && Cout <= 16 && !split && (int64_t)N * Ho * Wo >= (1 << 17) && (int)grid >= 1024 &&
        (int64_t)N * H * W * cin_ld * 4 < (1ll << 31)) {
        if (Cin == 8) conv_bf16_tiled_kernel<8, 1, 1, 16, 1><<<1024, 256, TiledCfg<8, 1, 1, 16, 1>::LDS_BYTES, st>>>(a, cin_ld, N, H, W, w, Cout, out, N, Ho, Wo, stats_part, (int)grid);
        else conv_bf16_tiled_kernel<16, 1, 1, 8, 1><<<768, 256, TiledCfg<16, 1, 1, 8, 1>::LDS_BYTES, st>>>(a, cin_ld, N, H, W, w, Cout, out, N, Ho, Wo, stats_part, (int)grid);
        MVS_LAUNCH_CHECK();
        return MVSNERF_OK;
    }
#define MVS_C2(CIN, NT, S, K) do { if (split) conv_bf16_kernel<CIN, NT, S, 4, 1, K><<<grid, 256, 0, st>>>(a, b, cin_ld, N, H, W, w, Cout, out, N, Ho, Wo, stats_part, bias); \
                                   else conv_bf16_kernel<CIN, NT, S, 1, 1, K><<<grid, 256, 0, st>>>(a, b, cin_ld, N, H, W, w, Cout, out, N, Ho, Wo, stats_part, bias); } while (0)
    switch (((Cin * 100 + n_col_blocks(Cout)) * 10 + ksize) * 10 + stride) {
        case ((4 * 100 + 1) * 10 + 3) * 10 + 1: MVS_C2(4, 1, 1, 3); break;        // conv0.0 (3 -> 8)
        case ((8 * 100 + 1) * 10 + 3) * 10 + 1: MVS_C2(8, 1, 1, 3); break;        // conv0.1, and data gradients 8 -> 8 / 8 -> 4
        case ((8 * 100 + 1) * 10 + 5) * 10 + 2: MVS_C2(8, 1, 2, 5); break;        // conv1.0 (8 -> 16)
        case ((16 * 100 + 1) * 10 + 3) * 10 + 1: MVS_C2(16, 1, 1, 3); break;      // conv1.1, conv1.2 (+ data gradients)
        case ((16 * 100 + 2) * 10 + 5) * 10 + 2: MVS_C2(16, 2, 2, 5); break;      // conv2.0 (16 -> 32)
        case ((32 * 100 + 2) * 10 + 3) * 10 + 1: MVS_C2(32, 2, 1, 3); break;      // conv2.1, conv2.2 (+ data gradients)
        case ((32 * 100 + 2) * 10 + 1) * 10 + 1: MVS_C2(32, 2, 1, 1); break;      // toplayer (+ data gradient)
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_C2
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
