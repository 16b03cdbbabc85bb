// conv1 (8 -> 16, stride 2, full resolution) and conv2 (16 -> 16, stride 1, half resolution) of CostRegNet (models.py:725-769) with fp32-GRADE results
// from the fp16 matrix cores, for the NO-GRAD scene encode (encoder.ENCODER_PRECISION = "auto"), where they ran on round-2 fp32 kernels until round 5
// (conv3d_k3_mfma16_kernel<8, 2> 107 us, conv3d_k3s1_tiled_kernel<16, 16, 16> 120 us at config 2; here 52 / 43 us, profiles/r05_conv12_f16x3_tiled.txt):
// csrc/conv3d_bf16.hip's LDS-tiled kernel (halo activated ONCE on its way into LDS, next tile's halo prefetched under the MFMAs) with the two-piece operand
// split of csrc/conv_f16x3.hip:
//     x = x0 + x1,  x0 = fp16(x 2^4), x1 = fp16(x 2^4 - x0);   w likewise with 2^8;   x w 2^12 ~= x1 w0 + x0 w1 + x0 w0   (dropped: x1 w1 <= 2^-22 |x w|)
// three v_mfma_f32_16x16x32_f16 per product, fp32 accumulation, the power-of-two scales taken out of the accumulator exactly.  The scales keep the SECOND
// pieces of ordinary activations (|x| ~ 1e-3 .. 50 after InPlaceABN) and weights (~1e-2 .. 1) out of fp16's subnormals.  Range: |x| < 4094, |w| < 255; a value
// beyond becomes inf, every output it touches NaN, and the NaN partial sums set *guard: mvsnerf_conv3d_f16x3_guarded_fwd (encoder.hip) enqueues the fp32 kernel
// of the layer behind this one, predicated on that word (include/mvsnerf_hip.h, "guarded 16-bit sequences").
// Measured on the way (scratch/r5_prep/NOTES.md, profiles/r04_r5prep_*): the compiler's own schedule (4 reads, wait, 3 dependent MFMAs per k-step and M-tile) 90 us at
// conv2; fragments of k-step ks + 1 requested before the MFMAs of ks, the three piece products as three passes over the M-tiles: 78; the tile-independent half of the
// prefetch addressing hoisted out of the tile loop: 73; 16-wide tiles for rows that are not multiples of 32 voxels (104: 93 % of the M-tile rows used instead of 81 %): 43.
#include "common.h"
#include "act.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr float X_SCALE = 16.0f, W_SCALE = 256.0f, OUT_SCALE = 1.0f / (16.0f * 256.0f);

template <int CIN, int S, int TOZ, int TOY, int KZ = 3, int TXP = 32>
struct TiledCfgH {
    static constexpr int TX = TXP, HX = (TX - 1) * S + 3, HY = (TOY - 1) * S + 3, HZ = KZ == 3 ? (TOZ - 1) * S + 3 : TOZ, SZ = KZ == 3 ? S : 1;
    static constexpr int NVH = HX * HY * HZ, ROWB = CIN * 2, XQ = CIN / 4, NX = (NVH * XQ + 255) / 256;
    static constexpr int PLANE = (NVH * ROWB + 63) & ~63;              // bytes of one piece plane [voxel][CIN] fp16
    static constexpr int NTAP = KZ * 9, KS = (NTAP * CIN + 31) / 32;
    static constexpr bool W_IN_REGS = CIN == 8;                      // first weight pieces in registers (7 x 16 B per lane); with 16 channels (14 x 16 B) both pieces
                                                                     // live in LDS: in registers the kernel needs more than the 256 VGPRs of two waves per SIMD and spills
    static constexpr int LDS_BYTES = 2 * PLANE + (W_IN_REGS ? 1 : 2) * KS * 1024 + 64;
};

template <int CIN, int S, int TOZ, int TOY, int KZ = 3, int TXP = 32>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_f16x3_tiled_kernel(
    ActSrc a, int ld, int Di, int Hi, int Wi, const _Float16* __restrict__ wq, int Cout, float* __restrict__ out, int Do, int Ho, int Wo,
    float* __restrict__ stats, int nslots, int* __restrict__ guard)
{
    using C = TiledCfgH<CIN, S, TOZ, TOY, KZ, TXP>;
    constexpr int TX = C::TX, HX = C::HX, HY = C::HY, NVH = C::NVH, ROWB = C::ROWB, XQ = C::XQ, NX = C::NX, KS = C::KS, NTAP = C::NTAP, SZ = C::SZ, PLANE = C::PLANE, COUT = 16;
    constexpr int XT = TX / 16;                                   // M-tiles (16 voxels along x) per tile row: 2, or 1 for the 16-wide tiles (rows of 104 voxels: 93 % instead of 81 % used)
    constexpr int MT_PER_WAVE = TOZ * TOY * XT / 4;
    static_assert(CIN == 8 || CIN == 16, "the layers with >= 0.5 M output voxels");
    static_assert((TX == 16 || TX == 32) && (TOZ * TOY * XT) % 4 == 0 && NX <= 32, "M-tiles divide among the four waves; one mask bit per prefetched quad");
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* xt = lds;                                               // hi plane [NVH][CIN] fp16, lo plane at + PLANE
    __shared__ float red[4][2][COUT];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), m = lane & 15, kg = lane >> 4;
    const int nbx = (Wo + TX - 1) / TX, nby = (Ho + TOY - 1) / TOY, nbz = (Do + TOZ - 1) / TOZ;
    const int n_tiles = nbx * nby * nbz, n_ranges = gridDim.x;
    const int t_begin = (int)((int64_t)n_tiles * blockIdx.x / n_ranges), t_end = (int)((int64_t)n_tiles * (blockIdx.x + 1) / n_ranges);
    // the weights of every k-step (wq[piece][ks][lane][8]): first pieces in registers, second pieces in LDS (both in registers: 256 VGPRs and spills
    // at two waves per SIMD); and the LDS offset of this lane's tap in each k-step
    constexpr bool WR = C::W_IN_REGS;
    f16x8 w_hi[WR ? KS : 1];
    f16x8* wl = reinterpret_cast<f16x8*>(lds + 2 * PLANE);          // [ks][lane] second pieces, then (16 channels) [ks][lane] first pieces
    int toff[KS];
    for (int i = tid; i < KS * 64; i += 256) {
        wl[i] = reinterpret_cast<const f16x8*>(wq)[KS * 64 + i];
        if (!WR) wl[KS * 64 + i] = reinterpret_cast<const f16x8*>(wq)[i];
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (WR) w_hi[ks] = reinterpret_cast<const f16x8*>(wq)[ks * 64 + lane];
        const int kb = ks * 32 + kg * 8, tap = CIN == 16 ? (kb >> 4) : (kb >> 3), c0 = CIN == 16 ? (kb & 15) : 0;
        const int t = tap < NTAP ? tap : 0;                       // the padding k-values multiply zero weights: any address will do
        const int dz = KZ == 3 ? t / 9 : 0, dy = (t / 3) % 3, dx = t % 3;
        toff[ks] = ((dz * HY + dy) * HX + dx) * ROWB + c0 * 2;
    }
    if (guard && blockIdx.x == 0 && tid == 0 && (float)wq[(size_t)2 * KS * 512] != 0.0f) guard[0] = 1;      // status word behind the weights: one was clamped at pack time
    // this thread's channel quad of every staged item, its activation
    const int xq = (tid & (XQ - 1)) * 4;
    f32x4 sc{1.f, 1.f, 1.f, 1.f}, sh{0.f, 0.f, 0.f, 0.f};
    const bool act_on = a.scale != nullptr;
    if (act_on) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sc[j] = a.scale[xq + j] * X_SCALE; sh[j] = a.shift[xq + j] * X_SCALE; }
    }
    typedef unsigned u32x4v __attribute__((__vector_size__(16)));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)((int64_t)Di * Hi * Wi * ld * 4), 0x00020000);
    f32x4 px[NX];
    unsigned mx = 0;
    // Tile-independent half of the prefetch addressing, once per thread: item u = halo voxel (vx, vy, vz) and its byte offset inside the halo box.  (Inside the tile loop
    // the divisions by HX / HY and the three 32-bit multiplies per item were ~200 quarter-rate integer instructions per tile and wave: the ablation of NOTES.md.)
    int loff[NX];
    unsigned vxyz[NX];                                            // vx | vy << 16 | vz << 24; vx = 0x3fff for the items past the halo (never inside the volume)
#pragma unroll
    for (int u = 0; u < NX; ++u) {
        const int v = (tid + 256 * u) / XQ;
        const int vx = v % HX, vy = (v / HX) % HY, vz = v / (HX * HY);
        loff[u] = (((vz * Hi + vy) * Wi + vx) * ld) * 4;
        vxyz[u] = v < NVH ? (unsigned)vx | ((unsigned)vy << 16) | ((unsigned)vz << 24) : 0x3fffu;
    }
    auto prefetch = [&](int tile) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int ix0 = bx * TX * S - 1, iy0 = by * TOY * S - 1, iz0 = KZ == 3 ? bz * TOZ * S - 1 : bz * TOZ;
        const int base = (((iz0 * Hi + iy0) * Wi + ix0) * ld + xq) * 4;       // may be negative (halo origin -1); base + loff is not for a voxel inside the volume
        mx = 0;
#pragma unroll
        for (int u = 0; u < NX; ++u) {
            const unsigned c = vxyz[u];
            const bool in = (unsigned)(ix0 + (int)(c & 0xffffu)) < (unsigned)Wi && (unsigned)(iy0 + (int)((c >> 16) & 0xffu)) < (unsigned)Hi && (unsigned)(iz0 + (int)(c >> 24)) < (unsigned)Di;
            const unsigned off = in ? (unsigned)(base + loff[u]) : 0xffffffffu;
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
            f16x4 h, l;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                // x 2^4 with the pending activation: leaky(16 y) = 16 leaky(y) and fma(x, 16 sc, 16 sh) = 16 fma(x, sc, sh) exactly, so the scale rides in sc / sh;
                // leaky(y) = max(y, 0.01 y).  The zero padding is padding of the ACTIVATED input.  No clamp: a value beyond fp16 becomes inf, its second piece
                // -inf, every output it touches NaN - which the partial sums carry to the guard below (7 operations per value instead of 12).
                float xs = act_on ? fmaf(v[j], sc[j], sh[j]) : v[j] * X_SCALE;
                if (act_on) xs = fmaxf(xs, 0.01f * xs);
                if (!((mx >> u) & 1)) xs = 0.0f;
                const _Float16 p0 = (_Float16)xs;
                h[j] = p0; l[j] = (_Float16)(xs - (float)p0);
            }
            if (it < NVH * XQ) { *reinterpret_cast<f16x4*>(xt + it * 8) = h; *reinterpret_cast<f16x4*>(xt + PLANE + it * 8) = l; }
        }
        __syncthreads();
        if (tile + 1 < t_end) prefetch(tile + 1);
        const char* base[MT_PER_WAVE];
        f32x4 acc[MT_PER_WAVE];
#pragma unroll
        for (int q = 0; q < MT_PER_WAVE; ++q) {
            const int mt = wave * MT_PER_WAVE + q, xh = mt % XT, row = mt / XT, oy_l = row % TOY, oz_l = row / TOY;
            base[q] = xt + ((oz_l * SZ * HY + oy_l * S) * HX + (xh * 16 + m) * S) * ROWB;
            acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // Software-pipelined MFMA phase: the fragments of k-step ks + 1 are requested (second register set) before the MFMAs of ks, whose three piece products run as three
        // passes over the M-tiles (independent accumulators next to each other); sched_barrier fences keep the scheduler from sinking the reads back to their uses (sched_group_barrier alone did not).
        // (What the compiler does on its own - 4 reads, wait, 3 dependent MFMAs per (k-step, M-tile) - costs ~250 cycles per MFMA: NOTES.md.)
        {
            f16x8 ah[2][MT_PER_WAVE], al[2][MT_PER_WAVE], wh[2], wlo[2];
            auto load_k = [&](int ks, int b) {
                wlo[b] = wl[ks * 64 + lane];
                wh[b] = WR ? w_hi[WR ? ks : 0] : wl[(KS + ks) * 64 + lane];
#pragma unroll
                for (int q = 0; q < MT_PER_WAVE; ++q) {
                    ah[b][q] = *reinterpret_cast<const f16x8*>(base[q] + toff[ks]);
                    al[b][q] = *reinterpret_cast<const f16x8*>(base[q] + PLANE + toff[ks]);
                }
            };
            load_k(0, 0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int b = ks & 1;
                if (ks + 1 < KS) load_k(ks + 1, b ^ 1);
                __builtin_amdgcn_sched_barrier(0);                                       // all reads of the next k-step are issued before the MFMAs of this one ...
#pragma unroll
                for (int q = 0; q < MT_PER_WAVE; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[b][q], wh[b], acc[q], 0, 0, 0);     // the small products first
#pragma unroll
                for (int q = 0; q < MT_PER_WAVE; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][q], wlo[b], acc[q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < MT_PER_WAVE; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[b][q], wh[b], acc[q], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);                                       // ... and nothing moves across: the reads stay a whole k-step ahead
            }
        }
#pragma unroll
        for (int q = 0; q < MT_PER_WAVE; ++q) {
            // D: lane (n = lane & 15, g = lane >> 4): acc[r] = voxel 4 g + r of the M-tile, channel n
            const int mt = wave * MT_PER_WAVE + q, xh = mt % XT, row = mt / XT, oy_l = row % TOY, oz_l = row / TOY;
            const int oz = oz0 + oz_l, oy = oy0 + oy_l;
            if (oz < Do && oy < Ho && m < Cout) {
                float* orow = out + (((int64_t)oz * Ho + oy) * Wo) * Cout + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ox = ox0 + xh * 16 + 4 * kg + r;
                    if (ox < Wo) { const float v = acc[q][r] * OUT_SCALE; orow[(int64_t)ox * Cout] = v; s_sum += v; q_sum = fmaf(v, v, q_sum); }
                }
            }
        }
    }
    if (guard && !(fabsf(q_sum) <= 3.0e38f)) guard[0] = 1;         // an operand left fp16's range (or the input held a NaN / inf): outputs are NaN there
    if (stats) {
        s_sum += __shfl_xor(s_sum, 16); q_sum += __shfl_xor(q_sum, 16);
        s_sum += __shfl_xor(s_sum, 32); q_sum += __shfl_xor(q_sum, 32);
        if (kg == 0) { red[wave][0][m] = s_sum; red[wave][1][m] = q_sum; }
        __syncthreads();
        if (tid < 2 * Cout) {
            const int which = tid / Cout, c = tid - which * Cout;
            stats[abn_part_at(which, c, Cout, blockIdx.x, nslots)] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
        }
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < (int64_t)(nslots - (int)gridDim.x) * 2 * Cout; i += (int64_t)gridDim.x * 256) {   // the slots this grid does not own
            const int64_t slot = gridDim.x + i / (2 * Cout);
            const int r = (int)(i % (2 * Cout));
            stats[abn_part_at(r / Cout, r % Cout, Cout, slot, nslots)] = 0.f;
        }
    }
}

// nn weight w[Cout][Cin][ntaps] (fp32) -> wq[piece hi | lo][ks][lane][8] fp16 pieces of w * 2^8: the B fragments (k = 32 ks + 8 (lane >> 4) + j = tap * Cin + ci,
// column lane & 15 = output channel), then 8 status elements ([0] != 0: a weight left fp16's range)
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, int Cin, int Cout, int ntaps, int KS, _Float16* __restrict__ wq)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= KS * 512) return;
    const int j = i & 7, lane = (i >> 3) & 63, ks = i >> 9;
    const int kb = ks * 32 + (lane >> 4) * 8 + j, log = Cin == 8 ? 3 : 4, tap = kb >> log, ci = kb & (Cin - 1), co = lane & 15;
    float v = 0.0f;
    if (tap < ntaps && co < Cout) v = w[((int64_t)co * Cin + ci) * ntaps + tap] * W_SCALE;
    if (!(fabsf(v) <= 65504.0f)) wq[(size_t)2 * KS * 512] = (_Float16)1.0f;
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);
    const _Float16 hi = (_Float16)v;
    wq[i] = hi;
    wq[(size_t)KS * 512 + i] = (_Float16)(v - (float)hi);
}

}  // namespace

extern "C" size_t mvsnerf_conv3d_f16x3_packed_elems(int Cin) { return (Cin == 8 || Cin == 16) ? (size_t)2 * ((27 * Cin + 31) / 32) * 512 + 8 : 0; }

extern "C" int mvsnerf_conv3d_f16x3_pack(const float* w, int Cin, int Cout, void* wq, void* stream)
{
    const int ntaps = 27;
    if (!w || !wq || (Cin != 8 && Cin != 16) || Cout < 1 || Cout > 16) return MVSNERF_EINVAL;
    if (!mvs_aligned16(wq)) return MVSNERF_EALIGN;
    const int KS = (ntaps * Cin + 31) / 32;
    hipError_t e = hipMemsetAsync(reinterpret_cast<_Float16*>(wq) + (size_t)2 * KS * 512, 0, 8 * sizeof(_Float16), (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    pack_kernel<<<mvs_cdiv((int64_t)KS * 512, 256), 256, 0, (hipStream_t)stream>>>(w, Cin, Cout, ntaps, KS, reinterpret_cast<_Float16*>(wq));
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

extern "C" int mvsnerf_conv3d_f16x3_supported(int Cin, int Cout, int stride) { return ((Cin == 8 && stride == 2) || (Cin == 16 && stride == 1)) && Cout >= 1 && Cout <= 16; }

// workgroups of a launch = InPlaceABN partial-sum slots it leaves (persistent: every workgroup walks a contiguous range of tiles)
extern "C" int mvsnerf_conv3d_f16x3_slots(void) { return 512; }

// x: [D][H][W][cin_ld] fp32 raw + pending InPlaceABN (scale / shift may be null), k3 p1; out [Do][Ho][Wo][Cout] fp32; stats_part: 2 * Cout * mvsnerf_conv3d_f16x3_slots()
// floats or null; guard: null, or the guard words of the sequence (guard[0] = 1 when an operand left fp16's range: the outputs are NaN there)
int mvs_conv3d_f16x3_tiled_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int D, int H, int W, const void* wq, int Cout,
                               int stride, float* out, float* stats_part, int* guard, hipStream_t st)
{
    if (!x || !wq || !out || D < 1 || H < 1 || W < 1 || (scale == nullptr) != (shift == nullptr)) return MVSNERF_EINVAL;
    if (!mvsnerf_conv3d_f16x3_supported(Cin, Cout, stride) || (cin_ld & 3) || cin_ld < Cin) return MVSNERF_EUNSUPPORTED;
    if (!mvs_aligned16(x) || !mvs_aligned16(wq)) return MVSNERF_EALIGN;
    if ((int64_t)D * H * W * cin_ld * 4 >= (1ll << 31)) return MVSNERF_EUNSUPPORTED;
    const ActSrc a{x, scale, shift};
    const int Do = (D - 1) / stride + 1, Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int grid = mvsnerf_conv3d_f16x3_slots(), nslots = grid;
    const bool narrow = ((Wo + 15) / 16) * 16 < ((Wo + 31) / 32) * 32;       // rows whose last 32-wide tile would be less than half full
    const _Float16* w = reinterpret_cast<const _Float16*>(wq);
    static unsigned long long cap16 = 0, cap8 = 0, cap16n = 0, cap8n = 0;
    if (narrow && Cin == 16) {
        using Cfg = TiledCfgH<16, 1, 2, 8, 3, 16>;
        if (int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(conv_f16x3_tiled_kernel<16, 1, 2, 8, 3, 16>), Cfg::LDS_BYTES, &cap16n)) return rc;
        conv_f16x3_tiled_kernel<16, 1, 2, 8, 3, 16><<<grid, 256, Cfg::LDS_BYTES, st>>>(a, cin_ld, D, H, W, w, Cout, out, Do, Ho, Wo, stats_part, nslots, guard);
    } else if (narrow && Cin == 8) {
        using Cfg = TiledCfgH<8, 2, 2, 4, 3, 16>;
        if (int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(conv_f16x3_tiled_kernel<8, 2, 2, 4, 3, 16>), Cfg::LDS_BYTES, &cap8n)) return rc;
        conv_f16x3_tiled_kernel<8, 2, 2, 4, 3, 16><<<grid, 256, Cfg::LDS_BYTES, st>>>(a, cin_ld, D, H, W, w, Cout, out, Do, Ho, Wo, stats_part, nslots, guard);
    } else if (Cin == 16) {
        using Cfg = TiledCfgH<16, 1, 2, 4>;
        if (int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(conv_f16x3_tiled_kernel<16, 1, 2, 4>), Cfg::LDS_BYTES, &cap16)) return rc;
        conv_f16x3_tiled_kernel<16, 1, 2, 4><<<grid, 256, Cfg::LDS_BYTES, st>>>(a, cin_ld, D, H, W, w, Cout, out, Do, Ho, Wo, stats_part, nslots, guard);
    } else {
        using Cfg = TiledCfgH<8, 2, 2, 2>;
        if (int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(conv_f16x3_tiled_kernel<8, 2, 2, 2>), Cfg::LDS_BYTES, &cap8)) return rc;
        conv_f16x3_tiled_kernel<8, 2, 2, 2><<<grid, 256, Cfg::LDS_BYTES, st>>>(a, cin_ld, D, H, W, w, Cout, out, Do, Ho, Wo, stats_part, nslots, guard);
    }
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// the UNGUARDED kernel alone (an operand beyond fp16's range leaves NaNs); tests and encoder_precision("fp16x3")
extern "C" int mvsnerf_conv3d_f16x3_fwd(const float* x, const float* scale, const float* shift, int Cin, int cin_ld, int D, int H, int W, const void* wq, int Cout,
                                        int stride, float* out, float* stats_part, void* stream)
{
    return mvs_conv3d_f16x3_tiled_fwd(x, scale, shift, Cin, cin_ld, D, H, W, wq, Cout, stride, out, stats_part, nullptr, (hipStream_t)stream);
}
