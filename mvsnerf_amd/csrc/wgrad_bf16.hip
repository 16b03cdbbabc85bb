// Weight gradients of the encoder's convolutions behind conv0 on v_mfma_f32_16x16x32_bf16 (use_amp: train_mvs_nerf_pl.py:317-318; the nine
// 3-D layers of models.py:725-769 and FeatureNet's 2-D layers, models.py:688-722):
//     gW[a][b][tap] = sum_o G[o][a] * X[o * S - pad + tap][b]          (the definition of mvsnerf_conv3d_wgrad / mvsnerf_conv2d_wgrad)
// G lives on the convolution's output grid (A channels), X on its input grid (B channels), each optionally lazily activated / a sum of two
// tensors; both are rounded to bf16 (round to nearest even) when they are staged, products accumulate in fp32.
//
// The reduction runs over VOXELS, so both MFMA operands want eight consecutive voxels of one channel per lane while the data is channel-last.
// As in conv0's bf16 weight gradient (conv_bf16.hip) the tiles sit in LDS as [voxel][16 channels] bf16 rows and gfx950's transposing read
// (ds_read_b64_tr_b16: the 16 lanes of a group fetch a [4 rows][16 columns] block and lane i receives COLUMN i; any row stride) delivers
//     A = X^T : rows = 16 channels of X-block bb, k = 32 consecutive output positions along x (input voxels S apart), shifted by the tap
//     B = G   : k = the same 32 output voxels, columns = 16 channels of G-block ab
// so D[b][a] is the tap's 16 x 16 block of gW.  A workgroup (4 waves) owns ONE (ab, bb) pair and a range of output tiles (TOZ x TOY x 32);
// wave w owns taps w, w + 4, ... over ALL rows of the tile: one G fragment per row, one X fragment and one MFMA per own tap.  The taps'
// accumulators (7 x 4 registers) persist over the range and the wave writes them straight into its block of partial result `range` in gW's own
// [a][b][tap] layout - deterministic, reduced by mvsnerf_partial_sum_multi like every other weight-gradient kernel.
// Channel counts below 16 are zero-padded (8-channel X tensors get 8-channel LDS rows and a block of zeros for the upper columns).
#include "common.h"
#include "act.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

// lane i of a 16-lane group passes the address of row (i >> 2), columns 4 (i & 3) .. + 3 of a [4][16] block and receives column i
__device__ __forceinline__ bf16x4 tr_read(const char* p)
{
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(const void*)p);
    return __builtin_bit_cast(bf16x4, v);
}

// One thread stages the SAME channel quad of every item it touches (items are tid + 256 u and 256 is a multiple of the quads per voxel), so the
// lazy activation's scale / shift are four registers each, fetched once - not eight dependent global loads behind every staged quad.
struct Act4 { f32x4 sc, sh; bool on; };
__device__ __forceinline__ Act4 act4_of(const ActSrc& s, int c, bool real)
{
    Act4 a;
    a.on = s.scale != nullptr;
    a.sc = f32x4{1.f, 1.f, 1.f, 1.f}; a.sh = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.on && real) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { a.sc[j] = s.scale[c + j]; a.sh[j] = s.shift[c + j]; }
    }
    return a;
}
__device__ __forceinline__ f32x4 act4(f32x4 v, const Act4& a)
{
    if (a.on) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], a.sc[j], a.sh[j]);
    }
    return v;
}
__device__ __forceinline__ bf16x4 to_bf16x4(f32x4 v)
{
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
    return o;
}

// XCH: channels per X row in LDS - 16, or 8 for the 8-channel tensors (conv1 / conv11 read 150 MB of them; the upper eight columns of a
// transposing read then come from a block of zeros)
template <int S, int KZ, int K, int TOZ, int TOY, int XCH>
struct WgCfg {
    static constexpr int TX = 32, NTAP = KZ * K * K, PZ = KZ / 2, P = K / 2, SZ = KZ == 1 ? 1 : S;
    static constexpr int HX = (TX - 1) * S + K, HY = (TOY - 1) * S + K, HZ = (TOZ - 1) * SZ + KZ;
    static constexpr int NVO = TX * TOY * TOZ, NVH = HX * HY * HZ, XRB = XCH * 2;
    static constexpr int TILE_BYTES = NVO * 32 + ((NVH * XRB + 63) & ~63) + 64, RED_BYTES = NTAP * 64 * 16;
    static constexpr int LDS_BYTES = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
};

// MODE 0: any operands, tiles staged in batches of eight loads.  MODE 1 / 2 (single-source X; G single / two-source): the NEXT tile's global
// loads are issued before the current tile's multiplications and land in registers while the matrix cores work - the tile loop is otherwise
// a chain of exposed memory latencies (measured: 22 us per tile on conv1 / conv11).
template <int S, int KZ, int K, int TOZ, int TOY, int XCH, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE ? 2 : 1))) void conv_wgrad_bf16_kernel(ActSrc g1, ActSrc g2, int A, ActSrc x1, ActSrc x2, int B, int ldx,
                                                             int Do, int Ho, int Wo, int Di, int Hi, int Wi, int n_ranges, int NB,
                                                             float* __restrict__ partial)
{
    using C = WgCfg<S, KZ, K, TOZ, TOY, XCH>;
    constexpr int TX = C::TX, NTAP = C::NTAP, PZ = C::PZ, P = C::P, SZ = C::SZ, HX = C::HX, HY = C::HY, NVO = C::NVO, NVH = C::NVH, XRB = C::XRB, XQ = XCH / 4;
    constexpr bool PIPE = MODE != 0, GTWO = MODE != 1;
    // the four waves share the TAPS (wave w owns taps w, w + 4, ...: seven accumulators instead of 27, no cross-wave sum at the end, and the
    // registers that frees hold the prefetched tile); the 1 x 1 layers have one tap and share the tile's rows instead
    constexpr bool TAPSPLIT = NTAP >= 4;
    constexpr int NACC = TAPSPLIT ? (NTAP + 3) / 4 : NTAP;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* gt = lds;                                               // [NVO][16] bf16
    char* xt = lds + NVO * 32;                                    // [NVH][XCH] bf16
    char* zt = xt + ((NVH * XRB + 63) & ~63);                     // 64 bytes of zeros (XCH = 8: columns 8..15 of a transposing read)
    if (threadIdx.x < 16) reinterpret_cast<float*>(zt)[threadIdx.x] = 0.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int range = blockIdx.x, ab = blockIdx.y / NB, bb = blockIdx.y - ab * NB;
    const int a0 = ab * 16, b0 = bb * 16;
    const int nbx = (Wo + TX - 1) / TX, nby = (Ho + TOY - 1) / TOY, nbz = (Do + TOZ - 1) / TOZ;
    const int n_tiles = nbx * nby * nbz;
    const int t_begin = (int)((int64_t)n_tiles * range / n_ranges), t_end = (int)((int64_t)n_tiles * (range + 1) / n_ranges);
    f32x4 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0, 0, 0, 0};
    const int i16 = lane & 15, kg = lane >> 4;
    const int row_in_frag = 8 * kg + (i16 >> 2), chunk = i16 & 3;  // this lane's part of a transposing read: row of the fragment, column chunk
    // ---- this thread's channel quads and their activations
    const int gq = (tid & 3) * 4, xq = (tid & (XQ - 1)) * 4;
    const bool g_ok = a0 + gq < A, x_ok = b0 + xq < B;              // channel counts are multiples of 4: a quad is real or padding as a whole
    const Act4 ga1 = act4_of(g1, a0 + gq, g_ok), ga2 = act4_of(g2, a0 + gq, g_ok && g2.x != nullptr && GTWO);
    const Act4 xa1 = act4_of(x1, b0 + xq, x_ok), xa2 = act4_of(x2, b0 + xq, x_ok && x2.x != nullptr && !PIPE);
    const float* g1p = g1.x + a0 + gq; const float* g2p = g2.x ? g2.x + a0 + gq : nullptr;
    const float* x1p = x1.x + b0 + xq; const float* x2p = x2.x ? x2.x + b0 + xq : nullptr;
    // voxel index of staging item u of a tile (-1: outside the tensor / past the tile / padding quad)
    auto g_vox = [&](int u, int ox0, int oy0, int oz0) -> int {
        const int v = (tid + 256 * u) >> 2;
        const int ox = ox0 + v % TX, oy = oy0 + (v / TX) % TOY, oz = oz0 + v / (TX * TOY);
        return (v < NVO && ox < Wo && oy < Ho && oz < Do && g_ok) ? (oz * Ho + oy) * Wo + ox : -1;
    };
    auto x_vox = [&](int u, int ix0, int iy0, int iz0) -> int {
        const int v = (tid + 256 * u) / XQ;
        const int ix = ix0 + v % HX, iy = iy0 + (v / HX) % HY, iz = iz0 + v / (HX * HY);
        return (v < NVH && ix >= 0 && ix < Wi && iy >= 0 && iy < Hi && iz >= 0 && iz < Di && x_ok) ? (iz * Hi + iy) * Wi + ix : -1;
    };
    const f32x4 zero4{0.f, 0.f, 0.f, 0.f};
    constexpr int NG = (NVO * 4 + 255) / 256, NX = (NVH * XQ + 255) / 256;
    f32x4 pg[PIPE ? NG : 1], pg2[PIPE && GTWO ? NG : 1], px[PIPE ? NX : 1];
    unsigned mg = 0, mx = 0;                                      // which of the prefetched quads are real (the padding is not activated)
    // prefetch through buffer descriptors: a 32-bit byte offset per quad instead of a 64-bit address, and an offset past the end reads zeros -
    // no branch around the load (the host sends tensors of 2 GB and more to MODE 0)
    typedef unsigned u32x4v __attribute__((__vector_size__(16)));
    const auto rs_g1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g1.x), 0, (int)((int64_t)Do * Ho * Wo * A * 4), 0x00020000);
    const auto rs_g2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g2.x ? g2.x : g1.x), 0, (int)((int64_t)Do * Ho * Wo * A * 4), 0x00020000);
    const auto rs_x1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x1.x), 0, (int)((int64_t)Di * Hi * Wi * ldx * 4), 0x00020000);
    auto prefetch = [&](int tile) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int ox0 = bx * TX, oy0 = by * TOY, oz0 = bz * TOZ;
        mg = 0; mx = 0;
#pragma unroll
        for (int u = 0; u < (PIPE ? NG : 0); ++u) {
            const int vox = g_vox(u, ox0, oy0, oz0);
            const unsigned off = vox >= 0 ? (unsigned)(vox * A + a0 + gq) * 4u : 0xffffffffu;
            pg[u] = __builtin_bit_cast(f32x4, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs_g1, off, 0, 0));
            if (GTWO) pg2[GTWO ? u : 0] = __builtin_bit_cast(f32x4, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs_g2, off, 0, 0));
            mg |= (unsigned)(vox >= 0) << u;
        }
#pragma unroll
        for (int u = 0; u < (PIPE ? NX : 0); ++u) {
            const int vox = x_vox(u, ox0 * S - P, oy0 * S - P, oz0 * SZ - PZ);
            const unsigned off = vox >= 0 ? (unsigned)(vox * ldx + b0 + xq) * 4u : 0xffffffffu;
            px[u] = __builtin_bit_cast(f32x4, (u32x4v)__builtin_amdgcn_raw_buffer_load_b128(rs_x1, off, 0, 0));
            mx |= (unsigned)(vox >= 0) << u;
        }
    };
    if constexpr (PIPE) { if (t_begin < t_end) prefetch(t_begin); }
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();                                          // everybody is done with the previous tile
        if constexpr (PIPE) {
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const int it = tid + 256 * u;
                f32x4 v = pg[u];
                if ((mg >> u) & 1) { v = act4(v, ga1); if (GTWO && g2p) v += act4(pg2[GTWO ? u : 0], ga2); }
                if (it < NVO * 4) *reinterpret_cast<bf16x4*>(gt + it * 8) = to_bf16x4(v);
            }
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                const int it = tid + 256 * u;
                f32x4 v = px[u];
                if ((mx >> u) & 1) v = act4(v, xa1);
                if (it < NVH * XQ) *reinterpret_cast<bf16x4*>(xt + it * 8) = to_bf16x4(v);
            }
            __syncthreads();
            if (tile + 1 < t_end) prefetch(tile + 1);
        } else {
            const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
            const int ox0 = bx * TX, oy0 = by * TOY, oz0 = bz * TOZ;
            // item = (voxel, channel quad), consecutive threads -> consecutive 16 bytes; SB items per thread are loaded before the first is converted
            constexpr int SB = 8;
#pragma unroll 1
            for (int u0 = 0; u0 < NG; u0 += SB) {
                f32x4 ra[SB], rb[SB]; bool real[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int vox = u0 + u < NG ? g_vox(u0 + u, ox0, oy0, oz0) : -1;
                    real[u] = vox >= 0; ra[u] = zero4; rb[u] = zero4;
                    if (real[u]) { ra[u] = *reinterpret_cast<const f32x4*>(g1p + (int64_t)vox * A); if (g2p) rb[u] = *reinterpret_cast<const f32x4*>(g2p + (int64_t)vox * A); }
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int it = tid + 256 * (u0 + u);
                    f32x4 v = ra[u];
                    if (real[u]) { v = act4(v, ga1); if (g2p) v += act4(rb[u], ga2); }
                    if (it < NVO * 4) *reinterpret_cast<bf16x4*>(gt + it * 8) = to_bf16x4(v);
                }
            }
#pragma unroll 1
            for (int u0 = 0; u0 < NX; u0 += SB) {
                f32x4 ra[SB], rb[SB]; bool real[SB];
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int vox = u0 + u < NX ? x_vox(u0 + u, ox0 * S - P, oy0 * S - P, oz0 * SZ - PZ) : -1;
                    real[u] = vox >= 0; ra[u] = zero4; rb[u] = zero4;
                    if (real[u]) { ra[u] = *reinterpret_cast<const f32x4*>(x1p + (int64_t)vox * ldx); if (x2p) rb[u] = *reinterpret_cast<const f32x4*>(x2p + (int64_t)vox * ldx); }
                }
#pragma unroll
                for (int u = 0; u < SB; ++u) {
                    const int it = tid + 256 * (u0 + u);
                    f32x4 v = ra[u];
                    if (real[u]) { v = act4(v, xa1); if (x2p) v += act4(rb[u], xa2); }
                    if (it < NVH * XQ) *reinterpret_cast<bf16x4*>(xt + it * 8) = to_bf16x4(v);
                }
            }
            __syncthreads();
        }
        // ---- multiply
        const bool zc = XCH == 8 && chunk >= 2;
        if constexpr (TAPSPLIT) {
            int toff[NACC];                                       // LDS offset of this wave's taps (wave-uniform)
#pragma unroll
            for (int j = 0; j < NACC; ++j) {
                const int tap = 4 * j + wave, dz = tap / (K * K), dy = (tap / K) % K, dx = tap % K;
                toff[j] = ((dz * HY + dy) * HX + dx) * XRB;
            }
#pragma unroll 2
            for (int r = 0; r < TOZ * TOY; ++r) {
                const int oz_l = r / TOY, oy_l = r - oz_l * TOY;
                bf16x8 bg;
                {
                    const char* rowp = gt + (r * TX + row_in_frag) * 32 + chunk * 8;
                    const bf16x4 lo = tr_read(rowp), hi = tr_read(rowp + 4 * 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bg[e] = lo[e]; bg[4 + e] = hi[e]; }
                }
                const char* rowx = xt + ((oz_l * SZ * HY + oy_l * S) * HX + row_in_frag * S) * XRB + chunk * 8;
#pragma unroll
                for (int j = 0; j < NACC; ++j) {
                    if (4 * j + wave < NTAP) {
                        const bf16x4 lo = tr_read(zc ? zt : rowx + toff[j]), hi = tr_read(zc ? zt : rowx + toff[j] + 4 * S * XRB);
                        bf16x8 a;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] = lo[e]; a[4 + e] = hi[e]; }
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bg, acc[j], 0, 0, 0);
                    }
                }
            }
        } else {
#pragma unroll 1
            for (int r = wave; r < TOZ * TOY; r += 4) {
                const int oz_l = r / TOY, oy_l = r - oz_l * TOY;
                bf16x8 bg;
                {
                    const char* rowp = gt + (r * TX + row_in_frag) * 32 + chunk * 8;
                    const bf16x4 lo = tr_read(rowp), hi = tr_read(rowp + 4 * 32);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bg[e] = lo[e]; bg[4 + e] = hi[e]; }
                }
#pragma unroll
                for (int t = 0; t < NTAP; ++t) {
                    const int dz = t / (K * K), dy = (t / K) % K, dx = t % K;
                    const char* rowp = xt + (((oz_l * SZ + dz) * HY + oy_l * S + dy) * HX + dx + row_in_frag * S) * XRB + chunk * 8;
                    const bf16x4 lo = tr_read(zc ? zt : rowp), hi = tr_read(zc ? zt : rowp + 4 * S * XRB);
                    bf16x8 a;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { a[e] = lo[e]; a[4 + e] = hi[e]; }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bg, acc[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- this (ab, bb) block of partial result `range`.  D: lane = (column n = G channel an, row group), register r = row m = X channel 4 (lane >> 4) + r
    float* pr = partial + (int64_t)range * A * B * NTAP;
    if constexpr (TAPSPLIT) {
        const int a = a0 + (lane & 15);
#pragma unroll
        for (int j = 0; j < NACC; ++j) {
            const int tap = 4 * j + wave;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = b0 + 4 * (lane >> 4) + r;
                if (tap < NTAP && a < A && b < B) pr[((int64_t)a * B + b) * NTAP + tap] = acc[j][r];
            }
        }
    } else {
        // fixed-order sum of the four waves (red[tap][lane][r])
        __syncthreads();
        float* red = reinterpret_cast<float*>(lds);
#pragma unroll 1
        for (int w = 0; w < 4; ++w) {
            if (wave == w) {
#pragma unroll
                for (int k = 0; k < NTAP; ++k) {
                    f32x4* dst = reinterpret_cast<f32x4*>(red + (k * 64 + lane) * 4);
                    *dst = w == 0 ? acc[k] : (*dst + acc[k]);
                }
            }
            __syncthreads();
        }
        for (int idx = tid; idx < NTAP * 256; idx += 256) {
            const int an = idx & 15, bm = (idx >> 4) & 15, tap = idx >> 8;
            const int a = a0 + an, b = b0 + bm;
            if (a < A && b < B) pr[((int64_t)a * B + b) * NTAP + tap] = red[(tap * 64 + ((bm >> 2) * 16 + an)) * 4 + (bm & 3)];
        }
    }
}

// Tile (TOZ x TOY x 32 output voxels) per layer shape.  The 16-channel 3-D layers take SMALL tiles (2 x 6, 1 x 2): their prefetched tile then fits
// beside the accumulators at two waves per SIMD, and the pipelined kernel on a tile with a 30 % larger halo (the half- and quarter-resolution
// tensors are L2 / MALL residents) beat the unpipelined kernel on the big tile: 46 -> 42 us (stride 1), 47 -> 37 us (stride 2) per layer.
struct WgShape { int toz, toy; };
__host__ inline bool wg_shape(int kz, int k, int stride, int B, WgShape& s)
{
    if (kz == 3 && k == 3 && stride == 1) { s = B > 8 ? WgShape{2, 6} : WgShape{4, 6}; return true; }
    if (kz == 3 && k == 3 && stride == 2) { s = B > 8 ? WgShape{1, 2} : WgShape{2, 2}; return true; }
    if (kz == 1 && k == 3 && stride == 1) { s = {1, 8}; return true; }
    if (kz == 1 && k == 5 && stride == 2) { s = {1, 4}; return true; }
    if (kz == 1 && k == 1 && stride == 1) { s = {1, 8}; return true; }
    return false;
}

int n_ranges_of(int A, int B, int Do, int Ho, int Wo, const WgShape& s)
{
    const int pairs = ((A + 15) / 16) * ((B + 15) / 16);
    const int64_t n_tiles = (int64_t)((Wo + 31) / 32) * ((Ho + s.toy - 1) / s.toy) * ((Do + s.toz - 1) / s.toz);
    int nr = (B <= 8 && s.toz == 1 ? 1024 : 512) / pairs;        // workgroups resident at once: two per CU (3-D), four (8-channel 2-D layers)
    if (nr < 1) nr = 1;
    return (int)(n_tiles < nr ? n_tiles : nr);
}

}  // namespace

// number of partial results (rows of A * B * kz * k * k floats at the start of the workspace); 0: shape not built
extern "C" int mvsnerf_conv_wgrad_bf16_parts(int A, int B, int Do, int Ho, int Wo, int kz, int k, int stride)
{
    WgShape s;
    if (A < 4 || B < 4 || (A & 3) || (B & 3) || A > 64 || B > 64 || Do < 1 || Ho < 1 || Wo < 1 || !wg_shape(kz, k, stride, B, s)) return 0;
    return n_ranges_of(A, B, Do, Ho, Wo, s);
}

extern "C" size_t mvsnerf_conv_wgrad_bf16_workspace_floats(int A, int B, int kz, int k)
{
    return (size_t)(1024 + MVS_RED_SLICES) * A * B * kz * k * k;
}

extern "C" int mvsnerf_conv_wgrad_bf16(const float* g1, const float* g1_scale, const float* g1_shift,
                                       const float* g2, const float* g2_scale, const float* g2_shift, int A,
                                       const float* x1, const float* x1_scale, const float* x1_shift,
                                       const float* x2, const float* x2_scale, const float* x2_shift, int B, int ldx,
                                       int Do, int Ho, int Wo, int Di, int Hi, int Wi, int kz, int k, int stride,
                                       float* gw, float* workspace, void* stream)
{
    if (!g1 || !x1 || !workspace || Di < 1 || Hi < 1 || Wi < 1) return MVSNERF_EINVAL;
    if ((g1_scale == nullptr) != (g1_shift == nullptr) || (x1_scale == nullptr) != (x1_shift == nullptr)) return MVSNERF_EINVAL;
    if ((g2 && (g2_scale == nullptr) != (g2_shift == nullptr)) || (x2 && (x2_scale == nullptr) != (x2_shift == nullptr))) return MVSNERF_EINVAL;
    WgShape s;
    if (A < 4 || B < 4 || (A & 3) || (B & 3) || A > 64 || B > 64 || !wg_shape(kz, k, stride, B, s)) return MVSNERF_EUNSUPPORTED;
    if ((ldx & 3) || ldx < B || !mvs_aligned16(g1) || !mvs_aligned16(x1) || (g2 && !mvs_aligned16(g2)) || (x2 && !mvs_aligned16(x2))) return MVSNERF_EALIGN;
    const ActSrc G1{g1, g1_scale, g1_shift}, G2{g2, g2_scale, g2_shift}, X1{x1, x1_scale, x1_shift}, X2{x2, x2_scale, x2_shift};
    const int NB = (B + 15) / 16, NA = (A + 15) / 16;
    const int nr = n_ranges_of(A, B, Do, Ho, Wo, s);
    hipStream_t st = (hipStream_t)stream;
    static unsigned long long cap[30] = {};
    // single-source X below 2 GB: the pipelined kernels (32-bit buffer offsets)
    const bool pipe = !x2 && (int64_t)Di * Hi * Wi * ldx * 4 < (1ll << 31) && (int64_t)Do * Ho * Wo * A * 4 < (1ll << 31);
#define MVS_WG_ONE(SLOT, S_, KZ_, K_, TOZ_, TOY_, XCH_, MODE_)                                                                          \
    do {                                                                                                                                \
        constexpr int bytes = WgCfg<S_, KZ_, K_, TOZ_, TOY_, XCH_>::LDS_BYTES;                                                          \
        if (bytes > 48 * 1024)                                                                                                          \
            if (int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(conv_wgrad_bf16_kernel<S_, KZ_, K_, TOZ_, TOY_, XCH_, MODE_>), bytes, &cap[SLOT])) return rc; \
        conv_wgrad_bf16_kernel<S_, KZ_, K_, TOZ_, TOY_, XCH_, MODE_><<<dim3(nr, NA * NB), 256, bytes, st>>>(G1, G2, A, X1, X2, B, ldx, Do, Ho, Wo, Di, Hi, Wi, nr, NB, workspace); \
    } while (0)
    // pipelined where the prefetched tile fits beside the accumulators at two waves per SIMD: the 2-D layers and the 8-channel stride-2 3-D
    // layers (the others would spill 17..239 registers: they take MODE 0, which the tap split already runs at two waves per SIMD)
#define MVS_WG_X(I, S_, KZ_, K_, TOZ_, TOY_, XCH_)                                                                                      \
    do {                                                                                                                                \
        if constexpr (KZ_ == 1 || (S_ == 2 && XCH_ == 8)) {                                                                             \
            if (!pipe) MVS_WG_ONE(6 * I + (XCH_ == 8 ? 0 : 3), S_, KZ_, K_, TOZ_, TOY_, XCH_, 0);                                       \
            else if (!g2) MVS_WG_ONE(6 * I + (XCH_ == 8 ? 1 : 4), S_, KZ_, K_, TOZ_, TOY_, XCH_, 1);                                    \
            else MVS_WG_ONE(6 * I + (XCH_ == 8 ? 2 : 5), S_, KZ_, K_, TOZ_, TOY_, XCH_, 2);                                             \
        } else {                                                                                                                        \
            MVS_WG_ONE(6 * I + (XCH_ == 8 ? 0 : 3), S_, KZ_, K_, TOZ_, TOY_, XCH_, 0);                                                  \
        }                                                                                                                               \
    } while (0)
#define MVS_WG(I, S_, KZ_, K_, TOZ_, TOY_)                                                                                              \
    do {                                                                                                                                \
        if (B <= 8) MVS_WG_X(I, S_, KZ_, K_, TOZ_, TOY_, 8);                                                                            \
        else MVS_WG_X(I, S_, KZ_, K_, TOZ_, TOY_, 16);                                                                                  \
    } while (0)
    if (kz == 3 && stride == 1 && s.toz == 2) {
        if (!pipe || g2) MVS_WG_ONE(24, 1, 3, 3, 2, 6, 16, 0); else MVS_WG_ONE(25, 1, 3, 3, 2, 6, 16, 1);
    } else if (kz == 3 && stride == 2 && s.toz == 1) {
        if (!pipe) MVS_WG_ONE(26, 2, 3, 3, 1, 2, 16, 0); else if (!g2) MVS_WG_ONE(27, 2, 3, 3, 1, 2, 16, 1); else MVS_WG_ONE(28, 2, 3, 3, 1, 2, 16, 2);
    } else if (kz == 3 && stride == 1) MVS_WG(0, 1, 3, 3, 4, 6);
    else if (kz == 3) MVS_WG(1, 2, 3, 3, 2, 2);
    else if (k == 3) MVS_WG(2, 1, 1, 3, 1, 8);
    else if (k == 5) MVS_WG(3, 2, 1, 5, 1, 4);
    else MVS_WG(4, 1, 1, 1, 1, 8);
#undef MVS_WG_ONE
#undef MVS_WG_X
#undef MVS_WG
    MVS_LAUNCH_CHECK();
    if (!gw) return MVSNERF_OK;                                   // partials left for mvsnerf_partial_sum_multi
    const int64_t n_out = (int64_t)A * B * kz * k * k;
    mvs_partial_sum(workspace, nr, n_out, workspace + (size_t)1024 * n_out, gw, st);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
