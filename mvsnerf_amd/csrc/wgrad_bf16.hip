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
// a wave takes every fourth (z, y) row of the tile: one G fragment per row, one X fragment and one MFMA per tap.  The taps' accumulators
// (27 x 4 registers) persist over the range; the four waves are summed in a fixed order and the workgroup writes its block of partial result
// `range` in gW's own [a][b][tap] layout - deterministic, reduced by mvsnerf_partial_sum_multi like every other weight-gradient kernel.
// Channel counts below 16 (the 8-channel layers) are zero-padded in LDS.
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

// four channels c .. c + 3 of voxel `vox` of a lazily-activated (two-source) tensor with channel stride ld: raw loads (issued in batches so that
// many are in flight), then activation, sum and the rounding to bf16
struct Raw4 { f32x4 a, b; };
__device__ __forceinline__ Raw4 stage_load(const ActSrc& s1, const ActSrc& s2, int64_t vox, int ld, int c, bool real)
{
    Raw4 r;
    r.a = f32x4{0.f, 0.f, 0.f, 0.f}; r.b = r.a;
    if (real) {                                                   // channel counts are multiples of 4: a quad is real or padding as a whole
        r.a = *reinterpret_cast<const f32x4*>(s1.x + vox * ld + c);
        if (s2.x) r.b = *reinterpret_cast<const f32x4*>(s2.x + vox * ld + c);
    }
    return r;
}
__device__ __forceinline__ bf16x4 stage_finish(const ActSrc& s1, const ActSrc& s2, Raw4 r, int c, bool real)
{
    f32x4 v = r.a;
    if (real) {
        if (s1.scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = act_apply(v[j], s1.scale[c + j], s1.shift[c + j]);
        }
        if (s2.x) {
            f32x4 t = r.b;
            if (s2.scale) {
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = act_apply(t[j], s2.scale[c + j], s2.shift[c + j]);
            }
            v += t;
        }
    }
    bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (__bf16)v[j];
    return o;
}

template <int S, int KZ, int K, int TOZ, int TOY>
struct WgCfg {
    static constexpr int TX = 32, NTAP = KZ * K * K, PZ = KZ / 2, P = K / 2, SZ = KZ == 1 ? 1 : S;
    static constexpr int HX = (TX - 1) * S + K, HY = (TOY - 1) * S + K, HZ = (TOZ - 1) * SZ + KZ;
    static constexpr int NVO = TX * TOY * TOZ, NVH = HX * HY * HZ;
    static constexpr int TILE_BYTES = (NVO + NVH) * 32, RED_BYTES = NTAP * 64 * 16;
    static constexpr int LDS_BYTES = TILE_BYTES > RED_BYTES ? TILE_BYTES : RED_BYTES;
};

template <int S, int KZ, int K, int TOZ, int TOY>
__global__ __launch_bounds__(256) void conv_wgrad_bf16_kernel(ActSrc g1, ActSrc g2, int A, ActSrc x1, ActSrc x2, int B, int ldx,
                                                             int Do, int Ho, int Wo, int Di, int Hi, int Wi, int n_ranges, int NB,
                                                             float* __restrict__ partial)
{
    using C = WgCfg<S, KZ, K, TOZ, TOY>;
    constexpr int TX = C::TX, NTAP = C::NTAP, PZ = C::PZ, P = C::P, SZ = C::SZ, HX = C::HX, HY = C::HY, NVO = C::NVO, NVH = C::NVH;
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    char* gt = lds;                                               // [NVO][16] bf16
    char* xt = lds + NVO * 32;                                    // [NVH][16] bf16
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int range = blockIdx.x, ab = blockIdx.y / NB, bb = blockIdx.y - ab * NB;
    const int a0 = ab * 16, b0 = bb * 16;
    const int nbx = (Wo + TX - 1) / TX, nby = (Ho + TOY - 1) / TOY, nbz = (Do + TOZ - 1) / TOZ;
    const int n_tiles = nbx * nby * nbz;
    const int t_begin = (int)((int64_t)n_tiles * range / n_ranges), t_end = (int)((int64_t)n_tiles * (range + 1) / n_ranges);
    f32x4 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[t] = f32x4{0, 0, 0, 0};
    const int i16 = lane & 15, kg = lane >> 4;
    const int row_in_frag = 8 * kg + (i16 >> 2), chunk = i16 & 3;  // this lane's part of a transposing read: row of the fragment, column chunk
#pragma unroll 1
    for (int tile = t_begin; tile < t_end; ++tile) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int ox0 = bx * TX, oy0 = by * TOY, oz0 = bz * TOZ;
        const int ix0 = ox0 * S - P, iy0 = oy0 * S - P, iz0 = oz0 * SZ - PZ;
        __syncthreads();                                          // everybody is done with the previous tile
        // ---- stage G, then the X halo (the zero padding is not activated): item = (voxel, channel quad), consecutive threads -> consecutive
        // 16 bytes of a voxel's 16-channel block; SB items per thread are loaded before the first is converted
        constexpr int SB = 8;
#pragma unroll 1
        for (int it0 = tid; it0 < NVO * 4; it0 += 256 * SB) {
            Raw4 raw[SB]; bool real[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int it = it0 + 256 * u, v = it >> 2, c4 = (it & 3) * 4;
                const int ox = ox0 + v % TX, oy = oy0 + (v / TX) % TOY, oz = oz0 + v / (TX * TOY);
                real[u] = it < NVO * 4 && ox < Wo && oy < Ho && oz < Do && a0 + c4 < A;
                raw[u] = stage_load(g1, g2, real[u] ? ((int64_t)oz * Ho + oy) * Wo + ox : 0, A, a0 + c4, real[u]);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int it = it0 + 256 * u;
                if (it < NVO * 4) *reinterpret_cast<bf16x4*>(gt + (it >> 2) * 32 + (it & 3) * 8) = stage_finish(g1, g2, raw[u], a0 + (it & 3) * 4, real[u]);
            }
        }
#pragma unroll 1
        for (int it0 = tid; it0 < NVH * 4; it0 += 256 * SB) {
            Raw4 raw[SB]; bool real[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int it = it0 + 256 * u, v = it >> 2, c4 = (it & 3) * 4;
                const int ix = ix0 + v % HX, iy = iy0 + (v / HX) % HY, iz = iz0 + v / (HX * HY);
                real[u] = it < NVH * 4 && ix >= 0 && ix < Wi && iy >= 0 && iy < Hi && iz >= 0 && iz < Di && b0 + c4 < B;
                raw[u] = stage_load(x1, x2, real[u] ? ((int64_t)iz * Hi + iy) * Wi + ix : 0, ldx, b0 + c4, real[u]);
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int it = it0 + 256 * u;
                if (it < NVH * 4) *reinterpret_cast<bf16x4*>(xt + (it >> 2) * 32 + (it & 3) * 8) = stage_finish(x1, x2, raw[u], b0 + (it & 3) * 4, real[u]);
            }
        }
        __syncthreads();
        // ---- multiply: this wave's (z, y) rows of the tile
#pragma unroll 1
        for (int r = wave; r < TOZ * TOY; r += 4) {
            const int oz_l = r / TOY, oy_l = r - oz_l * TOY;
            bf16x8 bg;
            {
                const char* rowp = gt + ((oz_l * TOY + oy_l) * TX + row_in_frag) * 32 + chunk * 8;
                const bf16x4 lo = tr_read(rowp), hi = tr_read(rowp + 4 * 32);
#pragma unroll
                for (int e = 0; e < 4; ++e) { bg[e] = lo[e]; bg[4 + e] = hi[e]; }
            }
#pragma unroll
            for (int dz = 0; dz < KZ; ++dz)
#pragma unroll
                for (int dy = 0; dy < K; ++dy)
#pragma unroll
                    for (int dx = 0; dx < K; ++dx) {
                        const char* rowp = xt + (((oz_l * SZ + dz) * HY + oy_l * S + dy) * HX + dx + row_in_frag * S) * 32 + chunk * 8;
                        const bf16x4 lo = tr_read(rowp), hi = tr_read(rowp + 4 * S * 32);
                        bf16x8 a;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a[e] = lo[e]; a[4 + e] = hi[e]; }
                        acc[(dz * K + dy) * K + dx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bg, acc[(dz * K + dy) * K + dx], 0, 0, 0);
                    }
        }
    }
    // fixed-order sum of the four waves (red[tap][lane][r]), then this (ab, bb) block of partial `range`
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
    float* pr = partial + (int64_t)range * A * B * NTAP;
    for (int idx = tid; idx < NTAP * 256; idx += 256) {
        const int an = idx & 15, bm = (idx >> 4) & 15, tap = idx >> 8;
        const int a = a0 + an, b = b0 + bm;
        // D: lane (column n = G channel an, row group bm / 4), register bm % 4 = row m = X channel bm
        if (a < A && b < B) pr[((int64_t)a * B + b) * NTAP + tap] = red[(tap * 64 + ((bm >> 2) * 16 + an)) * 4 + (bm & 3)];
    }
}

struct WgShape { int toz, toy; };
__host__ inline bool wg_shape(int kz, int k, int stride, WgShape& s)
{
    if (kz == 3 && k == 3 && stride == 1) { s = {4, 6}; return true; }
    if (kz == 3 && k == 3 && stride == 2) { s = {2, 2}; return true; }
    if (kz == 1 && k == 3 && stride == 1) { s = {1, 8}; return true; }
    if (kz == 1 && k == 5 && stride == 2) { s = {1, 4}; return true; }
    if (kz == 1 && k == 1 && stride == 1) { s = {1, 8}; return true; }
    return false;
}

int n_ranges_of(int A, int B, int Do, int Ho, int Wo, const WgShape& s)
{
    const int pairs = ((A + 15) / 16) * ((B + 15) / 16);
    const int64_t n_tiles = (int64_t)((Wo + 31) / 32) * ((Ho + s.toy - 1) / s.toy) * ((Do + s.toz - 1) / s.toz);
    int nr = 512 / pairs;
    if (nr < 1) nr = 1;
    return (int)(n_tiles < nr ? n_tiles : nr);
}

}  // namespace

// number of partial results (rows of A * B * kz * k * k floats at the start of the workspace); 0: shape not built
extern "C" int mvsnerf_conv_wgrad_bf16_parts(int A, int B, int Do, int Ho, int Wo, int kz, int k, int stride)
{
    WgShape s;
    if (A < 4 || B < 4 || (A & 3) || (B & 3) || A > 64 || B > 64 || Do < 1 || Ho < 1 || Wo < 1 || !wg_shape(kz, k, stride, s)) return 0;
    return n_ranges_of(A, B, Do, Ho, Wo, s);
}

extern "C" size_t mvsnerf_conv_wgrad_bf16_workspace_floats(int A, int B, int kz, int k)
{
    return (size_t)(512 + MVS_RED_SLICES) * A * B * kz * k * k;
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
    if (A < 4 || B < 4 || (A & 3) || (B & 3) || A > 64 || B > 64 || !wg_shape(kz, k, stride, s)) return MVSNERF_EUNSUPPORTED;
    if ((ldx & 3) || ldx < B || !mvs_aligned16(g1) || !mvs_aligned16(x1) || (g2 && !mvs_aligned16(g2)) || (x2 && !mvs_aligned16(x2))) return MVSNERF_EALIGN;
    const ActSrc G1{g1, g1_scale, g1_shift}, G2{g2, g2_scale, g2_shift}, X1{x1, x1_scale, x1_shift}, X2{x2, x2_scale, x2_shift};
    const int NB = (B + 15) / 16, NA = (A + 15) / 16;
    const int nr = n_ranges_of(A, B, Do, Ho, Wo, s);
    hipStream_t st = (hipStream_t)stream;
    static unsigned long long cap[5] = {0, 0, 0, 0, 0};
#define MVS_WG(I, S_, KZ_, K_, TOZ_, TOY_)                                                                                              \
    do {                                                                                                                                \
        constexpr int bytes = WgCfg<S_, KZ_, K_, TOZ_, TOY_>::LDS_BYTES;                                                                \
        if (bytes > 48 * 1024)                                                                                                          \
            if (int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(conv_wgrad_bf16_kernel<S_, KZ_, K_, TOZ_, TOY_>), bytes, &cap[I])) return rc; \
        conv_wgrad_bf16_kernel<S_, KZ_, K_, TOZ_, TOY_><<<dim3(nr, NA * NB), 256, bytes, st>>>(G1, G2, A, X1, X2, B, ldx, Do, Ho, Wo, Di, Hi, Wi, nr, NB, workspace); \
    } while (0)
    if (kz == 3 && stride == 1) MVS_WG(0, 1, 3, 3, 4, 6);
    else if (kz == 3) MVS_WG(1, 2, 3, 3, 2, 2);
    else if (k == 3) MVS_WG(2, 1, 1, 3, 1, 8);
    else if (k == 5) MVS_WG(3, 2, 1, 5, 1, 4);
    else MVS_WG(4, 1, 1, 1, 1, 8);
#undef MVS_WG
    MVS_LAUNCH_CHECK();
    if (!gw) return MVSNERF_OK;                                   // partials left for mvsnerf_partial_sum_multi
    const int64_t n_out = (int64_t)A * B * kz * k * k;
    mvs_partial_sum(workspace, nr, n_out, workspace + (size_t)512 * n_out, gw, st);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
