// Weight gradients of CostRegNet's 3x3x3 convolutions with 16 / 32 / 64 "a" channels on v_mfma_f32_4x4x1_16B_f32 (fp32 in, fp32
// accumulate, exact fp32 products):
//   gW[a][b][tap] = sum_o G[o][a] * X[o * S - 1 + tap][b]          (mvsnerf_conv3d_wgrad; S = 1 | 2)
// The output voxels o are the k dimension.  The 16 blocks of one instruction are (voxel slot mb) x (quad of `a` channels nb): 64 / A
// voxels x A channels, so the B operand is 64 contiguous floats of the channel-last G tile and nothing is padded for any A in
// {16, 32, 64}; the A operand is X[.][4 cg .. 4 cg + 3] of the same voxels shifted by the tap (broadcast over nb).  D[r][j] of block
// (mb, nb) is the partial gW[a = 4 nb + j][b = 4 cg + r][tap] of voxel slot mb: 27 accumulators (108 registers) hold every tap of one
// block of four `b` channels.  A wave owns one such block; a workgroup (NW waves) shares the staged tiles; grid.y covers B / (4 NW).
// Marching along y the taps slide over the same X rows: a step reads one G operand and one (S = 1) or two (S = 2) new X rows of
// 3 x 3 operands for 27 MFMAs.  Accumulators stay in registers across all tiles a workgroup visits; one partial result per
// workgroup (deterministic; summed by mvs_partial_sum / mvsnerf_partial_sum_multi).
// The VALU kernel this replaces (conv3d_wgrad_rows_kernel) ran the six half- and quarter-resolution layers at 13 - 25 TFLOP/s and
// wrote one partial result per 1 - 14 output rows (up to 150 MB per layer).
#include "common.h"
#include "act.h"

namespace {

constexpr int WTX = 8;                                                // output tile width; height and depth are template parameters

// KZ x K x K taps (3,3 for the 3-D layers; 1,3 / 1,5 for FeatureNet's 2-D layers with the images as z), padding K / 2 (KZ / 2 along z)
template <int A, int NCG, int NQ, int S, int TOZ, int TOY, int KZ, int K>      // NCG blocks of four b channels x NQ work shares = 4 waves
__global__ __launch_bounds__(256, 2) void conv_wgrad_mfma4_kernel(ActSrc g1, ActSrc g2, ActSrc x1, int ldx, int B,
                                                                 int Do, int Ho, int Wo, int Di, int Hi, int Wi, float* __restrict__ partial)
{
    static_assert(NCG * NQ == 4, "four waves");
    constexpr int VM = 64 / A;                                        // voxels per MFMA
    constexpr int NXG = WTX / VM;                                     // x groups per tile row
    constexpr int SZ = KZ == 1 ? 1 : S;                               // the images of a 2-D layer are not strided
    constexpr int HX = (WTX - 1) * S + K, HY = (TOY - 1) * S + K, HZ = (TOZ - 1) * SZ + KZ;
    constexpr int NVH = HX * HY * HZ, NVO = WTX * TOY * TOZ;
    constexpr int NTAP = KZ * K * K, PZ = KZ / 2, P = K / 2;
    constexpr int NT = 256;
    constexpr int XIT = (NCG * NVH + NT - 1) / NT, GIT = (NVO * (A / 4) + NT - 1) / NT;      // staging slots per thread
    constexpr int XB = A == 16 ? 8 : 4;                               // X loads in flight together (register budget: accumulators + window)
    constexpr int YSEG = (TOZ * NXG >= NQ) ? TOY : TOY / 4;           // a work item = (oz, x group) column x YSEG output rows
    constexpr int NSEG = TOY / YSEG, NITEM = TOZ * NXG * NSEG;
    __shared__ __attribute__((aligned(16))) float lds[NVO * A + NCG * NVH * 4];
    float* gt = lds;                                                  // [o voxel][A]
    float* xt = lds + NVO * A;                                        // [cg local][halo voxel][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cgl = wave % NCG, qpart = wave / NCG;
    const int cg0 = blockIdx.y * NCG;
    const bool active = (cg0 + cgl) * 4 < B;                          // (B = 8 with four blocks per workgroup: two waves only stage)
    const int nbx = (Wo + WTX - 1) / WTX, nby = (Ho + TOY - 1) / TOY, nbz = (Do + TOZ - 1) / TOZ;
    const int ntiles = nbx * nby * nbz;
    // tile-invariant staging slots of this thread.  X: item = tid + 256 j -> (cg local, halo voxel) packed hx | hy << 8 | hz << 16 | cgl << 24
    int xpk[XIT];
#pragma unroll
    for (int j = 0; j < XIT; ++j) {
        const int it = tid + NT * j, c = it / NVH, hv = it - c * NVH;
        xpk[j] = (it < NCG * NVH && (cg0 + c) * 4 < B) ? ((hv % HX) | (((hv / HX) % HY) << 8) | ((hv / (HX * HY)) << 16) | (c << 24)) : -1;
    }
    f32x4 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t) acc[t] = f32x4{0, 0, 0, 0};
    const float* gl = gt + lane;                                      // B operand: 64 contiguous floats per voxel group
    const float* xl = xt + (cgl * NVH + (lane / A) * S) * 4 + (lane & 3);       // A operand: voxel slot mb = lane / A, channel i = lane & 3
#pragma unroll 1
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int bx = tile % nbx, by = (tile / nbx) % nby, bz = tile / (nbx * nby);
        const int ox0 = bx * WTX, oy0 = by * TOY, oz0 = bz * TOZ;
        const int ix0 = ox0 * S - P, iy0 = oy0 * S - P, iz0 = oz0 * SZ - PZ;
        const bool interior = ix0 >= 0 && ix0 + HX <= Wi && iy0 >= 0 && iy0 + HY <= Hi && iz0 >= 0 && iz0 + HZ <= Di;      // wave-uniform
        __syncthreads();                                              // everybody finished reading the previous tile
        // ---- stage the X halo: all loads of a batch in flight together, then (activation ->) LDS
#pragma unroll
        for (int j0 = 0; j0 < XIT; j0 += XB) {
            f32x4 val[XB];
#pragma unroll
            for (int j = j0; j < j0 + XB && j < XIT; ++j) {
                const int hx = xpk[j] & 255, hy = (xpk[j] >> 8) & 255, hz = (xpk[j] >> 16) & 255, c = (xpk[j] >> 24) & 7;
                const int ix = ix0 + hx, iy = iy0 + hy, iz = iz0 + hz;
                const bool ok = xpk[j] >= 0 && (interior || (ix >= 0 && ix < Wi && iy >= 0 && iy < Hi && iz >= 0 && iz < Di));
                val[j - j0] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok) val[j - j0] = *reinterpret_cast<const f32x4*>(x1.x + (((int64_t)iz * Hi + iy) * Wi + ix) * ldx + (cg0 + c) * 4);
            }
#pragma unroll
            for (int j = j0; j < j0 + XB && j < XIT; ++j) {
                const int it = tid + NT * j;
                if (it >= NCG * NVH) continue;
                f32x4 v = val[j - j0];
                if (x1.scale && xpk[j] >= 0) {
                    const int cb = (cg0 + ((xpk[j] >> 24) & 7)) * 4;
                    const int hx = xpk[j] & 255, hy = (xpk[j] >> 8) & 255, hz = (xpk[j] >> 16) & 255;
                    const int ix = ix0 + hx, iy = iy0 + hy, iz = iz0 + hz;
                    if (interior || (ix >= 0 && ix < Wi && iy >= 0 && iy < Hi && iz >= 0 && iz < Di)) {       // the zero padding is not activated
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (cb + k < B) v[k] = act_apply(v[k], x1.scale[cb + k], x1.shift[cb + k]);
                    }
                }
                *reinterpret_cast<f32x4*>(xt + it * 4) = v;
            }
        }
        // ---- stage G (activation / skip sum applied here)
#pragma unroll 2
        for (int j = 0; j < GIT; ++j) {
            const int it = tid + NT * j;
            if (it >= NVO * (A / 4)) continue;
            const int v = it / (A / 4), c4 = (it - v * (A / 4)) * 4;
            const int ox = ox0 + v % WTX, oy = oy0 + (v / WTX) % TOY, oz = oz0 + v / (WTX * TOY);
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (ox < Wo && oy < Ho && oz < Do) load_act4<A>(g1, g2, ((int64_t)oz * Ho + oy) * Wo + ox, A, c4, val);
            *reinterpret_cast<f32x4*>(gt + v * A + c4) = val;
        }
        __syncthreads();
        if (!active) continue;
        // ---- multiply: this wave's work items, each marching along oy
#pragma unroll 1
        for (int item = qpart; item < NITEM; item += NQ) {
            const int col = item / NSEG, seg = item - col * NSEG;
            const int oz = col / NXG, xg = col - oz * NXG, y0 = seg * YSEG;
            const float* gb = gl + ((oz * TOY + y0) * WTX + xg * VM) * A;
            const float* xb = xl + ((oz * SZ * HY + y0 * S) * HX + xg * VM * S) * 4;
            float win[K][KZ][K];                                      // [halo row % K][dz][dx]
            auto load_row = [&](int hy) {
#pragma unroll
                for (int a = 0; a < KZ; ++a)
#pragma unroll
                    for (int c = 0; c < K; ++c) win[hy % K][a][c] = xb[((a * HY + hy) * HX + c) * 4];
            };
#pragma unroll
            for (int r = 0; r < K - S; ++r) load_row(r);
#pragma unroll
            for (int oy = 0; oy < YSEG; ++oy) {
                // rows S oy .. S oy + K - 1: all but the last S are here already
#pragma unroll
                for (int r = K - S; r < K; ++r) load_row(S * oy + r);
                const float gcur = gb[oy * WTX * A];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < K; ++b)
#pragma unroll
                    for (int a = 0; a < KZ; ++a)
#pragma unroll
                        for (int c = 0; c < K; ++c)
                            acc[(a * K + b) * K + c] = __builtin_amdgcn_mfma_f32_4x4x1f32(win[(S * oy + b) % K][a][c], gcur, acc[(a * K + b) * K + c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // ---- one partial result per wave share: row blockIdx.x * NQ + qpart of partial[.][A][B][NTAP].  After folding the voxel slots (lane
    // bits above log2 A) lane a < A holds gW[a][4 cg + r][tap]: per `a` that is 4 NTAP CONTIGUOUS floats of the row, but the lanes are
    // B * NTAP floats apart - stored directly, every instruction would be 64 separate 4-byte writes (6912 of them per wave for A = 64:
    // the deep layers spent more time here than multiplying).  So: through LDS, RC `a` rows at a time, and out as whole rows.
    __syncthreads();                                                  // nobody reads the tiles any more
    constexpr int ROWF = 4 * NTAP;                                    // floats per (a, channel block)
    constexpr int WBUD = (NVO * A + NCG * NVH * 4) / 4;               // LDS floats per wave
    constexpr int RC0 = WBUD / ROWF < A ? WBUD / ROWF : A;
    constexpr int RC = RC0 >= 64 ? 64 : RC0 >= 32 ? 32 : RC0 >= 16 ? 16 : RC0 >= 8 ? 8 : RC0 >= 4 ? 4 : RC0 >= 2 ? 2 : 1;
    static_assert(RC0 >= 1, "LDS too small for the epilogue");
    if (!active) return;
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = acc[t][r];
            if (VM >= 2) v += __shfl_xor(v, 32);
            if (VM >= 4) v += __shfl_xor(v, 16);
            if (VM >= 8) v += __shfl_xor(v, 8);
            acc[t][r] = v;
        }
    float* tw = lds + wave * WBUD;
    const int cb = (cg0 + cgl) * 4;
    float* prow = partial + ((int64_t)blockIdx.x * NQ + qpart) * A * B * NTAP + (int64_t)cb * NTAP;
    const int ncol = (B - cb >= 4 ? 4 : B - cb) * NTAP;               // columns of the block that exist (B = 3: three of four)
#pragma unroll 1
    for (int a0 = 0; a0 < A; a0 += RC) {
        if (lane >= a0 && lane < a0 + RC) {
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) tw[(lane - a0) * ROWF + r * NTAP + t] = acc[t][r];
        }
        // (same wave: the LDS writes above are complete before the reads below are served)
        for (int e = lane; e < RC * ROWF; e += 64) {
            const int row = e / ROWF, colx = e - row * ROWF;
            if (colx < ncol) prow[(int64_t)(a0 + row) * B * NTAP + colx] = tw[e];
        }
    }
}

template <int A, int NCG, int NQ, int S, int TOZ, int TOY, int KZ, int K>
int launch(const ActSrc& g1, const ActSrc& g2, const ActSrc& x1, int ldx, int B, int Do, int Ho, int Wo, int Di, int Hi, int Wi, float* partial, int nx,
           hipStream_t st)
{
    const dim3 grid(nx, ((B + 3) / 4 + NCG - 1) / NCG);
    conv_wgrad_mfma4_kernel<A, NCG, NQ, S, TOZ, TOY, KZ, K><<<grid, 256, 0, st>>>(g1, g2, x1, ldx, B, Do, Ho, Wo, Di, Hi, Wi, partial);
    return MVSNERF_OK;
}

// workgroups along x: two resident per CU, every workgroup the same number of tiles (the last one may take fewer)
int balanced_nx(int ntiles, int gy, int nq, int cap_parts)
{
    int nmax = 512 / gy;
    if (nmax * nq > cap_parts) nmax = cap_parts / nq;
    if (nmax < 1) return 0;
    const int rounds = (ntiles + nmax - 1) / nmax;
    return (ntiles + rounds - 1) / rounds;
}

}  // namespace

// ---- 3-D (CostRegNet).  Number of partial results the matrix-core weight gradient leaves for (A, B, stride) on an output grid
// Do x Ho x Wo (workgroups along x times the work shares of a workgroup); 0 = this combination takes the VALU kernel
static int wgrad_mfma4_nx(int A, int B, int Do, int Ho, int Wo, int stride, int cap_parts)
{
    const bool ok = (A == 16 && (B == 8 || B == 16)) || (A == 32 && (B == 16 || B == 32)) || (A == 64 && (B == 32 || B == 64));
    if (!ok || (stride != 1 && stride != 2)) return 0;
    const int toz = stride == 1 ? 2 : 1;
    const int ntiles = ((Wo + WTX - 1) / WTX) * ((Ho + 7) / 8) * ((Do + toz - 1) / toz);
    const int ncg = B == 8 ? 2 : 4, nq = 4 / ncg, gy = (B / 4 + ncg - 1) / ncg;
    return balanced_nx(ntiles, gy, nq, cap_parts);
}

int mvs_conv3d_wgrad_mfma4_parts(int A, int B, int Do, int Ho, int Wo, int stride, int cap_parts)
{
    return wgrad_mfma4_nx(A, B, Do, Ho, Wo, stride, cap_parts) * (B == 8 ? 2 : 1);
}

int mvs_conv3d_wgrad_mfma4(const ActSrc& g1, const ActSrc& g2, int A, const ActSrc& x1, const ActSrc& x2, int B, int ldx, int Do, int Ho, int Wo,
                           int Di, int Hi, int Wi, int stride, float* partial, int cap_parts, hipStream_t st)
{
    if (x2.x) return MVSNERF_EUNSUPPORTED;
    const int nx = wgrad_mfma4_nx(A, B, Do, Ho, Wo, stride, cap_parts);
    if (nx <= 0) return MVSNERF_EUNSUPPORTED;
#define MVS_WM(A_, NCG_, NQ_, S_, TOZ_) launch<A_, NCG_, NQ_, S_, TOZ_, 8, 3, 3>(g1, g2, x1, ldx, B, Do, Ho, Wo, Di, Hi, Wi, partial, nx, st)
    const int key = (A * 100 + B) * 10 + stride;
    switch (key) {
        case (16 * 100 + 8) * 10 + 2:  MVS_WM(16, 2, 2, 2, 1); break;     // conv1, conv11^T
        case (16 * 100 + 8) * 10 + 1:  MVS_WM(16, 2, 2, 1, 2); break;
        case (16 * 100 + 16) * 10 + 1: MVS_WM(16, 4, 1, 1, 2); break;     // conv2
        case (16 * 100 + 16) * 10 + 2: MVS_WM(16, 4, 1, 2, 1); break;
        case (32 * 100 + 16) * 10 + 2: MVS_WM(32, 4, 1, 2, 1); break;     // conv3, conv9^T
        case (32 * 100 + 16) * 10 + 1: MVS_WM(32, 4, 1, 1, 2); break;
        case (32 * 100 + 32) * 10 + 1: MVS_WM(32, 4, 1, 1, 2); break;     // conv4
        case (32 * 100 + 32) * 10 + 2: MVS_WM(32, 4, 1, 2, 1); break;
        case (64 * 100 + 32) * 10 + 2: MVS_WM(64, 4, 1, 2, 1); break;     // conv5, conv7^T
        case (64 * 100 + 32) * 10 + 1: MVS_WM(64, 4, 1, 1, 2); break;
        case (64 * 100 + 64) * 10 + 1: MVS_WM(64, 4, 1, 1, 2); break;     // conv6
        case (64 * 100 + 64) * 10 + 2: MVS_WM(64, 4, 1, 2, 1); break;
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_WM
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

// ---- 2-D (FeatureNet, models.py:688-722): N images as the z axis, k x k taps.  (A, B, k, stride) of its eight convolutions:
//   conv0.0 (8, 3, 3, 1)  conv0.1 (8, 8, 3, 1)  conv1.0 (16, 8, 5, 2)  conv1.1/2 (16, 16, 3, 1)  conv2.0 (32, 16, 5, 2)  conv2.1/2 (32, 32, 3, 1)
static int wgrad2d_mfma4_cfg(int A, int B, int ksize, int stride, int& ncg, int& toy)
{
    const int key = ((A * 100 + B) * 10 + ksize) * 10 + stride;
    ncg = B <= 4 ? 1 : (B == 8 ? 2 : 4);
    toy = ksize == 3 ? 16 : 8;
    switch (key) {
        case ((8 * 100 + 3) * 10 + 3) * 10 + 1: case ((8 * 100 + 8) * 10 + 3) * 10 + 1: case ((16 * 100 + 8) * 10 + 5) * 10 + 2:
        case ((16 * 100 + 16) * 10 + 3) * 10 + 1: case ((32 * 100 + 16) * 10 + 5) * 10 + 2: case ((32 * 100 + 32) * 10 + 3) * 10 + 1: return 1;
        default: return 0;
    }
}

static int wgrad2d_mfma4_nx(int A, int B, int N, int Ho, int Wo, int ksize, int stride, int cap_parts)
{
    int ncg, toy;
    if (!wgrad2d_mfma4_cfg(A, B, ksize, stride, ncg, toy)) return 0;
    const int ntiles = ((Wo + WTX - 1) / WTX) * ((Ho + toy - 1) / toy) * N;
    return balanced_nx(ntiles, ((B + 3) / 4 + ncg - 1) / ncg, 4 / ncg, cap_parts);
}

int mvs_conv2d_wgrad_mfma4_parts(int A, int B, int N, int Ho, int Wo, int ksize, int stride, int cap_parts)
{
    int ncg, toy;
    if (!wgrad2d_mfma4_cfg(A, B, ksize, stride, ncg, toy)) return 0;
    return wgrad2d_mfma4_nx(A, B, N, Ho, Wo, ksize, stride, cap_parts) * (4 / ncg);
}

int mvs_conv2d_wgrad_mfma4(const float* g, int A, const ActSrc& x1, int B, int ldx, int N, int Ho, int Wo, int Hi, int Wi, int ksize, int stride,
                           float* partial, int cap_parts, hipStream_t st)
{
    const int nx = wgrad2d_mfma4_nx(A, B, N, Ho, Wo, ksize, stride, cap_parts);
    if (nx <= 0 || (ldx & 3)) return MVSNERF_EUNSUPPORTED;
    const ActSrc g1{g, nullptr, nullptr}, g2{nullptr, nullptr, nullptr};
#define MVS_W2(A_, NCG_, NQ_, S_, TOY_, K_) launch<A_, NCG_, NQ_, S_, 1, TOY_, 1, K_>(g1, g2, x1, ldx, B, N, Ho, Wo, N, Hi, Wi, partial, nx, st)
    const int key = ((A * 100 + B) * 10 + ksize) * 10 + stride;
    switch (key) {
        case ((8 * 100 + 3) * 10 + 3) * 10 + 1:   MVS_W2(8, 1, 4, 1, 16, 3); break;
        case ((8 * 100 + 8) * 10 + 3) * 10 + 1:   MVS_W2(8, 2, 2, 1, 16, 3); break;
        case ((16 * 100 + 8) * 10 + 5) * 10 + 2:  MVS_W2(16, 2, 2, 2, 8, 5); break;
        case ((16 * 100 + 16) * 10 + 3) * 10 + 1: MVS_W2(16, 4, 1, 1, 16, 3); break;
        case ((32 * 100 + 16) * 10 + 5) * 10 + 2: MVS_W2(32, 4, 1, 2, 8, 5); break;
        case ((32 * 100 + 32) * 10 + 3) * 10 + 1: MVS_W2(32, 4, 1, 1, 16, 3); break;
        default: return MVSNERF_EUNSUPPORTED;
    }
#undef MVS_W2
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}
