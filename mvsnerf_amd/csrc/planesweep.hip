// Plane-sweep cost volume, forward (utils.py:580-630 homo_warp + models.py:787-893 build_volume_costvar[_img]): the kernels that were part of
// encoder.hip until round 4, in their own translation unit because this file is compiled with -fno-slp-vectorize (Makefile), i.e. WITHOUT
// packed fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32; `make` checks the code object for them).
//
// Why (scratch/r4/race_probe2.py, profiles/r04_pk_mfma_hazard.txt): with the SLP vectoriser's packed fp32 arithmetic this kernel computed wrong
// variances - ONE accumulator register, lanes 48..63 of a wave, plausible but different values - whenever waves issuing 16-bit matrix
// instructions (v_mfma_f32_16x16x32_{f16,bf16}: the fp16x3 / bf16 conv0 kernels) were resident on the same SIMD: another process sharing the
// GPU (tests/test_gpu_shared.py: 10-35 % of the scene encodes differed), or a second stream of the same process (192 of 192 sweeps).  Stream
// order keeps the two kernels apart inside one stream, which is why single-process runs never showed it.  The aggressor is the MFMA stream (the
// conv0 kernel without its LDS-DMA still disturbs, without its MFMAs it does not; the fp32 4x4x1 MFMA conv0 never did), the victim the packed
// fp32 VALU arithmetic: the same source compiled without it is bit-identical in its results, never differed again (0 of 1152 encodes with
// three processes, 0 of 192 with two streams) and is 3 % FASTER (0.312 vs 0.322 ms: 10 % more VALU instructions, fewer register moves).
// Nothing in the ISA explains it (waits and hazards nops are the compiler's and are correct for in-order return): treated as a hardware
// interaction to stay away from, see DESIGN.md section 9.
#include "common.h"
#include "act.h"
#include "knobs.h"

// =============================================================================================
// plane sweep: homo_warp (utils.py:580-630) + build_volume_costvar[_img] (models.py:787-893) in ONE pass.
// planesweep_kernel: ONE WAVE per workgroup = 16 voxel columns x 4 consecutive depth planes, in two phases:
//  (1) one lane per (plane, column): projection into every source view (4 divisions per view; T / depth is per plane), in-frustum mask,
//      bilinear weights and tap indices -> LDS; the per-view masks and the view count are written from here;
//  (2) FOUR lanes per column, lane q owning the channel quads {q, q+4} of the 32-channel feature vector, walk the column's 4 planes:
//      a wave-level 16-byte load covers 16 voxels x 64 CONTIGUOUS bytes of a source pixel and is repeated for the next plane only if a
//      tap address changed; Sigma / Sigma^2 / variance; the 16 finished voxel vectors of a plane are staged in LDS and flushed as one
//      contiguous span of coalesced 16-byte stores.
// History (config 2, 4.69 M voxels): one lane per voxel (8 x 16-byte loads per tap, every load instruction touched 64 different lines)
// 0.57 ms; four lanes per voxel each repeating phase 1: 0.39 ms; phase 1 once per voxel through LDS, 256 consecutive voxels of one plane
// per 256-thread workgroup: 0.33 ms (planesweep_blocks_kernel, dev build).  What bound THAT was not what it looked like: XCD banding cut
// the fetch from 1.27 GB to 25 MB (-8 % time), keeping unchanged taps in registers cut the gathers by 3x (-0 %), 32-bit index arithmetic
// and hoisted divisions (-0 %), and an ablation run (scratch/r3/psw_dbg.py) showed the skeleton - barriers, staging, flush with every
// store, gather and blend removed - still took 0.17 ms: a chain of barrier-separated phases with 3-4 workgroups per CU.  Single-wave
// workgroups have no barriers to wait at and interleave at instruction granularity: 0.29 ms.  (A column form with one workgroup walking
// all planes of 32 columns, taps in registers - the backward kernel's design - was bit-identical and SLOWER, 0.40 ms: eight columns per
// wave change their tap set at different planes, so every plane stalled some lane group of every wave on a gather;
// scratch/r3/psw_fwd_columns_dropped.hip.txt.)  Writes the voxel's CP-channel vector once (the reference moves ~10 GB for the same result).
// =============================================================================================

template <int C, int NP>   // feature channels (32), depth planes per wave; `bid`: the workgroup (= wave) index
__device__ __forceinline__ void planesweep_tile(
    const unsigned bid,
    const float* __restrict__ feat,   // [V][H][W][C]
    const float* __restrict__ img,    // [V][H][W][4] or null
    const float* __restrict__ proj,   // [V][3][4]
    const float* __restrict__ depth,  // [D]
    int V, int H, int W, int D, int pad,
    float* __restrict__ cost, int CP,   // [D][Hp][Wp][CP]
    float* __restrict__ masks,          // with img: [V][D][Hp][Wp] per-view; else [D][Hp][Wp] count
    int with_img, int blocked,          // blocked 1: cost[CP/4][D*Hp*Wp][4] (channel blocks of four, see mvsnerf_planesweep_costvar_blocked_fwd);
                                        // 2: bf16 in channel blocks of sixteen, cost16[ceil(CP/16)][D*Hp*Wp][16] (mvsnerf_planesweep_costvar_bf16_fwd)
                                        // 3: two fp16 pieces of x / 16 in that layout (mvsnerf_planesweep_costvar_f16x2_fwd)
    int* __restrict__ guard)            // blocked 3 in a guarded sequence (include/mvsnerf_hip.h): guard[0] = 1 when a value did not fit an fp16 piece
{
    // fp32 arithmetic of the CPU reference path, operation for operation (scratch/keep/cpu_arith_probe.py, cpu_var_probe.py compare candidate
    // formulas with reference-generated fixtures BIT FOR BIT): the projection is a k-ordered fma chain (sgemm), grid_sample's blend is
    // fma(se, w_se, fma(sw, w_sw, fma(ne, w_ne, nw * w_nw))), and everything in models.py:879-890 is one ATen op per rounding
    // (x**2, +, *count, -): no contraction anywhere else.
#pragma clang fp contract(off)
    static_assert(C == 32, "lane q owns float4 numbers q and q + 4 of a 32-channel pixel");
    constexpr int NC = 64 / NP, VPB = 64;                        // columns x depth planes of a workgroup = one wave
    extern __shared__ __attribute__((aligned(16))) float lds_[];
    // per voxel: per source view {w_nw,w_ne,w_sw,w_se, t_nw,t_ne,t_sw,t_se}; then {1/count, ref pixel}.  Row strides in floats with
    // stride / 4 ODD: consecutive rows then start on different 16-byte bank groups (16 rows cover all 64 banks once), so that the 16-byte
    // reads of 16 voxels' rows (phase 2, flush) do not collide.  (Round 2 had 18 and CP + 4 = 48: PMC SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    // = 0.61; 48 floats put rows v and v + 4 on the same banks for the flush's ds_read_b128 and all rows on two bank phases for the stores.)
    const int GS = mvs_odd_quad_stride((V - 1) * 8 + 2);
    float* geo = lds_;                                           // [NP][NC][GS]
    const int RS = mvs_odd_quad_stride((blocked >= 2 ? ((CP + 15) & ~15) : CP) + 1);   // staging row (bf16 mode stages whole 16-channel blocks)
    float* stage = lds_ + ((VPB * GS + 3) & ~3);                 // [16][RS]: the 16 columns of a group
    const int TS = 3 * V + 1;                                    // warped thumbnails of a voxel, written and read back by the same lane
    float* thumbs = stage + 16 * RS;                             // [NP][NC][TS] (with_img only)
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    const int64_t nvox = (int64_t)D * Hp * Wp;
    // Workgroup (= wave) -> voxels: NC = 16 consecutive voxels of a (row band) slab x NP = 4 consecutive depth planes.
    //  * Bands.  Workgroups are dealt to the 8 XCDs round-robin and every XCD has its own 4 MB L2; the source feature maps are V x 2.6 MB.
    //    With workgroup b taking voxels 256 b .. 256 b + 255 every XCD swept every row of every plane and its L2 kept missing: PMC FETCH_SIZE
    //    1.27 GB per launch for 8.6 MB of input.  XCD k owns the band of rows [k RB, (k+1) RB) of every plane: its taps fall into ~RB + 2 rows
    //    of each source view (1.5 MB at config 2), which stay in its L2 for the whole launch (24.5 MB fetched).
    //  * Planes.  Along a voxel column the sample point in a source view moves by the disparity step per plane - a fraction of a pixel for
    //    any rig a sweep is meant for (0.07 px at config 2) - so the four tap PIXELS of consecutive planes are usually the same and only the
    //    bilinear weights change.  A lane walks its column through the 4 planes with the 8 tap vectors in registers and gathers again only
    //    when a tap address changes: ~1.2 gathers per column, view and 4 planes instead of 4 (1 KB gathered per voxel before, 4.8 GB through
    //    the L1s per launch).  A rig whose taps move every plane gathers as often as before.
    const int RB = (Hp + 7) >> 3;                                // rows per band
    const int CPS = (RB * Wp + NC - 1) / NC;                     // chunks per band slab
    const int xcd = bid & 7, jb = bid >> 3;
    const int dg = jb / CPS, chunk = jb - dg * CPS;
    const int band_rows = min(RB, Hp - xcd * RB);                // the last band may be short (or empty)
    const int slab = band_rows > 0 ? band_rows * Wp : 0;
    const int n_col = min(NC, slab - chunk * NC);                // columns of this workgroup (<= 0: nothing to do)
    if (n_col <= 0) return;
    const int d0 = dg * NP, np = min(NP, D - d0);                // planes of this workgroup
    const int64_t plane = (int64_t)Hp * Wp;
    const int64_t base0 = (int64_t)d0 * plane + (int64_t)xcd * RB * Wp + (int64_t)chunk * NC;   // voxel (plane d0, column 0); plane p: + p * plane
    // T / depth of utils.py:612 depends on (plane, view) only: NP x (V-1) x 3 divisions per workgroup instead of 3 per voxel and view
    float* tdv = thumbs + (with_img ? VPB * TS : 0);             // [NP][V-1][3]
    for (int t = threadIdx.x; t < np * (V - 1) * 3; t += VPB) {
        const int pl = t / ((V - 1) * 3), r = t - pl * (V - 1) * 3, vs = r / 3, k = r - vs * 3;
        tdv[t] = proj[(vs + 1) * 12 + 4 * k + 3] / depth[d0 + pl];
    }
    __syncthreads();
    {   // ---- phase 1: lane -> (plane tid / 16, column tid % 16)
        const int pl = threadIdx.x / NC, col = threadIdx.x % NC;
        if (col < n_col && pl < np) {
            const unsigned r = (unsigned)(xcd * RB * Wp + chunk * NC + col);   // voxel index inside its plane (32-bit: no 64-bit divisions)
            const int y = (int)(r / (unsigned)Wp), x = (int)(r - (unsigned)y * (unsigned)Wp);
            const int64_t i = base0 + pl * plane + col;
            float* o = geo + threadIdx.x * GS;
            const float u = (float)(x - pad), v = (float)(y - pad);     // utils.py:603-605
            const float* td = tdv + pl * (V - 1) * 3;
            const bool interior = x >= pad && x < W + pad && y >= pad && y < H + pad;
            if (with_img) masks[i] = 1.0f;                          // view 0 mask (models.py:869)
            float cnt = 1.0f;
            for (int vv = 1; vv < V; ++vv) {
                const float* P = proj + vv * 12;
                // utils.py:612  R @ (u,v,1) + T/depth   (k-ordered fma chain like the reference's bmm)
                const float p0 = fmaf(P[2], 1.0f, fmaf(P[1], v, P[0] * u)) + td[(vv - 1) * 3];
                const float p1 = fmaf(P[6], 1.0f, fmaf(P[5], v, P[4] * u)) + td[(vv - 1) * 3 + 1];
                const float p2 = fmaf(P[10], 1.0f, fmaf(P[9], v, P[8] * u)) + td[(vv - 1) * 3 + 2];
                const float gx = (p0 / p2) / ((float)(W - 1) / 2.0f) - 1.0f;          // :617-620 (un-padded W,H)
                const float gy = (p1 / p2) / ((float)(H - 1) / 2.0f) - 1.0f;
                const float m = (gx > -1.0f && gx < 1.0f && gy > -1.0f && gy < 1.0f) ? 1.0f : 0.0f;   // models.py:875-876
                cnt += m;
                if (with_img) masks[(int64_t)vv * nvox + i] = m;
                // F.grid_sample bilinear, zeros padding, align_corners=True (utils.py:625)
                const float ix = ((gx + 1.0f) / 2.0f) * (float)(W - 1);
                const float iy = ((gy + 1.0f) / 2.0f) * (float)(H - 1);
                const float fx = floorf(ix), fy = floorf(iy);
                const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
                const bool x0in = fx >= 0.f && fx <= (float)(W - 1), x1in = fx + 1.f >= 0.f && fx + 1.f <= (float)(W - 1);
                const bool y0in = fy >= 0.f && fy <= (float)(H - 1), y1in = fy + 1.f >= 0.f && fy + 1.f <= (float)(H - 1);
                float* ov = o + (vv - 1) * 8;
                // two 16-byte stores per view (rows of GS floats, GS / 4 odd: eight lanes' 16-byte stores cover all banks once; the same values as
                // eighteen 4-byte stores at a row stride of 80 bytes landed on 8 of the 32 banks - PMC r4 / r5: LDS bank conflicts 0.335 of the active cycles)
                f32x4 wv;
                wv[0] = (x0in && y0in) ? wx0 * wy0 : 0.f; wv[1] = (x1in && y0in) ? wx1 * wy0 : 0.f;
                wv[2] = (x0in && y1in) ? wx0 * wy1 : 0.f; wv[3] = (x1in && y1in) ? wx1 * wy1 : 0.f;
                // clamp the tap addresses (weights are already zero where a tap is outside)
                const bool any = (x0in || x1in) && (y0in || y1in);
                const int xa = any ? min(max((int)fx, 0), W - 1) : 0, xb = any ? min(max((int)fx + 1, 0), W - 1) : 0;
                const int ya = any ? min(max((int)fy, 0), H - 1) : 0, yb = any ? min(max((int)fy + 1, 0), H - 1) : 0;
                f32x4 av;
                av[0] = __int_as_float(ya * W + xa); av[1] = __int_as_float(ya * W + xb);
                av[2] = __int_as_float(yb * W + xa); av[3] = __int_as_float(yb * W + xb);
                *reinterpret_cast<f32x4*>(ov) = wv;
                *reinterpret_cast<f32x4*>(ov + 4) = av;
            }
            if (!with_img) masks[i] = cnt;                          // build_volume_costvar returns the count (models.py:821)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 tail;
            tail[0] = 1.0f / cnt;                                   // models.py:889
            tail[1] = __int_as_float(interior ? (y - pad) * W + (x - pad) : -1);
            *reinterpret_cast<f32x2*>(o + (V - 1) * 8) = tail;
        }
    }
    __syncthreads();
    // ---- phase 2: four lanes per column, lane q owning channels 4q..4q+3 and 16+4q..16+4q+3, all np planes of the column
    const int q = threadIdx.x & 3;
    const int c_var = with_img ? 3 * V : 0;
    for (int cg = 0; cg < NC / 16; ++cg) {                       // 16 columns at a time (NP < 4: the wave's columns in NC / 16 groups)
    if (cg * 16 >= n_col) break;
    const int vrow = threadIdx.x >> 2, vloc = cg * 16 + vrow;
    const int colv = vloc < n_col ? vloc : 0;                    // dead lanes recompute column 0 (their rows are not flushed)
    const float* g0 = geo + colv * GS;                           // plane p: + p * NC * GS
    const int refpix = __float_as_int(g0[(V - 1) * 8 + 1]);      // the column's own pixel in the reference view
    const bool interior = refpix >= 0;
    float s[NP][8], s2[NP][8];
    {                                                            // ref volume: zero-padded ref feature (models.py:856,862)
        f32x4 t0 = {0.f, 0.f, 0.f, 0.f}, t1 = t0;
        if (interior) {
            const f32x4* r = reinterpret_cast<const f32x4*>(feat + (int64_t)refpix * C);
            t0 = r[q]; t1 = r[q + 4];
        }
#pragma unroll
        for (int p = 0; p < NP; ++p)
#pragma unroll
            for (int k = 0; k < 4; ++k) { s[p][k] = t0[k]; s2[p][k] = t0[k] * t0[k]; s[p][4 + k] = t1[k]; s2[p][4 + k] = t1[k] * t1[k]; }
    }
    for (int vv = 1; vv < V; ++vv) {
        const float* fb = feat + (int64_t)vv * H * W * C;
        const bool mine = with_img && q == (vv & 3);             // warped thumbnail with the same grid (models.py:872), one lane per view
        int a_nw = -1, a_ne = -1, a_sw = -1, a_se = -1;
        f32x4 tp[4][2], tt[4];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p < np) {
                const float* gv = g0 + p * NC * GS + (vv - 1) * 8;
                const f32x4 wq = *reinterpret_cast<const f32x4*>(gv), aq = *reinterpret_cast<const f32x4*>(gv + 4);
                const float w_nw = wq[0], w_ne = wq[1], w_sw = wq[2], w_se = wq[3];
                const int b_nw = __float_as_int(aq[0]), b_ne = __float_as_int(aq[1]), b_sw = __float_as_int(aq[2]), b_se = __float_as_int(aq[3]);
                if (p == 0 || ((b_nw != a_nw) | (b_ne != a_ne) | (b_sw != a_sw) | (b_se != a_se))) {   // new tap set: gather
                    a_nw = b_nw; a_ne = b_ne; a_sw = b_sw; a_se = b_se;
                    const f32x4* t_nw = reinterpret_cast<const f32x4*>(fb + (int64_t)a_nw * C);
                    const f32x4* t_ne = reinterpret_cast<const f32x4*>(fb + (int64_t)a_ne * C);
                    const f32x4* t_sw = reinterpret_cast<const f32x4*>(fb + (int64_t)a_sw * C);
                    const f32x4* t_se = reinterpret_cast<const f32x4*>(fb + (int64_t)a_se * C);
                    tp[0][0] = t_nw[q]; tp[0][1] = t_nw[q + 4]; tp[1][0] = t_ne[q]; tp[1][1] = t_ne[q + 4];
                    tp[2][0] = t_sw[q]; tp[2][1] = t_sw[q + 4]; tp[3][0] = t_se[q]; tp[3][1] = t_se[q + 4];
                    if (mine) {
                        const float* ib = img + (int64_t)vv * H * W * 4;
                        tt[0] = *reinterpret_cast<const f32x4*>(ib + (int64_t)a_nw * 4); tt[1] = *reinterpret_cast<const f32x4*>(ib + (int64_t)a_ne * 4);
                        tt[2] = *reinterpret_cast<const f32x4*>(ib + (int64_t)a_sw * 4); tt[3] = *reinterpret_cast<const f32x4*>(ib + (int64_t)a_se * 4);
                    }
                }
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float wv = fmaf(tp[3][hh][k], w_se, fmaf(tp[2][hh][k], w_sw, fmaf(tp[1][hh][k], w_ne, tp[0][hh][k] * w_nw)));   // ATen's nw,ne,sw,se chain
                        s[p][hh * 4 + k] += wv;                      // models.py:880
                        s2[p][hh * 4 + k] += wv * wv;                // :881 (the square is rounded before it is added)
                    }
                if (mine) {
                    float* th = thumbs + (p * NC + vloc) * TS + 3 * vv;
#pragma unroll
                    for (int k = 0; k < 3; ++k) th[k] = fmaf(tt[3][k], w_se, fmaf(tt[2][k], w_sw, fmaf(tt[1][k], w_ne, tt[0][k] * w_nw)));
                }
            }
        }
    }
    float* o = stage + vrow * RS;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p >= np) break;                                      // (uniform)
        const float inv = g0[p * NC * GS + (V - 1) * 8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = (j < 4 ? 4 * q : 16 + 4 * q - 4) + j;
            const float mean = s[p][j] * inv;
            o[c_var + c] = s2[p][j] * inv - mean * mean;             // :890
        }
        if (with_img) {
            if (q == 0) {                                        // channels 0:3 = ref thumbnail, border := 0 (models.py:858-860)
                const float* ri = img + (int64_t)(interior ? refpix : 0) * 4;
                o[0] = interior ? ri[0] : 0.f; o[1] = interior ? ri[1] : 0.f; o[2] = interior ? ri[2] : 0.f;
            }
            const float* th = thumbs + (p * NC + vloc) * TS;
            for (int vv = 1; vv < V; ++vv)
                if (q == (vv & 3)) { o[3 * vv] = th[3 * vv]; o[3 * vv + 1] = th[3 * vv + 1]; o[3 * vv + 2] = th[3 * vv + 2]; }
        }
        if (q == 0)
            for (int c = c_var + C; c < (blocked >= 2 ? ((CP + 15) & ~15) : CP); ++c) o[c] = 0.0f;
        // flush this plane's n_col consecutive voxels (one contiguous span of the cost volume) with coalesced 16-byte stores
        __syncthreads();
        {
            const int64_t p0 = base0 + p * plane + cg * 16;
            const int nv = min(16, n_col - cg * 16);
            const int cpq = CP >> 2, n4 = nv * cpq;
            const float inv_cpq = 1.0f / (float)cpq;
            if (!blocked) {
                f32x4* dst = reinterpret_cast<f32x4*>(cost + p0 * CP);
                for (int k = threadIdx.x; k < n4; k += VPB) {
                    const int vox = (int)(((float)k + 0.5f) * inv_cpq), c = (k - vox * cpq) * 4;   // k / cpq, exact for these sizes
                    dst[k] = *reinterpret_cast<const f32x4*>(stage + vox * RS + c);
                }
            } else if (blocked == 2) {
                // bf16 (round to nearest even), blocks of sixteen channels: the two 16-byte halves of a voxel's block, voxels consecutive
                typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
                __bf16* cost16 = reinterpret_cast<__bf16*>(cost);
                const int nb16 = (CP + 15) >> 4, per = nv * 2, n_k = per * nb16;
                const float inv_per = 1.0f / (float)per;
                for (int k = threadIdx.x; k < n_k; k += VPB) {
                    const int cb = (int)(((float)k + 0.5f) * inv_per), rem = k - cb * per, vox = rem >> 1, c0 = cb * 16 + (rem & 1) * 8;      // k / per, exact for these sizes
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + vox * RS + c0), hi = *reinterpret_cast<const f32x4*>(stage + vox * RS + c0 + 4);
                    bf16x8_t h;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { h[e] = (__bf16)lo[e]; h[4 + e] = (__bf16)hi[e]; }
                    *reinterpret_cast<bf16x8_t*>(cost16 + (((int64_t)cb * nvox + p0 + vox) << 4) + (rem & 1) * 8) = h;
                }
            } else if (blocked == 3) {
                // two fp16 pieces of x * 2^-4 (conv_f16x3.hip): hi = fp16(x'), lo = fp16(x' - hi), both round to nearest, in the bf16 mode's blocks
                // of sixteen channels; the lo plane follows the hi plane
                typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
                _Float16* cost16 = reinterpret_cast<_Float16*>(cost);
                const int nb16 = (CP + 15) >> 4, per = nv * 2, n_k = per * nb16;
                const int64_t lo_plane = (int64_t)nb16 * nvox * 16;
                const float inv_per = 1.0f / (float)per;                // (an integer division per 16-byte chunk was 6 % of this kernel's VALU instructions)
                for (int k = threadIdx.x; k < n_k; k += VPB) {
                    const int cb = (int)(((float)k + 0.5f) * inv_per), rem = k - cb * per, vox = rem >> 1, c0 = cb * 16 + (rem & 1) * 8;      // k / per, exact for these sizes (k < 96, per <= 32)
                    const f32x4 lo = *reinterpret_cast<const f32x4*>(stage + vox * RS + c0), hi = *reinterpret_cast<const f32x4*>(stage + vox * RS + c0 + 4);
                    // (round 6: the same two roundings in three instructions per PAIR - v_cvt_pk_f16_f32, then v_fma_mix{lo,hi}_f16 form fp16(v - hi) straight
                    // from the packed hi pieces - instead of six; the flush was a quarter of this VALU-bound kernel's instructions)
                    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                    u32x4_t h0, h1;
                    float big = 0.0f;
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const float x0 = (e2 < 2 ? lo[2 * e2] : hi[2 * e2 - 4]) * 0.0625f, x1 = (e2 < 2 ? lo[2 * e2 + 1] : hi[2 * e2 - 3]) * 0.0625f;
                        big = fmaxf(big, fmaxf(fabsf(x0), fabsf(x1)));
                        const float v0 = __builtin_amdgcn_fmed3f(x0, -65504.0f, 65504.0f), v1 = __builtin_amdgcn_fmed3f(x1, -65504.0f, 65504.0f);
                        unsigned hp, lp;
                        asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_fma_mixlo_f16 %1, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                            : "=&v"(hp), "=&v"(lp) : "v"(v0), "v"(v1));
                        h0[e2] = hp; h1[e2] = lp;
                    }
                    if (guard && big > 65504.0f) guard[0] = 1;          // saturated: the fp32 sweep + conv0 behind this launch take over
                    const int64_t at = (((int64_t)cb * nvox + p0 + vox) << 4) + (rem & 1) * 8;
                    *reinterpret_cast<u32x4_t*>(cost16 + at) = h0;
                    *reinterpret_cast<u32x4_t*>(cost16 + lo_plane + at) = h1;
                }
            } else {
                // channel block cb (four channels) of these nv voxels is one contiguous run of nv * 16 bytes: k -> (cb, voxel)
                const int nblk = CP >> 2, n_k = nv * nblk;
                const float inv_nv = 1.0f / (float)nv;
                for (int k = threadIdx.x; k < n_k; k += VPB) {
                    const int cb = (int)(((float)k + 0.5f) * inv_nv), vox = k - cb * nv;      // k / nv, exact for these sizes (k < 16 * 16)
                    *reinterpret_cast<f32x4*>(cost + (((int64_t)cb * nvox + p0 + vox) << 2)) =
                        *reinterpret_cast<const f32x4*>(stage + vox * RS + cb * 4);
                }
            }
        }
        __syncthreads();                                         // the staging rows are rewritten by the next plane
    }
    }
}

template <int C, int NP>
__global__ __launch_bounds__(64) void planesweep_kernel(
    const float* __restrict__ feat, const float* __restrict__ img, const float* __restrict__ proj, const float* __restrict__ depth,
    int V, int H, int W, int D, int pad, float* __restrict__ cost, int CP, float* __restrict__ masks, int with_img, int blocked, int* __restrict__ guard)
{
    planesweep_tile<C, NP>(blockIdx.x, feat, img, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, guard);
}

// fp32 half of a guarded sequence (mvsnerf_sweep_conv0_guarded_fwd): a persistent grid that walks the `n_wg` workgroup indices of the plain
// launch only when *run_if != 0 - when the guard is clear 2048 waves leave at once instead of 73 k (16 us of every encode).
template <int C, int NP>
__global__ __launch_bounds__(64) void planesweep_if_kernel(
    const float* __restrict__ feat, const float* __restrict__ img, const float* __restrict__ proj, const float* __restrict__ depth,
    int V, int H, int W, int D, int pad, float* __restrict__ cost, int CP, float* __restrict__ masks, int with_img, int blocked,
    unsigned n_wg, const int* __restrict__ run_if)
{
    if (*run_if == 0) return;
    for (unsigned bid = blockIdx.x; bid < n_wg; bid += gridDim.x) {
        planesweep_tile<C, NP>(bid, feat, img, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, nullptr);
        __syncthreads();
    }
}

int mvs_planesweep_launch(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                          int V, int C, int H, int W, int D, int pad, float* cost, int CP, float* masks,
                          int with_img, int blocked, void* stream, int* guard = nullptr, const int* run_if = nullptr);

extern "C" int mvsnerf_planesweep_costvar_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                              int V, int C, int H, int W, int D, int pad, float* cost, int CP, float* masks,
                                              int with_img, void* stream)
{
    return mvs_planesweep_launch(feats_cl, imgs_cl, proj, depth, V, C, H, W, D, pad, cost, CP, masks, with_img, 0, stream);
}

extern "C" int mvsnerf_planesweep_costvar_blocked_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                                      int V, int C, int H, int W, int D, int pad, float* cost_blocked, int CP, float* masks,
                                                      int with_img, void* stream)
{
    return mvs_planesweep_launch(feats_cl, imgs_cl, proj, depth, V, C, H, W, D, pad, cost_blocked, CP, masks, with_img, 1, stream);
}

// The cost volume rounded to bf16 in channel blocks of sixteen: cost16[ceil(CP/16)][D*Hp*Wp][16] (channels >= CP are zero) - what the bf16
// conv0 kernels (conv_bf16.hip) stage with 1 KB DMA pieces.  The sweep's own arithmetic is the fp32 one; only the store rounds.
extern "C" int mvsnerf_planesweep_costvar_bf16_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                                   int V, int C, int H, int W, int D, int pad, void* cost16, int CP, float* masks,
                                                   int with_img, void* stream)
{
    return mvs_planesweep_launch(feats_cl, imgs_cl, proj, depth, V, C, H, W, D, pad, reinterpret_cast<float*>(cost16), CP, masks, with_img, 2, stream);
}

// The cost volume as two fp16 pieces of x * 2^-4, each in the bf16 mode's layout: cost16[2][ceil(CP/16)][D*Hp*Wp][16] (hi plane, then lo plane; channels
// >= CP are zero) - the operand of the fp32-grade fp16 conv0 (conv_f16x3.hip).  The sweep's own arithmetic is the fp32 one; only the store splits.
extern "C" int mvsnerf_planesweep_costvar_f16x2_fwd(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                                                    int V, int C, int H, int W, int D, int pad, void* cost16x2, int CP, float* masks,
                                                    int with_img, void* stream)
{
    return mvs_planesweep_launch(feats_cl, imgs_cl, proj, depth, V, C, H, W, D, pad, reinterpret_cast<float*>(cost16x2), CP, masks, with_img, 3, stream);
}


int mvs_planesweep_launch(const float* feats_cl, const float* imgs_cl, const float* proj, const float* depth,
                          int V, int C, int H, int W, int D, int pad, float* cost, int CP, float* masks,
                          int with_img, int blocked, void* stream, int* guard, const int* run_if)
{
    if (!feats_cl || !proj || !depth || !cost || !masks || V < 1 || H < 2 || W < 2 || D < 1 || pad < 0) return MVSNERF_EINVAL;
    if (with_img && !imgs_cl) return MVSNERF_EINVAL;
    if (C != 32) return MVSNERF_EUNSUPPORTED;
    if (CP < (with_img ? 3 * V : 0) + C) return MVSNERF_EINVAL;
    if (!mvs_aligned16(feats_cl) || (imgs_cl && !mvs_aligned16(imgs_cl))) return MVSNERF_EALIGN;
    const int64_t nvox = (int64_t)D * (H + 2 * pad) * (W + 2 * pad);
    if ((CP & 3) || !mvs_aligned16(cost)) return MVSNERF_EALIGN;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad, RB = (Hp + 7) >> 3;
    const size_t lds_geo = (((size_t)256 * mvs_odd_quad_stride((V - 1) * 8 + 2) + 3) & ~(size_t)3);
    const size_t lds_stage = (size_t)64 * mvs_odd_quad_stride((blocked >= 2 ? ((CP + 15) & ~15) : CP) + 1);
    const size_t lds = ((((size_t)64 * mvs_odd_quad_stride((V - 1) * 8 + 2) + 3) & ~(size_t)3) +
                        (size_t)16 * mvs_odd_quad_stride((blocked >= 2 ? ((CP + 15) & ~15) : CP) + 1) + (with_img ? (size_t)64 * (3 * V + 1) : 0) + (size_t)4 * (V > 1 ? V - 1 : 1) * 3) * sizeof(float);   // geo | stage | thumbs | tdv[NP = 4][V-1][3]
    static unsigned long long cap_mask = 0;
    if (lds > 48 * 1024) {      // many source views: raise the dynamic-LDS cap (idempotent, per device)
        const int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(planesweep_kernel<32, 4>), (int)lds, &cap_mask);
        if (rc != MVSNERF_OK) return rc;
    }
    const int CPS = (RB * Wp + 15) / 16;
    const unsigned n_wg = (unsigned)(8 * ((D + 3) / 4) * CPS);
    if (run_if) {
        static unsigned long long cap_if = 0;
        if (lds > 48 * 1024)
            if (const int rc = mvs_raise_lds_cap(reinterpret_cast<const void*>(planesweep_if_kernel<32, 4>), (int)lds, &cap_if)) return rc;
        planesweep_if_kernel<32, 4><<<n_wg < 4096u ? n_wg : 4096u, 64, lds, (hipStream_t)stream>>>(feats_cl, imgs_cl, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, n_wg, run_if);
        MVS_LAUNCH_CHECK();
        return MVSNERF_OK;
    }
    planesweep_kernel<32, 4><<<n_wg, 64, lds, (hipStream_t)stream>>>(feats_cl, imgs_cl, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, guard);
    MVS_LAUNCH_CHECK();
    return MVSNERF_OK;
}

