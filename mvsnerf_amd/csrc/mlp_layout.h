// Packed-weight layout of the fused Renderer_ours MLP kernel (shared by the pack kernel, the
// compute kernel and the host-side size query).
//
// The kernel computes every layer TRANSPOSED: out^T[n][m] = sum_k W[n][k] * act[k][m] with the weights
// as the MFMA A operand and 32 points (m) as the B operand of v_mfma_f32_32x32x2_f32.  The C/D
// fragment of that instruction puts, in lane (m = lane&31, half = lane>>5), the outputs
//     n(q, half) = (q>>4)*32 + (q&3) + 8*((q&15)>>2) + 4*half ,  q = 16*block + reg
// and the B operand of k-step t wants, in the same lane, act[k_t(half)][m].  Choosing the k-pairing of
// the NEXT layer as k_t(half) = n(t, half) makes the output registers of one layer the B operands of
// the next with no data movement: activations never leave the register file.  The weights are
// re-ordered once (mvsnerf_mlp_pack) so that the A fragment of k-steps 4j..4j+3 of block b is one
// float4 per lane, lane-linear in LDS (conflict-free ds_read_b128).
#pragma once
#include <stddef.h>

namespace mlp {

constexpr int WIDTH = 128;      // netwidth (reference default opt.py:38, shipped checkpoint)
constexpr int PE_DIM = 63;      // 3 + 3*2*10 (models.py:53-68, multires=10)
constexpr int PE_STEPS = 32;    // 64 padded inputs / 2
constexpr int ACT_STEPS = 64;   // 128 / 2
constexpr int VIEW_STEPS = 68;  // 64 (feature) + 2 (dir xyz + pad) rounded up to a multiple of 4
constexpr int MAX_F = 40;     // feat_dim = 8 + 4*V, V <= 8

// k-maps: which input column feeds (k-step t, lane half h); -1 = zero padding
enum KMap { K_PE = 0, K_FEAT = 1, K_ACT = 2, K_VIEWS = 3 };

__host__ __device__ inline int act_n(int q, int h) { return (q >> 4) * 32 + (q & 3) + 8 * ((q & 15) >> 2) + 4 * h; }

__host__ __device__ inline int kmap_col(int kmap, int t, int h, int F)
{
    switch (kmap) {
    case K_PE:      // t=0:(x,y) t=1:(z,0) t>=2: j=t-2 -> (sin_j, cos_j); embedding layout [xyz | sin(30) | cos(30)]
        if (t == 0) return h;
        if (t == 1) return h ? -1 : 2;
        return t < PE_STEPS ? 3 + (t - 2) + 30 * h : -1;
    case K_FEAT:    // half h owns feature columns [h*F/2, (h+1)*F/2)
        return t < F / 2 ? h * (F / 2) + t : -1;
    case K_ACT:
        return t < ACT_STEPS ? act_n(t, h) : -1;
    case K_VIEWS:   // [feature(128) | dir(3)]  (models.py:211)
        if (t < ACT_STEPS) return act_n(t, h);
        if (t == ACT_STEPS) return WIDTH + h;
        if (t == ACT_STEPS + 1) return h ? -1 : WIDTH + 2;
        return -1;
    }
    return -1;
}

__host__ __device__ inline int feat_steps(int F) { return ((F / 2) + 3) & ~3; }

// segment sizes in floats: steps * blocks * 64 lanes
__host__ __device__ constexpr inline size_t seg_floats(int steps, int nb) { return (size_t)steps * nb * 64; }

// Offsets (floats) of the weight segments, in the order the kernel streams them.
struct Layout {
    size_t biasw, l0, l1, l2, l3, l4, l5a, l5b, feat, views, vec, total;
    int fsteps;
};
// vector block (fragment-ordered biases and the two small heads), floats from `vec`
constexpr int V_BIASG = 0;              // [2][64] pts_bias bias
constexpr int V_L0 = 128;               // V_L0 + 128*i : pts_linears.i bias, i = 0..5
constexpr int V_FEAT = 128 * 7;         // feature_linear bias
constexpr int V_VIEWS = 128 * 8;        // [2][32] views_linears.0 bias
constexpr int V_WA = V_VIEWS + 64;      // [2][64] alpha_linear weight
constexpr int V_BA = V_WA + 128;        // alpha bias (+3 pad)
constexpr int V_WR = V_BA + 4;          // [3][2][32] rgb_linear weight
constexpr int V_BR = V_WR + 192;        // rgb bias (3, +1 pad)
constexpr int V_TOTAL = V_BR + 4;       // 1416

__host__ __device__ inline Layout layout(int F)
{
    Layout L;
    L.fsteps = feat_steps(F);
    size_t o = 0;
    L.biasw = o; o += seg_floats(L.fsteps, 4);
    L.l0 = o;    o += seg_floats(PE_STEPS, 4);
    L.l1 = o;    o += seg_floats(ACT_STEPS, 4);
    L.l2 = o;    o += seg_floats(ACT_STEPS, 4);
    L.l3 = o;    o += seg_floats(ACT_STEPS, 4);
    L.l4 = o;    o += seg_floats(ACT_STEPS, 4);
    L.l5a = o;   o += seg_floats(PE_STEPS, 4);
    L.l5b = o;   o += seg_floats(ACT_STEPS, 4);
    L.feat = o;  o += seg_floats(ACT_STEPS, 4);
    L.views = o; o += seg_floats(VIEW_STEPS, 2);
    L.vec = o;   o += V_TOTAL;
    L.total = o;
    return L;
}

// ------------------------------------------------------------------------------------------------
// Training: activations saved by the forward pass / gradients written by the dgrad kernel, in "slot" format.
// A tile = the 32 points of one wave.  A slot = 64 floats = one register of the wave = two rows (lane half 0
// and 1) of 32 points each: address = ((tile * SLOTS + slot) * 64 + lane).  A row is the 32-point vector of one
// feature, which is exactly the 128-byte operand row the weight-gradient MFMAs contract over.
constexpr int S_E = 0;            // 32 slots: positional-encoding operands, slot t = (kmap PE column (t,0), (t,1))
constexpr int S_FV = 32;          // 16 slots: feature operands, slot t = feat columns (t, F/2+t)           (F <= 32)
constexpr int S_BM = 48;          // 64 slots: pts_bias output b, slot q = features n(q,0), n(q,1)
constexpr int S_H = 112;          // 6 x 64 slots: h_0..h_5 (post-ReLU)
constexpr int S_FE = 496;         // 64 slots: feature_linear output
constexpr int S_HV = 560;         // 32 slots: relu(views_linears[0])
constexpr int S_DR = 592;         // 16 slots: slot 0 = (d0,d1), slot 1 = (d2,0); rest unused
constexpr int SLOTS_SAVED = 608;
// written by the dgrad kernel
constexpr int G_GP = 0;           // 6 x 64 slots: grad wrt the pre-activation of pts_linears[i] (already x bias)
constexpr int G_GBM = 384;        // 64 slots: grad wrt pts_bias output
constexpr int G_GF = 448;         // 64 slots: grad wrt feature_linear output
constexpr int G_GPV = 512;        // 32 slots: grad wrt views_linears[0] pre-activation
constexpr int G_G4 = 544;         // 16 slots: slot 0 = (d rgb_r pre-sigmoid, d rgb_g), slot 1 = (d rgb_b, d sigma pre-relu)
constexpr int SLOTS_GRAD = 560;

// Transposed (dgrad) weight segments: A operand = W^T, i.e. fragment(t, kb, lane(i,h)) = W[n(t,h)][col_off + kb*32 + i]
struct LayoutBwd {
    size_t views, feat, l5, l4, l3, l2, l1, bias, total;   // views: 32 steps x 4 blocks; feat, l1..l5: 64 x 4; bias: 64 x 1
};
__host__ __device__ inline LayoutBwd layout_bwd()
{
    LayoutBwd L;
    size_t o = 0;
    L.views = o; o += seg_floats(32, 4);
    L.feat = o;  o += seg_floats(ACT_STEPS, 4);
    L.l5 = o;    o += seg_floats(ACT_STEPS, 4);
    L.l4 = o;    o += seg_floats(ACT_STEPS, 4);
    L.l3 = o;    o += seg_floats(ACT_STEPS, 4);
    L.l2 = o;    o += seg_floats(ACT_STEPS, 4);
    L.l1 = o;    o += seg_floats(ACT_STEPS, 4);
    L.bias = o;  o += seg_floats(ACT_STEPS, 1);
    L.total = o;
    return L;
}

}  // namespace mlp
