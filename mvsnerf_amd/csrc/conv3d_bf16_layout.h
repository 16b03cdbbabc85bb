// Fragment layout of the bf16 weight buffers of conv3d_bf16.hip, shared with the multi-job weight pack of encoder.hip (kinds 3 / 4).
#pragma once
#include "common.h"

inline bool mvs_conv3d_bf16_shape_ok(int Cin, int Cout) { return (Cin == 4 || Cin == 8 || Cin == 16 || Cin == 32 || Cin == 64) && (Cout == 8 || Cout == 16 || Cout == 32 || Cout == 64); }

// ntaps: 27 for the 3-D layers; k * k (1, 9, 25) for FeatureNet's 2-D layers (never transposed)
inline size_t mvs_conv3d_bf16_elems(int Cin, int Cout, int transposed, int ntaps = 27)
{
    if (!mvs_conv3d_bf16_shape_ok(Cin, Cout) || ntaps < 1 || ntaps > 27) return 0;
    if (transposed) {                                             // four (pz, py) classes, both x parities in the columns; the instantiated shapes
        if (ntaps != 27 || !((Cin == 16 && Cout == 8) || (Cin == 32 && Cout == 16) || (Cin == 64 && Cout == 32))) return 0;
        return (size_t)(8 * Cin) / 32 * ((2 * Cout + 15) / 16) * 512 * 4;
    }
    return (size_t)(ntaps * Cin + 31) / 32 * ((Cout + 15) / 16) * 512;
}


// element i of the fragment layout -> (tap, ci, co) of the weight it holds; false: a zero (k or column padding, a parity without a tap at this offset)
__host__ __device__ inline bool mvs_conv3d_bf16_coords(int64_t i, int Cin, int Cout, int transposed, int& tap, int& ci, int& co, int ntaps = 27)
{
    const int log = (Cin == 4 ? 2 : Cin == 8 ? 3 : Cin == 16 ? 4 : Cin == 32 ? 5 : 6);
    const int NT = transposed ? (2 * Cout + 15) / 16 : (Cout + 15) / 16;
    const int ksn = transposed ? (8 * Cin) / 32 : (ntaps * Cin + 31) / 32;
    const int64_t per = (int64_t)ksn * NT * 64 * 8;
    const int cls = (int)(i / per);
    const int64_t r = i - cls * per;
    const int j = (int)(r & 7), lane = (int)((r >> 3) & 63), nt = (int)((r >> 9) % NT), ks = (int)((r >> 9) / NT);
    const int kb = ks * 32 + (lane >> 4) * 8 + j, t = kb >> log, col = nt * 16 + (lane & 15);
    ci = kb & (Cin - 1);
    tap = -1; co = col;
    if (!transposed) { if (t < ntaps && co < Cout) tap = t; }
    else {
        const int pz = cls >> 1, py = cls & 1, px = col / Cout;
        co = col - px * Cout;
        if (t < (1 + pz) * (1 + py) * 2 && px < 2) {
            int bits = t;
            const int sx = bits & 1; bits >>= 1;
            const int sy = py ? (bits & 1) : 1; bits >>= py;
            const int sz = pz ? (bits & 1) : 1;
            const int kx = px ? (sx ? 2 : 0) : (sx ? 1 : -1), ky = py ? (sy ? 2 : 0) : 1, kz = pz ? (sz ? 2 : 0) : 1;
            if (kx >= 0) tap = kz * 9 + ky * 3 + kx;
        }
    }
    return tap >= 0;
}

