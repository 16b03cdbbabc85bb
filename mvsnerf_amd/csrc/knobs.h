// Which kernel family a dispatcher picks where more than one exists.  Compile-time constants: the library has no behavioural state
//  and no entry point to flip anything.  (Rounds 1-3 had a second, mutable build of these for A/B timing; the measured-and-
// dropped variants it carried - MLP schedules 0/1/2/4, the gather fused into the MLP prologue, the 256-voxel plane-sweep forward, the
// LDS-patch tile form of the plane-sweep backward - live in the history only: `git show 46eff22:mvsnerf_amd/csrc/<file>`.)
#pragma once

static constexpr int g_conv_mfma = 1;    // convolutions / weight gradients on the matrix cores where a kernel exists (the VALU kernels serve the other shapes)
static constexpr int g_conv_tiled = 1;   // LDS-tiled VALU 3x3x3 kernels for the stride-1 shapes without a matrix-core kernel
static constexpr int g_conv_xcd = 1;     // tiles renumbered so that an XCD walks a contiguous range
