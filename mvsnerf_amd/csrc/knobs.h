// A/B switches between kernel variants.
//
// PRODUCT build (libmvsnerf_hip.so, `make`): every switch is a compile-time constant - the library has no behavioural global state
// (SURVEY 8b), the losing variants are not even compiled into it, and there is no entry point to flip anything.
// DEV build (`make dev` -> scratch/lib/libmvsnerf_hip_dev.so, -DMVSNERF_DEV_KNOBS): the switches are mutable ints behind
// mvsnerf_tune() (declared in scratch/mvsnerf_hip_dev.h, NOT in include/mvsnerf_hip.h) so that scratch/dev_tests/ and the A/B scripts
// can time and cross-check the alternatives that DESIGN.md reports as measured-and-dropped.
#pragma once

#ifdef MVSNERF_DEV_KNOBS
#define MVS_KNOB_DECL(name, value) extern int name;
#define MVS_KNOB_DEF(name, value) int name = value;
#else
#define MVS_KNOB_DECL(name, value) static constexpr int name = value;
#define MVS_KNOB_DEF(name, value)
#endif

MVS_KNOB_DECL(g_conv_mfma, 1)       // convolutions / weight gradients on the matrix cores where a kernel exists (0: the VALU kernels)
MVS_KNOB_DECL(g_conv_tiled, 1)      // LDS-tiled VALU 3x3x3 kernels (0: one thread per voxel through L1)
MVS_KNOB_DECL(g_conv_xcd, 1)        // tiles renumbered so that an XCD walks a contiguous range (0: round-robin)
MVS_KNOB_DECL(g_psw_fwd_reuse, 1)    // plane sweep: a workgroup walks 64 columns x 4 planes and keeps unchanged taps in registers; 0: 256 voxels of one plane
MVS_KNOB_DECL(g_psw_bwd_tiles, 2)   // plane-sweep backward: 2 column form (taps and sums in registers along depth), 1 LDS-patch tiles, 0 one float atomic per tap
MVS_KNOB_DECL(g_mlp_variant, 3)     // 3: 32 points/wave, 2 waves/SIMD, LDS-DMA weight slabs; 0/1/2/4: the dropped schedules
MVS_KNOB_DECL(g_mlp_gather, 0)      // 1: gen_pts_feats in the MLP kernel's prologue (measured 1 % slower)
MVS_KNOB_DECL(g_split_sched, 0)     // bf16x6 kernel: 0 = two waves/SIMD, 1 = one wave/SIMD hand-interleaved
