"""Regenerates encoder_variant.hip (the victim-side variants of the shared-GPU probes: PSW_VARIANT_4WAVES, PSW_VARIANT_WAVES_PER_EU) from the encoder.hip of the commit
BEFORE the plane sweep moved to csrc/planesweep.hip (8472d93); run in the repository (needs git).  The file itself is not kept: it is 2000 lines of that revision plus the macros below.
For variants of the CURRENT sweep use planesweep_variant.hip / build_sweep_variants.sh."""
import subprocess, os
HERE = os.path.dirname(os.path.abspath(__file__))
src = subprocess.run(["git", "show", "8472d93:mvsnerf_amd/csrc/encoder.hip"], capture_output=True, text=True, check=True, cwd=HERE).stdout
a = src.index('template <int C, int NP>   // feature channels (32), depth planes per wave')
b = src.index('// fp32 half of a guarded sequence (mvsnerf_sweep_conv0_guarded_fwd)')
tile = src[a:b].replace('threadIdx.x', 'PSW_TID')
tile = tile.replace('    extern __shared__ __attribute__((aligned(16))) float lds_[];',
                    '    extern __shared__ __attribute__((aligned(16))) float lds_all_[];\n    float* lds_ = lds_all_ + PSW_LDS_SLICE;')
tile = tile.replace('__syncthreads();', 'PSW_SYNC();')
hdr = '''#ifdef PSW_VARIANT_4WAVES
// variant: FOUR independent waves per workgroup (each its own tile and LDS slice, no s_barrier): is the single-wave workgroup the victim condition?
#define PSW_TID (threadIdx.x & 63)
#define PSW_LDS_SLICE ((threadIdx.x >> 6) * lds_floats_per_wave)
#define PSW_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define PSW_TID threadIdx.x
#define PSW_LDS_SLICE 0
#define PSW_SYNC() __syncthreads()
#endif
'''
tile = tile.replace('    int* __restrict__ guard)            // blocked 3', '    int* __restrict__ guard, int lds_floats_per_wave = 0)            // blocked 3')
tile = tile.replace('__global__ __launch_bounds__(64) void planesweep_kernel(', '''#ifdef PSW_VARIANT_4WAVES
__global__ __launch_bounds__(256) void planesweep_kernel(
    const float* __restrict__ feat, const float* __restrict__ img, const float* __restrict__ proj, const float* __restrict__ depth,
    int V, int H, int W, int D, int pad, float* __restrict__ cost, int CP, float* __restrict__ masks, int with_img, int blocked, int* __restrict__ guard, unsigned n_wg, int lds_floats_per_wave)
{
    const unsigned bid = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bid >= n_wg) return;
    planesweep_tile<C, NP>(bid, feat, img, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, guard, lds_floats_per_wave);
}
#else
#ifdef PSW_VARIANT_WAVES_PER_EU
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(PSW_VARIANT_WAVES_PER_EU))) void planesweep_kernel(
#else
__global__ __launch_bounds__(64) void planesweep_kernel(
#endif''')
tile = tile.replace('''    planesweep_tile<C, NP>(blockIdx.x, feat, img, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, guard);
}
''', '''    planesweep_tile<C, NP>(blockIdx.x, feat, img, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, guard);
}
#endif
''', 1)
out = src[:a] + hdr + tile + src[b:]
out = out.replace('''    planesweep_kernel<32, 4><<<n_wg, 64, lds, (hipStream_t)stream>>>(feats_cl, imgs_cl, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, guard);''', '''#ifdef PSW_VARIANT_4WAVES
    planesweep_kernel<32, 4><<<(n_wg + 3) / 4, 256, 4 * lds, (hipStream_t)stream>>>(feats_cl, imgs_cl, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, guard, n_wg, (int)(lds / 4));
#else
    planesweep_kernel<32, 4><<<n_wg, 64, lds, (hipStream_t)stream>>>(feats_cl, imgs_cl, proj, depth, V, H, W, D, pad, cost, CP, masks, with_img, blocked, guard);
#endif''')
open(os.path.join(HERE, 'encoder_variant.hip'), 'w').write(out)
print("wrote encoder_variant.hip", len(out.splitlines()), "lines")
