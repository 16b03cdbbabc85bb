// Does global_load_lds_dwordx4 reach LDS addresses above 64 KB on gfx950 (M0 as the destination base)?  mlp_f16x3.hip puts its second
// weight slab at 64..128 KB.  One workgroup DMAs 1 KB pieces to LDS offsets 0, 60, 64, 100 and 150 KB and reads them back with ds_read.
// Build (in the container): hipcc --offload-arch=gfx950 -O2 scratch/r3/lds_dma_hi_probe.hip -o scratch/r3/lds_dma_hi_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mvsnerf_amd/csrc/lds_dma.h"

__global__ void probe(const unsigned* __restrict__ src, unsigned* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x;
    const int offs[5] = {0, 60 * 1024, 64 * 1024, 100 * 1024, 150 * 1024};
    for (int i = lane; i < 160 * 1024 / 4; i += 64) reinterpret_cast<unsigned*>(lds)[i] = 0xdeadbeefu;
    __syncthreads();
    for (int k = 0; k < 5; ++k) lds_dma_1k(src + k * 256, lds_byte_addr(lds + offs[k]), lane * 16);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int k = 0; k < 5; ++k)
        for (int j = 0; j < 4; ++j) out[k * 256 + lane * 4 + j] = reinterpret_cast<const unsigned*>(lds + offs[k])[lane * 4 + j];
}

int main()
{
    unsigned *src, *out, h[1280], r[1280];
    for (int i = 0; i < 1280; ++i) h[i] = 0x1000000u + i;
    hipMalloc(&src, sizeof(h)); hipMalloc(&out, sizeof(h));
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    probe<<<1, 64, 160 * 1024>>>(src, out);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(r, out, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[5] = {"0 KB", "60 KB", "64 KB", "100 KB", "150 KB"};
    int bad_total = 0;
    for (int k = 0; k < 5; ++k) {
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[k * 256 + i] != h[k * 256 + i];
        printf("LDS-DMA to offset %-7s: %s (%d of 256 words wrong, first word %08x)\n", names[k], bad ? "WRONG" : "ok", bad, r[k * 256]);
        bad_total += bad;
    }
    printf("launch status %d; LDS_DMA_HI_PROBE %s\n", (int)e, bad_total ? "FAIL" : "PASS");
    return bad_total != 0;
}
