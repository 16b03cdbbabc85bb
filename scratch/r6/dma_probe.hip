// round 6 probe: how fast can every CU pull the SAME small weight buffer (L2 / MALL resident) into LDS with global_load_lds_dwordx4?
// (the MLP kernels stream 250-500 KB of weights per workgroup: is the bf16 kernel's 256 MB in 40 us = 6.4 TB/s a ceiling?)
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../mvsnerf_amd/csrc/lds_dma.h"
template <int WAVES>
__global__ void k(const char* src, int src_kb, int iters, int depth, unsigned* sink)
{
    extern __shared__ char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // each wave streams its own 1 KB pieces round-robin over the source, into a 64 KB LDS ring
    unsigned off = wave * 1024;
    for (int it = 0; it < iters; ++it) {
        for (int d = 0; d < depth; ++d) {
            lds_dma_1k(src + (off % (unsigned)(src_kb * 1024)), lds_byte_addr(lds) + ((off) & 65535u), lane * 16);
            off += WAVES * 1024;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = lds[(off >> 3) & 65535];
}
template <int WAVES>
void run(const char* d_src, unsigned* sink, int src_kb, int depth, int blocks_per_cu)
{
    const int iters = 400 / depth * 8;
    hipFuncSetAttribute((const void*)k<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    const int blocks = 256 * blocks_per_cu;
    k<WAVES><<<blocks, WAVES * 64, 65536>>>(d_src, src_kb, 8, depth, sink);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<WAVES><<<blocks, WAVES * 64, 65536>>>(d_src, src_kb, iters, depth, sink);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 5.0 * blocks * WAVES * (double)iters * depth * 1024;
    printf("waves/WG %d, WGs/CU %d, source %5d KB, %2d pieces in flight per wave: %7.2f TB/s  (%.0f GB/s per CU)\n", WAVES, blocks_per_cu, src_kb, depth, bytes / (ms * 1e-3) / 1e12,
           bytes / (ms * 1e-3) / 1e9 / 256);
}
int main()
{
    char* d; unsigned* sink;
    hipMalloc(&d, 64 << 20); hipMemset(d, 1, 64 << 20); hipMalloc(&sink, 1 << 16);
    for (int src_kb : {256, 512, 16384})
        for (int depth : {4, 8, 16}) {
            run<4>(d, sink, src_kb, depth, 2);
            run<8>(d, sink, src_kb, depth, 1);
            run<8>(d, sink, src_kb, depth, 2);
        }
    return 0;
}
