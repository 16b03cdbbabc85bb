// LDS-DMA (global_load_lds_dwordx4) with explicit addressing for gfx950.
//
// A wave-instruction moves 1 KB (64 lanes x 16 B) from global memory straight into LDS.  The compiler builtin takes a
// per-lane 64-bit global pointer; fed `src + piece * 256 + lane * 4` in an unrolled loop it materialises one 64-bit VGPR
// address per piece, hoists them all out of the surrounding layer loop and - in the register-tight MLP kernels - spills
// some of them to scratch.  The reload then needs `s_waitcnt vmcnt(0)`, which also waits for every DMA piece issued just
// before it: a full L2 round trip on the critical path right after each weight-slab barrier.  Here the global base stays
// in an SGPR pair (saddr form), the LDS destination goes through M0, and the only VGPR is the shared lane offset.
#pragma once
#include <hip/hip_runtime.h>

// byte address of an LDS object inside the workgroup's allocation
__device__ __forceinline__ unsigned lds_byte_addr(const void* p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}

// one piece: lanes read 16 B each at src + lane_byte (src wave-uniform), LDS receives them at lds_byte + lane * 16
__device__ __forceinline__ void lds_dma_1k(const void* src_uniform, unsigned lds_byte_uniform, unsigned lane_byte)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(lane_byte), "s"(src_uniform), "s"(lds_byte_uniform) : "memory");   // M0 is not allocatable: the compiler only writes it right before its own M0 users
}

// n_pieces KB from src to dst, piece i handled by wave i % WAVES (wave = this wave's index, wave-uniform)
template <int WAVES>
__device__ __forceinline__ void lds_dma(void* __restrict__ dst, const void* __restrict__ src, int n_pieces, int wave, int lane)
{
    const char* s = reinterpret_cast<const char*>(src) + wave * 1024;
    unsigned l = lds_byte_addr(dst) + wave * 1024;
    const unsigned lb = lane * 16;
    for (int pc = wave; pc < n_pieces; pc += WAVES, s += WAVES * 1024, l += WAVES * 1024) lds_dma_1k(s, l, lb);
}

// the same with a compile-time piece count: no scalar branches when every wave gets the same number of pieces
template <int WAVES, int PIECES>
__device__ __forceinline__ void lds_dma_c(void* __restrict__ dst, const void* __restrict__ src, int wave, int lane)
{
    const char* s = reinterpret_cast<const char*>(src) + wave * 1024;
    const unsigned l = lds_byte_addr(dst) + wave * 1024;
    const unsigned lb = lane * 16;
#pragma unroll
    for (int k = 0; k < (PIECES + WAVES - 1) / WAVES; ++k)
        if ((k + 1) * WAVES <= PIECES || wave + k * WAVES < PIECES) lds_dma_1k(s + k * WAVES * 1024, l + k * WAVES * 1024, lb);
}
