#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int stride_elems)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, i = l & 15;
    // hypothesis: lane i of a 16-lane group fetches row (i>>2), columns 4(i&3)..+3 of a [4][16] block; gets column i, rows 0..3
    const unsigned short* p = lds + g * 4 * stride_elems + (i >> 2) * stride_elems + (i & 3) * 4;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int stride : {16, 32, 40}) {
        k<<<1, 64>>>(d, stride);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d\n", stride);
        for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d (r%d c%d)", h[l*4+j], (h[l*4+j] / stride) , h[l*4+j] % stride); printf("\n"); }
    }
    return 0;
}
