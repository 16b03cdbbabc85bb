#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// copy n_pieces KB from global to LDS via global_load_lds (16 B per lane), then write LDS back out
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ dst, int pieces_per_wave)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = 0; j < pieces_per_wave; ++j) {
        const int piece = wave * pieces_per_wave + j;
        const float* g = src + piece * 256 + lane * 4;                       // per-lane global address
        float* l = lds + piece * 256;                                         // wave-uniform LDS base of the piece
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    }
    __syncthreads();
    const int n = 4 * pieces_per_wave * 256;
    for (int i = threadIdx.x; i < n; i += 256) dst[i] = lds[i];
}
int main() {
    const int ppw = 8, n = 4 * ppw * 256;
    std::vector<float> h(n), o(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *d, *e; hipMalloc(&d, n * 4); hipMalloc(&e, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, n * 4);
    k<<<1, 256, n * 4>>>(d, e, ppw);
    hipMemcpy(o.data(), e, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; ++i) if (o[i] != h[i]) { if (bad < 5) printf("mismatch %d: %f\n", i, o[i]); ++bad; }
    printf("glds copy test: %d mismatches of %d\n", bad, n);
    return bad != 0;
}
