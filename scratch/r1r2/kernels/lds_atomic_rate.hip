// LDS atomic throughput on gfx950: wave-instructions of ds_add_f32 / ds_add_u32 / ds_add_u64 / (ds_read + add + ds_write) per CU clock.
// 256 threads (4 waves) per workgroup, one workgroup per CU, every lane its own address (no conflicts), 4096 operations per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters)
{
    __shared__ unsigned long long buf[4096];            // 32 KB
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0;
    __syncthreads();
    float* f = reinterpret_cast<float*>(buf);
    unsigned* u = reinterpret_cast<unsigned*>(buf);
    const int t = threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int slot = ((it + j) & 15) * 256 + t;
            if (MODE == 0) atomicAdd(f + slot, 1.0f);
            if (MODE == 1) atomicAdd(u + slot, 1u);
            if (MODE == 2) atomicAdd(buf + slot, 1ull);
            if (MODE == 3) f[slot] += 1.0f;
        }
    }
    __syncthreads();
    out[blockIdx.x * 256 + t] = f[t] + (float)buf[t + 256];
}
int main()
{
    float* out; hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 256;
    const char* names[4] = {"ds_add_f32", "ds_add_u32", "ds_add_u64", "read+add+write"};
    for (int m = 0; m < 4; ++m) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (m == 0) k<0><<<256, 256>>>(out, iters); if (m == 1) k<1><<<256, 256>>>(out, iters);
            if (m == 2) k<2><<<256, 256>>>(out, iters); if (m == 3) k<3><<<256, 256>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double wave_instrs = 4.0 * iters * 16;
        printf("%-16s %.3f ms  -> %.1f ns per wave-instruction per CU (4 waves issuing), ~%.0f clocks at 2.1 GHz\n", names[m], ms, ms * 1e6 / wave_instrs,
               ms * 1e6 / wave_instrs * 2.1);
    }
    return 0;
}
