// Stand-alone timing harness for conv0's matrix-core weight gradient (conv_mfma.hip: conv3d_k3s1_c8_wgrad4_kernel) with the DMA or the
// MFMA phase switched off.   hipcc --offload-arch=gfx950 -O3 -std=c++17 scratch/r2/wgrad_bench.hip -o /tmp/wgrad_bench
#define MVS_CONV_DBG 1
#include "../../mvsnerf_amd/csrc/conv_mfma.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int D = 128, H = 176, W = 208, CIN = 44, CREAL = 41;
    const int64_t nvox = (int64_t)D * H * W;
    float *x, *g, *gw, *ws;
    const int cap = 2048;
    hipMalloc(&x, nvox * CIN * 4); hipMalloc(&g, nvox * 8 * 4); hipMalloc(&gw, 8 * CREAL * 27 * 4); hipMalloc(&ws, (size_t)(cap + MVS_RED_SLICES) * 8 * CREAL * 27 * 4);
    std::vector<float> hx(1 << 20);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    for (int64_t o = 0; o < nvox * CIN; o += (int64_t)hx.size()) hipMemcpy(x + o, hx.data(), std::min<int64_t>(hx.size(), nvox * CIN - o) * 4, hipMemcpyHostToDevice);
    for (int64_t o = 0; o < nvox * 8; o += (int64_t)hx.size()) hipMemcpy(g + o, hx.data(), std::min<int64_t>(hx.size(), nvox * 8 - o) * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dbg : {0, 8, 16, 24, 0}) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &dbg, sizeof(int));
        for (int rep = 0; rep < 3; ++rep) mvs_conv3d_c8_wgrad4(x, CIN, CREAL, D, H, W, g, gw, ws, cap, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) mvs_conv3d_c8_wgrad4(x, CIN, CREAL, D, H, W, g, gw, ws, cap, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dbg %2d (8 = no DMA, 16 = no MFMA phase): %.3f ms  -> %.1f TFLOP/s algorithmic (41 channels)\n", dbg, ms / 5,
               (double)nvox * 27 * CREAL * 8 * 2 / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
