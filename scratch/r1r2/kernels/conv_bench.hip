// Stand-alone timing harness for the matrix-core conv0 kernels (mvsnerf_amd/csrc/conv_mfma.hip): the register-staged kernel (channel-last
// or 8-channel-chunk input) with parts switched off, and the DMA-staged kernel on a cost volume in channel blocks of four.
#define MVS_CONV_DBG 1
#include "../../mvsnerf_amd/csrc/conv_mfma.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int D = 128, H = 176, W = 208, CIN = 44;
    const int64_t nvox = (int64_t)D * H * W;
    float *x, *w, *out;
    hipMalloc(&x, nvox * 48 * 4); hipMalloc(&w, 27 * CIN * 8 * 4); hipMalloc(&out, nvox * 8 * 4);
    std::vector<float> hx(1 << 20);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.f - 0.5f;
    for (int64_t o = 0; o < nvox * 48; o += (int64_t)hx.size()) hipMemcpy(x + o, hx.data(), std::min<int64_t>(hx.size(), nvox * 48 - o) * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hx.data(), 27 * CIN * 8 * 4, hipMemcpyHostToDevice);
    ActSrc a{x, nullptr, nullptr}, b{nullptr, nullptr, nullptr};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int dbg : {0, 1, 2, 4, 6, 7, 3, 0}) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_dbg), &dbg, sizeof(int));
        for (int rep = 0; rep < 3; ++rep) mvs_conv3d_c8_mfma(a, b, CIN, 41, -8, D, H, W, w, out, 1, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) mvs_conv3d_c8_mfma(a, b, CIN, 41, -8, D, H, W, w, out, 1, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dbg %d (1 = no staging, 2 = no operand reads, 4 = no weight loads): %.3f ms  -> %.1f TFLOP/s issued\n", dbg, ms / 5,
               (double)nvox * 27 * CIN * 8 * 2 / (ms / 5 * 1e-3) / 1e12);
    }
    {   // DMA-staged kernel (blocks of four): x holds enough floats for [11][nvox][4]
        float* wq; hipMalloc(&wq, 27 * CIN * 8 * 4);
        mvs_conv_w4_repack(w, wq, CIN, 0);
        for (int rep = 0; rep < 3; ++rep) mvs_conv3d_c8_mfma4(x, CIN, 41, D, H, W, wq, out, 1, 0);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) mvs_conv3d_c8_mfma4(x, CIN, 41, D, H, W, wq, out, 1, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("DMA-staged kernel (cost volume in channel blocks of four): %.3f ms -> %.1f TFLOP/s algorithmic (41 channels)\n", ms / 5,
               (double)nvox * 27 * 41 * 8 * 2 / (ms / 5 * 1e-3) / 1e12);
    }
    return 0;
}
