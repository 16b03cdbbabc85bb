// round 6 probe: the hardware-sine positional encoding of mlp_b16_dev.h against double precision
#include "../../mvsnerf_amd/csrc/mlp_b16_dev.h"
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* x, float* s, float* c, float* s2, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mlp::PeArg a = mlp::pe_arg(x[i]);
    for (int f = 0; f < 10; ++f) {
        s[i * 10 + f] = mlp::pe_sc(a, f, 0); c[i * 10 + f] = mlp::pe_sc(a, f, 1);
        // the polynomial routine this replaces (mlp.hip, pe_sc of rounds 1-5)
        float xx = x[i] * (float)(1 << f);
        xx = fminf(fmaxf(xx, -65536.0f), 65536.0f);
        const float kk = rintf(xx * 0.63661977236758134f);
        float r = fmaf(kk, -1.5703125f, xx);
        r = fmaf(kk, -4.837512969970703125e-4f, r);
        r = fmaf(kk, -7.54978995489188e-8f, r);
        const float r2 = r * r;
        const float sn = fmaf(fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f) * r2, r, r);
        const float cs = fmaf(fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f) * r2, r2, fmaf(-0.5f, r2, 1.0f));
        const int q = (int)kk;
        const float v = (q & 1) ? cs : sn;
        s2[i * 10 + f] = (q & 2) ? -v : v;
    }
}
int main()
{
    const int n = 1 << 18;
    std::vector<float> x(n), s(n * 10), c(n * 10), s2(n * 10);
    for (int i = 0; i < n; ++i) x[i] = -0.6f + 2.2f * (i + 0.41f) / n;
    float *dx, *ds, *dc, *ds2;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&ds, n * 40); (void)hipMalloc(&dc, n * 40); (void)hipMalloc(&ds2, n * 40);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, ds, dc, ds2, n);
    (void)hipMemcpy(s.data(), ds, n * 40, hipMemcpyDeviceToHost); (void)hipMemcpy(c.data(), dc, n * 40, hipMemcpyDeviceToHost); (void)hipMemcpy(s2.data(), ds2, n * 40, hipMemcpyDeviceToHost);
    for (int f = 0; f < 10; ++f) {
        double es = 0, ec = 0, e2 = 0;
        for (int i = 0; i < n; ++i) {
            const double a = (double)x[i] * (1 << f);
            es = fmax(es, fabs(s[i * 10 + f] - sin(a))); ec = fmax(ec, fabs(c[i * 10 + f] - cos(a))); e2 = fmax(e2, fabs(s2[i * 10 + f] - sin(a)));
        }
        printf("f %d: sin err %.3g  cos err %.3g   polynomial routine (rounds 1-5) sin %.3g\n", f, es, ec, e2);
    }
    return 0;
}
