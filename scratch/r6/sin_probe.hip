// round 6 probe: absolute error of v_sin_f32 / v_cos_f32 (input in revolutions) against double precision, with an exact range reduction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* r, float* s, float* c, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { s[i] = __builtin_amdgcn_sinf(r[i]); c[i] = __builtin_amdgcn_cosf(r[i]); }
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> r(n), s(n), c(n);
    for (int i = 0; i < n; ++i) r[i] = (i + 0.37f) / n - 0.5f;            // revolutions in [-0.5, 0.5)
    float *dr, *ds, *dc;
    hipMalloc(&dr, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dc, n * 4);
    hipMemcpy(dr, r.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dr, ds, dc, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost);
    double es = 0, ec = 0, es_small = 0;
    for (int i = 0; i < n; ++i) {
        const double a = 2.0 * M_PI * (double)r[i];
        es = fmax(es, fabs(s[i] - sin(a))); ec = fmax(ec, fabs(c[i] - cos(a)));
        if (fabs(r[i]) < 0.01) es_small = fmax(es_small, fabs(s[i] - sin(a)) / fmax(fabs(sin(a)), 1e-30));
    }
    printf("v_sin_f32 max abs err %.3g, v_cos_f32 max abs err %.3g over [-0.5, 0.5) revolutions; sin relative err for |r| < 0.01: %.3g (fp32 half ulp at 1: 6e-8)\n", es, ec, es_small);
    return 0;
}
