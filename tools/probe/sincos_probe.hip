// dev probe: pe_sincos / pe_sincos_hw (csrc/common.hpp) against double precision over the range the positional encodings see
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I danbo-pytorch_amd/csrc -o sincos_probe tools/probe/sincos_probe.hip && ./sincos_probe
#include "common.hpp"
#include <cmath>
#include <cstdio>
#include <vector>
using namespace danbo;
// the candidate that was measured and dropped: the hardware's v_sin_f32 / v_cos_f32 (arguments in revolutions) behind a two-term
// Cody-Waite reduction by 2 pi, the rounding of the reduced argument put back to first order.  Result on gfx950: 2.2e-7 max abs
// error against 6.9e-8 of the polynomial pe_sincos; K3 1.5 % faster with it (3.3 % without the correction, 2.4e-5 instead of
// 2.0e-5 from the exact-fp32 kernel's logits); the training instantiation of the same body failed its gradient tests with it.
__device__ __forceinline__ void pe_sincos_hw(float a, float* sn, float* cs) {
    const float k = __builtin_rintf(a * 0.15915494309189535f);
    float r = __builtin_fmaf(k, -6.2831854820251465f, a);
    r = __builtin_fmaf(k, 1.7484555e-7f, r);
    const float rev = r * 0.15915494309189535f;
    const float s0 = __builtin_amdgcn_sinf(rev), c0 = __builtin_amdgcn_cosf(rev);
    float d = __builtin_fmaf(rev, -6.2831854820251465f, r);
    d = __builtin_fmaf(rev, 1.7484555e-7f, d);
    *sn = __builtin_fmaf(d, c0, s0);
    *cs = __builtin_fmaf(-d, s0, c0);
}
__global__ void k(const float* x, int n, float* o) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    pe_sincos(x[i], &s, &c);
    o[4 * i] = s; o[4 * i + 1] = c;
    pe_sincos_hw(x[i], &s, &c);
    o[4 * i + 2] = s; o[4 * i + 3] = c;
}
int main() {
    const int n = 1 << 22;
    std::vector<float> x(n);
    for (int i = 0; i < n; ++i) x[i] = -400.f + 800.f * (float)i / (float)(n - 1);
    x[0] = 0.f; x[1] = -0.f; x[2] = 1e-30f; x[3] = 3.14159265f; x[4] = 6.2831853f; x[5] = 1.5707963f; x[6] = 1e4f; x[7] = -1e4f;
    float *dx, *dout;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&dout, n * 16);
    (void)hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, n, dout);
    std::vector<float> o(4 * (size_t)n);
    (void)hipMemcpy(o.data(), dout, n * 16, hipMemcpyDeviceToHost);
    double e[4] = {0, 0, 0, 0}; int at[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const double s = sin((double)x[i]), c = cos((double)x[i]);
        const double d[4] = {fabs(o[4 * i] - s), fabs(o[4 * i + 1] - c), fabs(o[4 * i + 2] - s), fabs(o[4 * i + 3] - c)};
        for (int j = 0; j < 4; ++j) if (!(d[j] <= e[j])) { e[j] = d[j]; at[j] = i; }
    }
    printf("max abs error  polynomial sin %.3g (x=%g) cos %.3g (x=%g)   hardware sin %.3g (x=%g) cos %.3g (x=%g)\n", e[0], x[at[0]], e[1], x[at[1]], e[2], x[at[2]], e[3], x[at[3]]);
    for (int i = 0; i < 8; ++i) printf("x=%g  poly %.9g %.9g  hw %.9g %.9g  exact %.9g %.9g\n", x[i], o[4*i], o[4*i+1], o[4*i+2], o[4*i+3], sin((double)x[i]), cos((double)x[i]));
    return 0;
}
