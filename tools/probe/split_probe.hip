// dev probe (GPU): the hi/lo split of mlp16_core.hpp as v_cvt_pk_f16_f32 + v_fma_mixlo/mixhi_f16 (3 instructions per pair) against the
// plain C++ form (hi = fp16(x), lo = fp16(x - float(hi)): ~7 per pair as the compiler emits it) -- bit for bit, over random bit
// patterns of every exponent incl. fp16 subnormal / overflow ranges, NaN and inf.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o split_probe split_probe.hip && ./split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const float* x, unsigned long long* bad, unsigned* first, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; 2 * i + 1 < n; i += (long)gridDim.x * blockDim.x) {
        const float a = x[2 * i], b = x[2 * i + 1];
        unsigned h, l;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h) : "v"(a), "v"(b));
        asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(a), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(b), "v"(h));
        const _Float16 ha = (_Float16)a, hb = (_Float16)b;
        const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
        const unsigned h_ref = (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
        const unsigned l_ref = (unsigned)__builtin_bit_cast(unsigned short, la) | ((unsigned)__builtin_bit_cast(unsigned short, lb) << 16);
        const bool nan_ok = (a != a) || (b != b);   // NaN payloads may differ
        if ((h != h_ref || l != l_ref) && !nan_ok) {
            if (atomicAdd(bad, 1ull) == 0) { first[0] = __builtin_bit_cast(unsigned, a); first[1] = __builtin_bit_cast(unsigned, b); first[2] = h; first[3] = h_ref; first[4] = l; first[5] = l_ref; }
        }
    }
}
int main() {
    const long n = 1L << 26;
    std::vector<float> x(n);
    uint64_t s = 88172645463325252ull;
    for (long i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        uint32_t u = (uint32_t)(s >> 16);
        if ((i & 3) == 0) {   // a quarter of the values in the ranges the kernel sees: |x| in [2^-30, 2^17]
            const uint32_t e = 97 + (u >> 23) % 48;
            u = (u & 0x807fffffu) | (e << 23);
        }
        x[i] = *reinterpret_cast<float*>(&u);
    }
    float* dx; unsigned long long* dbad; unsigned* dfirst;
    hipMalloc(&dx, n * 4); hipMalloc(&dbad, 8); hipMalloc(&dfirst, 32);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice); hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, dx, dbad, dfirst, n);
    unsigned long long bad; unsigned first[8];
    hipMemcpy(&bad, dbad, 8, hipMemcpyDeviceToHost); hipMemcpy(first, dfirst, 32, hipMemcpyDeviceToHost);
    printf("%ld pairs, %llu differ", n / 2, bad);
    if (bad) printf("  first: a=%08x b=%08x hi %08x vs %08x lo %08x vs %08x", first[0], first[1], first[2], first[3], first[4], first[5]);
    printf("\n");
    return bad != 0;
}
