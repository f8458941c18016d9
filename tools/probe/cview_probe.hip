// Stand-alone repro attempt for the non-repeatable view constants of round 4 ("A non-repeatable step"; VERDICT r5
// item 6): k_train_cview -- 128 threads = features, 8 rays per workgroup, the rays' 155 view inputs in LDS, eight FMA chains per thread
// that the compiler pairs on v_pk_fma_f32 -- produced ONE wrong term in the sum of one ray in the last 16 lanes of a wavefront about
// once in 200 training steps while K2 (MFMA + LDS + L2 gathers) ran on the same CUs.  This program runs the SAME kernel body (copied
// from csrc/k_train_head.hip; packed form as the compiler emits it, and the scalar form that never failed) beside a neighbour kernel
// on a second stream and checks every output against a reference computed once without a neighbour:
//   neighbour 0: none            1: MFMA only (32x32x16 f16 chains, no LDS)        2: LDS only (ds_read / ds_write traffic)
//   neighbour 3: MFMA + LDS (K2's mix)            4: MFMA + LDS + a v_permlane32_swap / DPP mix (K2's lane trades)
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o cview_probe cview_probe.hip && ./cview_probe [iterations]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

constexpr int HW = 256, HVW = 128, CV_RAYS = 8;

// FORM 0: the product kernel as the compiler emits it (v_pk_fma_f32 pairs, ds_read_b128, counted lgkmcnt waits)
//      1: every FMA chain kept scalar by an empty asm (the form that never failed in the training step)
//      2: the same source compiled for a target WITHOUT packed fp32 instructions (wide LDS reads stay)
//      3: LDS rows 161 floats apart: no 16-byte LDS reads (packed FMAs stay)
//      4: an explicit s_waitcnt lgkmcnt(0) between the LDS reads of a k and their use (packed FMAs, wide reads stay)
#define FORM_ATTR
template <int FORM, int LDS_LD>
__device__ __forceinline__ void cview_body(const float* __restrict__ vin, int ldv, int Cv, const float* __restrict__ views_w,
                                           const float* __restrict__ b_eff, int R, float* __restrict__ cview) {
    constexpr bool SCALAR = FORM == 1;
    __shared__ float s_v[CV_RAYS][LDS_LD];
    const int f = threadIdx.x;
    const float* wrow = views_w + (size_t)f * (HW + Cv) + HW;
    for (int r0 = blockIdx.x * CV_RAYS; r0 < R; r0 += gridDim.x * CV_RAYS) {
        __syncthreads();
        for (int i = threadIdx.x; i < CV_RAYS * Cv; i += 128) {
            const int rr = i / Cv, k = i % Cv;
            s_v[rr][k] = r0 + rr < R ? vin[(size_t)(r0 + rr) * ldv + k] : 0.f;
        }
        __syncthreads();
        float acc[CV_RAYS];
#pragma unroll
        for (int rr = 0; rr < CV_RAYS; ++rr) acc[rr] = b_eff[f];
        for (int k = 0; k < Cv; ++k) {
            const float w = wrow[k];
            float sv[CV_RAYS];
#pragma unroll
            for (int rr = 0; rr < CV_RAYS; ++rr) sv[rr] = s_v[rr][k];
            if (FORM == 4) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int rr = 0; rr < CV_RAYS; ++rr) asm volatile("" : "+v"(sv[rr]));
            }
#pragma unroll
            for (int rr = 0; rr < CV_RAYS; ++rr) {
                acc[rr] = fmaf(sv[rr], w, acc[rr]);
                if (SCALAR) asm volatile("" : "+v"(acc[rr]));
            }
        }
#pragma unroll
        for (int rr = 0; rr < CV_RAYS; ++rr)
            if (r0 + rr < R) cview[(size_t)(r0 + rr) * HVW + f] = acc[rr];
    }
}

template <int FORM>
__global__ __launch_bounds__(128) void k_cview(const float* __restrict__ vin, int ldv, int Cv, const float* __restrict__ views_w,
                                               const float* __restrict__ b_eff, int R, float* __restrict__ cview) {
    cview_body<FORM, FORM == 3 ? 161 : 160>(vin, ldv, Cv, views_w, b_eff, R, cview);
}
// (a target attribute cannot depend on a template parameter: FORM 2 is its own kernel)
__global__ __launch_bounds__(128) __attribute__((target("no-packed-fp32-ops"))) void k_cview_nopk(const float* __restrict__ vin, int ldv, int Cv,
                                                                                                 const float* __restrict__ views_w,
                                                                                                 const float* __restrict__ b_eff, int R,
                                                                                                 float* __restrict__ cview) {
    cview_body<2, 160>(vin, ldv, Cv, views_w, b_eff, R, cview);
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// the neighbour: 256 threads, two per SIMD like K2, ~34 KB of LDS; runs `iters` rounds of its mix
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_neighbour(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float s[8704];          // 34 KB
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 8704; i += 256) s[i] = (float)((i * 2654435761u) >> 20) * 1e-3f;
    __syncthreads();
    f32x16 acc = {};
    half8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (lane + e)); b[e] = (_Float16)(0.02f * (lane - e)); }
    float v = (float)lane, t = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE >= 3) {      // (5: + LDS reads only, 6: + LDS writes only)
#pragma unroll
            for (int u = 0; u < 6; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        if (MODE >= 2 && MODE != 6) {
#pragma unroll
            for (int u = 0; u < 30; ++u) t = fmaf(s[(lane * 7 + u * 131 + it * 17) % 8704], 0.5f, t);
        }
        if ((MODE >= 2 && MODE != 5) || MODE == 6) s[(tid * 13 + it) % 8704] = t * 1e-6f + (float)it;
        if (MODE == 4) {
            const unsigned uu = __builtin_bit_cast(unsigned, v);
            const auto r = __builtin_amdgcn_permlane32_swap(uu, uu, false, false);
            v = __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]) + 1.0f;
            v += __shfl_xor(v, 1, 64);
        }
    }
    float sum = t + v;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += acc[i];
    out[blockIdx.x * 256 + tid] = sum;
}

// The packed instructions alone, operands in registers (no LDS in this kernel): 256 dependent steps on register pairs, once as the
// packed instruction with the operand selection (SEL = op_sel bits of src0, src1, src2: which dword feeds the LOW half; SELHI = op_sel_hi:
// which dword feeds the HIGH half; default 0 / 7) and once as two scalar instructions on the selected dwords; any lane whose two results
// differ is counted per lane quarter.  OP 0: v_pk_fma_f32 (acc = a * b + acc), 1: v_pk_mul_f32 (acc = a * b), 2: v_pk_add_f32 (acc = a + b)
template <int OP, int SEL, int SELHI>
__global__ __launch_bounds__(128) void k_pk(const float* __restrict__ in, unsigned* __restrict__ hist, int rounds) {
    const int lane = threadIdx.x & 63;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 x = {in[4 * i], in[4 * i + 1]}, w = {in[4 * i + 2], in[4 * i + 3]};
    unsigned bad = 0;
#define PK_SEL3 " op_sel:[%c3,%c4,%c5] op_sel_hi:[%c6,%c7,%c8]"
#define PK_SEL2 " op_sel:[%c3,%c4] op_sel_hi:[%c6,%c7]"
    for (int r = 0; r < rounds; ++r) {
        f2 acc = {0.25f, -0.5f}, ref = acc;
#pragma unroll 1
        for (int t = 0; t < 256; ++t) {
            const f2 c_in = ref;
            if (OP == 0)
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" PK_SEL3 : "+v"(acc) : "v"(x), "v"(w), "n"(SEL & 1), "n"((SEL >> 1) & 1), "n"((SEL >> 2) & 1),
                             "n"(SELHI & 1), "n"((SELHI >> 1) & 1), "n"((SELHI >> 2) & 1));
            else if (OP == 1)
                asm volatile("v_pk_mul_f32 %0, %1, %2" PK_SEL2 : "=v"(acc) : "v"(x), "v"(w), "n"(SEL & 1), "n"((SEL >> 1) & 1), "n"(0),
                             "n"(SELHI & 1), "n"((SELHI >> 1) & 1), "n"(0));
            else
                asm volatile("v_pk_add_f32 %0, %1, %2" PK_SEL2 : "=v"(acc) : "v"(x), "v"(w), "n"(SEL & 1), "n"((SEL >> 1) & 1), "n"(0),
                             "n"(SELHI & 1), "n"((SELHI >> 1) & 1), "n"(0));
            const float a_lo = x[SEL & 1], b_lo = w[(SEL >> 1) & 1], c_lo = c_in[(SEL >> 2) & 1];
            const float a_hi = x[SELHI & 1], b_hi = w[(SELHI >> 1) & 1], c_hi = c_in[(SELHI >> 2) & 1];
            float r0, r1;
            if (OP == 0) {
                r0 = c_lo; r1 = c_hi;
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r0) : "v"(a_lo), "v"(b_lo));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r1) : "v"(a_hi), "v"(b_hi));
            } else if (OP == 1) {
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r0) : "v"(a_lo), "v"(b_lo));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r1) : "v"(a_hi), "v"(b_hi));
            } else {
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(r0) : "v"(a_lo), "v"(b_lo));
                asm volatile("v_add_f32 %0, %1, %2" : "=v"(r1) : "v"(a_hi), "v"(b_hi));
            }
            ref[0] = r0; ref[1] = r1;
            bad += (__builtin_bit_cast(unsigned, acc[0]) != __builtin_bit_cast(unsigned, ref[0])) |
                   (__builtin_bit_cast(unsigned, acc[1]) != __builtin_bit_cast(unsigned, ref[1]));
            acc = ref;                                                     // (errors do not propagate: every step is checked on its own)
            x[0] = x[0] * 0.999f + 0.001f; x[1] = x[1] * 0.998f - 0.001f;  // (keeps the values moving and bounded)
        }
    }
    if (bad) atomicAdd(hist + lane, bad);
}

template <int OP, int SEL, int SELHI, int MODE>
void run_pk(int iters, const float* in, unsigned* hist, float* nb_out, hipStream_t s0, hipStream_t s1) {
    (void)hipMemsetAsync(hist, 0, 64 * 4, s0);
    for (int it = 0; it < iters; ++it) {
        if (MODE > 0) hipLaunchKernelGGL((k_neighbour<MODE>), dim3(512), dim3(256), 0, s1, nb_out, 40 + (it % 7) * 9);
        hipLaunchKernelGGL((k_pk<OP, SEL, SELHI>), dim3(1024), dim3(128), 0, s0, in, hist, 4);
        (void)hipStreamSynchronize(s0);
        (void)hipStreamSynchronize(s1);
    }
    unsigned h[64];
    (void)hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long tot = 0, q[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) { tot += h[l]; q[l >> 4] += h[l]; }
    const char* ops[3] = {"v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"};
    printf("%s op_sel:[%d,%d,%d] op_sel_hi:[%d,%d,%d] beside neighbour %d: %lu mismatches of %.0f M checks (lanes 0-15: %lu, 16-31: %lu, 32-47: %lu, 48-63: %lu)\n",
           ops[OP], SEL & 1, (SEL >> 1) & 1, (SEL >> 2) & 1, SELHI & 1, (SELHI >> 1) & 1, (SELHI >> 2) & 1, MODE, tot, iters * 134.2, q[0], q[1], q[2], q[3]);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int FORM, int MODE>
long run(const char* form, int iters, int R, int Cv, int ldv, const float* vin, const float* w, const float* beff, float* out, const float* ref,
         float* nb_out, hipStream_t s0, hipStream_t s1) {
    std::vector<float> host((size_t)R * HVW), want((size_t)R * HVW);
    CK(hipMemcpy(want.data(), ref, want.size() * 4, hipMemcpyDeviceToHost));
    long bad_iters = 0, bad_vals = 0;
    int first_ray = -1, first_f = -1;
    for (int it = 0; it < iters; ++it) {
        if (MODE > 0) hipLaunchKernelGGL((k_neighbour<MODE>), dim3(512), dim3(256), 0, s1, nb_out, 40 + (it % 7) * 9);
        // a few empty-ish launches move the relative timing around, like the step's prologue does
        if (it % 3 == 1) hipLaunchKernelGGL((k_neighbour<0>), dim3(1), dim3(256), 0, s0, nb_out + 512 * 256, 1);
        if (FORM == 2) hipLaunchKernelGGL(k_cview_nopk, dim3(R / CV_RAYS), dim3(128), 0, s0, vin, ldv, Cv, w, beff, R, out);
        else hipLaunchKernelGGL((k_cview<FORM>), dim3(R / CV_RAYS), dim3(128), 0, s0, vin, ldv, Cv, w, beff, R, out);
        CK(hipStreamSynchronize(s0));
        CK(hipMemcpy(host.data(), out, host.size() * 4, hipMemcpyDeviceToHost));
        long nb = 0;
        for (size_t i = 0; i < host.size(); ++i)
            if (memcmp(&host[i], &want[i], 4) != 0) {
                if (first_ray < 0) { first_ray = (int)(i / HVW); first_f = (int)(i % HVW); }
                ++nb;
            }
        bad_vals += nb;
        bad_iters += nb != 0;
        CK(hipStreamSynchronize(s1));
    }
    printf("%-7s neighbour %d: %6d launches, %ld with a wrong output (%ld values)%s", form, MODE, iters, bad_iters, bad_vals, bad_iters ? "" : "\n");
    if (bad_iters) printf("; first: ray %d feature %d (lane %d of its wavefront)\n", first_ray, first_f, first_f & 63);
    return bad_iters;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int R = 3072, Cv = 155, ldv = 156;
    std::vector<float> h_vin((size_t)R * ldv), h_w((size_t)HVW * (HW + Cv)), h_b(HVW);
    unsigned st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return ((st >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (auto& x : h_vin) x = rnd() * 2.f;
    for (auto& x : h_w) x = rnd() * 0.2f;
    for (auto& x : h_b) x = rnd();
    float *vin, *w, *beff, *out, *ref, *nb_out;
    CK(hipMalloc(&vin, h_vin.size() * 4)); CK(hipMalloc(&w, h_w.size() * 4)); CK(hipMalloc(&beff, h_b.size() * 4));
    CK(hipMalloc(&out, (size_t)R * HVW * 4)); CK(hipMalloc(&ref, (size_t)R * HVW * 4)); CK(hipMalloc(&nb_out, (512 * 256 + 256) * 4));
    CK(hipMemcpy(vin, h_vin.data(), h_vin.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h_w.data(), h_w.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(beff, h_b.data(), h_b.size() * 4, hipMemcpyHostToDevice));
    hipStream_t s0, s1;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    // reference: the scalar form alone on an idle device (the packed form alone gives the same bits: checked below as neighbour 0)
    hipLaunchKernelGGL((k_cview<1>), dim3(R / CV_RAYS), dim3(128), 0, s0, vin, ldv, Cv, w, beff, R, ref);
    CK(hipStreamSynchronize(s0));
    long bad = 0;
    bad += run<0, 0>("packed", iters / 4, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<0, 1>("packed", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<0, 2>("packed", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<0, 3>("packed", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<0, 5>("packed", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<0, 6>("packed", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<1, 3>("scalar", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<2, 3>("no-pk", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<3, 3>("ld161", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    bad += run<4, 3>("wait0", iters, R, Cv, ldv, vin, w, beff, out, ref, nb_out, s0, s1);
    {
        std::vector<float> h_in((size_t)1024 * 128 * 4);
        for (auto& v : h_in) v = rnd();
        float* in; unsigned* hist;
        CK(hipMalloc(&in, h_in.size() * 4)); CK(hipMalloc(&hist, 64 * 4));
        CK(hipMemcpy(in, h_in.data(), h_in.size() * 4, hipMemcpyHostToDevice));
        const int n = iters / 10 > 100 ? iters / 10 : 100;
#define RUN_PK(OP, SEL, SELHI, MODE) run_pk<OP, SEL, SELHI, MODE>(n, in, hist, nb_out, s0, s1)
        // without a neighbour: every form is exact
        RUN_PK(0, 0, 7, 0); RUN_PK(0, 2, 7, 0); RUN_PK(1, 2, 3, 0); RUN_PK(2, 2, 3, 0);
        // beside MFMA + LDS reads: v_pk_fma_f32, one selection bit at a time
        RUN_PK(0, 0, 7, 5);   // default
        RUN_PK(0, 1, 7, 5);   // low half from src0's HIGH dword
        RUN_PK(0, 2, 7, 5);   // low half from src1's HIGH dword  (the form of k_train_cview)
        RUN_PK(0, 4, 7, 5);   // low half from src2's HIGH dword
        RUN_PK(0, 0, 6, 5);   // high half from src0's LOW dword
        RUN_PK(0, 0, 5, 5);   // high half from src1's LOW dword   (the other form of k_train_cview: never wrong there)
        RUN_PK(0, 0, 3, 5);   // high half from src2's LOW dword
        RUN_PK(0, 7, 0, 5);   // halves swapped
        // ... v_pk_mul_f32 / v_pk_add_f32
        RUN_PK(1, 0, 3, 5); RUN_PK(1, 1, 3, 5); RUN_PK(1, 2, 3, 5); RUN_PK(1, 0, 1, 5); RUN_PK(1, 0, 2, 5);
        RUN_PK(2, 0, 3, 5); RUN_PK(2, 1, 3, 5); RUN_PK(2, 2, 3, 5); RUN_PK(2, 0, 1, 5); RUN_PK(2, 0, 2, 5);
        // ... and the neighbour's ingredients for the failing form
        RUN_PK(0, 2, 7, 1); RUN_PK(0, 2, 7, 2); RUN_PK(0, 2, 7, 6); RUN_PK(0, 2, 7, 3);
    }
    printf("total launches with a wrong output: %ld\n", bad);
    return 0;
}
