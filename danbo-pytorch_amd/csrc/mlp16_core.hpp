// Shared machinery of the register-resident MLP kernels (k_mlp16.hip: K3 and the forward of the training trunk;
// k_mlp16_bwd.hip: the input-gradient chain of the training trunk).  gfx950 only.
//
//   * the weight ring: 32 KB chunks of fp16 hi/lo MFMA A-fragments streamed L2 -> LDS by global_load_lds, four slots,
//     one hand-over (wait, barrier, refill) per chunk and wavefront;
//   * fp32-accurate products as hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16;
//   * the hi/lo split of eight fp32 values into two B fragments.
// The kernels differ in what they do between the chunks; the ring protocol is the same.
#pragma once
#include "common.hpp"

namespace danbo {

#ifndef DANBO_M16_BT
#define DANBO_M16_BT 2   // output tiles per batch of A-fragment reads (2 / 4 / 8 measured within 2 %)
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int M16_BM = 128;            // rows per workgroup iteration
constexpr int CHUNK_BYTES = 32768;     // 32 fragment pieces of 1 KB
constexpr int RING_SLOTS = 4;
constexpr int M16_THREADS = 512;

struct Pipe {
    const char* packed;
    char* ring;
    int issue_chunk, issue_slot, cons_slot, wave, lane;
    bool early;
};

// every wavefront loads 4 of the 32 pieces of a chunk
template <int NCH>
__device__ __forceinline__ void pipe_issue(Pipe& p) {
    const char* src = p.packed + (size_t)p.issue_chunk * CHUNK_BYTES + p.wave * 4096 + p.lane * 16;
    char* dst = p.ring + p.issue_slot * CHUNK_BYTES + p.wave * 4096;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + q * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + q * 1024), 16, 0, 0);
    p.issue_chunk = p.issue_chunk + 1 == NCH ? 0 : p.issue_chunk + 1;
    p.issue_slot = p.issue_slot + 1 == RING_SLOTS ? 0 : p.issue_slot + 1;
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Ring hand-over #c, executed once per chunk c by every wavefront -- by the "early" wavefronts (0-3) in the
// middle of chunk c, by the "late" ones (4-7, their SIMD partners) before they start it, so the two
// wavefronts of a SIMD run half a chunk apart and one's VALU epilogue work and LDS latencies fall under
// the other's MFMAs instead of both stalling at the same program point:
//   wait: my share of chunk c+1 has landed (<= 4 younger loads = chunk c+2 outstanding);
//   barrier: everybody's has, and everybody is past chunk c-1;  then refill that slot with chunk c+3.
// WAIT / extra (view layer only): `extra()` issues additional loads between the barrier and the ring refill, so
// they are OLDER than that refill and YOUNGER than the chunk the next hand-over waits for; that next hand-over
// therefore allows WAIT = 4 + (number of extra loads) operations to stay in flight.
// Kernels that also STORE between the hand-overs (the training trunk: activations for the backward pass) pass a larger WAIT:
// vmcnt retires in issue order, so the wait may leave in flight every operation YOUNGER than chunk c+1's loads -- chunk c+2 (4),
// the stores of the two intervals since hand-over c-2, the extra loads of the previous hand-over.  Where a code site is shared
// by layers with different store patterns (the layer loop is rolled), WAIT_ALT is used when `alt` is set (wave-uniform).
// Counting too FEW operations is safe (it waits for more than it needs).
struct NoExtra {
    __device__ __forceinline__ void operator()() const {}
};
template <int NCH, int WAIT, class Extra, int WAIT_ALT = WAIT>
__device__ __forceinline__ void pipe_handover(Pipe& p, const Extra& extra, bool alt = false) {
    if (WAIT_ALT != WAIT && alt) wait_vm<WAIT_ALT>();
    else wait_vm<WAIT>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    extra();
    pipe_issue<NCH>(p);
}

__device__ __forceinline__ half8 lds_frag(const char* base, int piece) {
    return *reinterpret_cast<const half8*>(base + piece * 1024);
}

// hi*hi + hi*lo + lo*hi into acc; FIRST: the accumulator starts from zero (no separate clear)
template <bool FIRST>
__device__ __forceinline__ void mfma3(f32x4& acc, const half8& ah, const half8& al, const half8& bh, const half8& bl) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
}

// One 32 KB chunk = 16 (tile, hi/lo) fragment pairs.  DENSE layers: one k-step, output tiles 0..NT-1 (NT < 16: the chunk's
// last pairs are padding and skipped), B = (b0h, b0l).  VIEW layer: two k-steps of 8 output tiles, B = b0 for pairs 0..7 and
// b1 for pairs 8..15.
template <int NCH, int NACC, bool VIEW, bool FIRST, int WAIT = 4, class Extra = NoExtra, int WAIT_ALT = WAIT, int NT = 16>
__device__ __forceinline__ void chunk_mfma(f32x4 (&acc)[NACC], Pipe& p, const half8& b0h, const half8& b0l,
                                           const half8& b1h, const half8& b1l, const Extra& extra = Extra(), bool alt = false) {
    if (!p.early) pipe_handover<NCH, WAIT, Extra, WAIT_ALT>(p, extra, alt);
    const char* base = p.ring + p.cons_slot * CHUNK_BYTES + p.lane * 16;
    p.cons_slot = p.cons_slot + 1 == RING_SLOTS ? 0 : p.cons_slot + 1;
    // batches of BT tiles: 2 BT ds_read_b128, then their 3 BT MFMAs.  (Reading a batch ahead buys nothing with
    // compiler-tracked LDS loads -- the compiler waits with lgkmcnt(0), i.e. for the look-ahead batch too; the other
    // wavefront of the SIMD covers the read latency.)
    constexpr int BT = DANBO_M16_BT;
#pragma unroll
    for (int b = 0; b < 16 / BT; ++b) {
        half8 ah[BT], al[BT];
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            if (BT * b + t < NT) {
                ah[t] = lds_frag(base, 2 * (BT * b + t));
                al[t] = lds_frag(base, 2 * (BT * b + t) + 1);
            }
        }
        if (BT * b == 8 && p.early) pipe_handover<NCH, WAIT, Extra, WAIT_ALT>(p, extra, alt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < BT; ++t) {
            const int T = BT * b + t;
            if (T < NT) {
                if (VIEW && T >= 8) mfma3<false>(acc[T - 8], ah[t], al[t], b1h, b1l);
                else mfma3<FIRST>(acc[T], ah[t], al[t], b0h, b0l);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const _Float16 hh = (_Float16)v[e];
        hi[e] = hh;
        lo[e] = (_Float16)(v[e] - (float)hh);
    }
}

// sum of the four lane-group partials of a sample; identical in all four groups
__device__ __forceinline__ float quad_sum(float p) {
    p += lane_xor16(p);
    p += lane_xor32(p);
    return p;
}

// p[0..3] and p[16..19] of an LDS table that is written once before the first barrier and never again, as two ds_read_b128 the
// compiler does not track, waited for here
__device__ __forceinline__ void lds_table_read2(const float* p, float4& a, float4& b) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float*)p;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64\n\ts_waitcnt lgkmcnt(0)" : "=&v"(a), "=&v"(b) : "v"(addr) : "memory");
}

// power of two that puts max |w| of a matrix into [2^13, 2^14) (1 for max == 0 / non-finite) and its exact reciprocal:
// the lo halves of small weights would otherwise be fp16 subnormals; the consumer's epilogue multiplies by the inverse
__device__ __forceinline__ void weight_pow2_scale(float maxabs, float& s, float& inv) {
    const unsigned E = (__builtin_bit_cast(unsigned, maxabs) >> 23) & 255u;
    unsigned se = (E == 0u || E == 255u) ? 127u : 267u - E;
    se = se < 1u ? 1u : (se > 253u ? 253u : se);
    s = __builtin_bit_cast(float, se << 23);
    inv = __builtin_bit_cast(float, (254u - se) << 23);
}

// 16-byte / 8-byte stores of a wavefront through ONE wave-uniform base (SGPR pair) + a 32-bit lane offset: no per-lane
// 64-bit pointers in the register-starved MLP kernels.  `off` < 4096 (the instruction's immediate).
template <int OFF>
__device__ __forceinline__ void store16_s(const void* base, unsigned lane_off, const f32x4& v) {
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3" ::"v"(lane_off), "v"(v), "s"(base), "i"(OFF) : "memory");
}

// Tile-boundary prefetch, so that no wavefront waits on HBM between two row tiles:
//   * the NEXT tile's rows (blended features h and list entries) are fetched by two LDS-DMA loads per wavefront
//     into a 1.25 KB staging area while the view layer of the current tile runs;
//   * the CURRENT tile's per-ray view constants are fetched by eight inline-asm loads one chunk before the
//     colour head needs them (inline asm: a compiler-tracked load would make the compiler wait with
//     s_waitcnt vmcnt(0), which also drains the weight ring).
// Both ride on the ring's in-order vmcnt accounting: see pipe_handover.
constexpr int STAGE_H_BYTES = 1024, STAGE_BYTES = 1024 + 256;  // per wavefront: h [16][16] floats, list [64] ints
struct TileSrc {
    const float* h;        // a.h
    const int32_t* list;   // a.list or nullptr
    const char* dummy;     // any readable 256 bytes (the packed weights)
    int next_row0;         // first row of this wavefront in the next tile
    int n;
    char* stage;           // this wavefront's staging area (wave-uniform)
};

__device__ __forceinline__ void prefetch_rows(const TileSrc& t, int lane) {
    int row0 = t.next_row0;
    asm volatile("" : "+s"(row0));  // addresses are formed here, not hoisted out of the layer loop and spilled
    const int rh = min(row0 + (lane >> 2), t.n - 1);
    const float* src_h = t.h + (size_t)rh * DANBO_H_STRIDE + 4 * (lane & 3);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_h,
                                     (__attribute__((address_space(3))) void*)t.stage, 16, 0, 0);
    const int rl = min(row0 + (lane & 15), t.n - 1);
    const void* src_l = t.list ? (const void*)(t.list + rl) : (const void*)(t.dummy + 4 * lane);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_l,
                                     (__attribute__((address_space(3))) void*)(t.stage + STAGE_H_BYTES), 4, 0, 0);
}
struct StageRows {      // pipe_handover's `extra`: the two staging loads of the next tile
    const TileSrc& t;
    int lane;
    __device__ __forceinline__ void operator()() const { prefetch_rows(t, lane); }
};
// lane * 16, re-derived where it is used (two VALU operations) instead of living in a register across the layer loop
__device__ __forceinline__ unsigned lane_off16() {
    int zero = 0;
    asm volatile("" : "+s"(zero));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero)) << 4;
}

}  // namespace danbo
