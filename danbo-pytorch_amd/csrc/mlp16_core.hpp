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

// lane * 16, re-derived where it is used (two VALU operations) instead of living in a register across the layer loop
__device__ __forceinline__ unsigned lane_off16() {
    int zero = 0;
    asm volatile("" : "+s"(zero));
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero)) << 4;
}


// every wavefront loads 4 of the 32 pieces of a chunk: ONE wave-uniform base (SGPR pair) + the lane's 32-bit offset, and the
// instruction's immediate offset -- which LDS-DMA applies to the global AND the LDS address -- for the four pieces: one M0 write and
// no 64-bit VALU address arithmetic per hand-over (round 4's form: 3 v_lshl_add_u64 + 3 M0 updates per hand-over)
template <int NCH>
__device__ __forceinline__ void pipe_issue(Pipe& p) {
    const char* src = p.packed + (size_t)p.issue_chunk * CHUNK_BYTES + p.wave * 4096;     // wave-uniform
    char* dst = p.ring + p.issue_slot * CHUNK_BYTES + p.wave * 4096;
    const char* lane_src = src + (size_t)lane_off16();
#define DANBO_PIECE(Q)                                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)lane_src,            \
                                     (__attribute__((address_space(3))) void*)dst, 16, (Q) * 1024, 0)
    DANBO_PIECE(0); DANBO_PIECE(1); DANBO_PIECE(2); DANBO_PIECE(3);
#undef DANBO_PIECE
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

// A wavefront without rows in a (partial) tile: it still owns an eighth of every chunk's LDS-DMA loads and takes part in every
// hand-over barrier -- 74 hand-overs, nothing else (no fragment reads, no MFMAs, no stores: WAIT = 4 counts its ring loads only).
template <int NCH>
__device__ __forceinline__ void idle_tile(Pipe& p) {
#pragma unroll 1
    for (int c = 0; c < NCH; ++c) pipe_handover<NCH, 4, NoExtra>(p, NoExtra());
    p.cons_slot = (p.cons_slot + NCH) % RING_SLOTS;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Round 5: the chunk as hand-scheduled inline asm (chunk_mfma2).  What the compiler made of chunk_mfma (ISA of round 4, one
// 2-tile batch): 4 ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 6 MFMAs, each the accumulator successor of the one before and every
// result landing in the register of its A operand -> `s_nop 5` until the last MFMA has retired before the next batch's reads may
// overwrite those registers.  A wavefront alternated between an LDS round trip with an idle matrix pipe and 6 dependent MFMAs;
// only its SIMD partner covered the gaps (matrix pipe 58 % busy, profiles/r03_pmc_sq.txt).  Here:
//   * accumulators in place (v_mfma acc, A, B, acc): no result ever lands in a fragment register;
//   * A fragments in PINNED registers v224..v255 (two buffers of one 2-tile group: hi0 lo0 hi1 lo1), written by ds_read_b128 and read
//     as the MFMAs' SrcA -- the kernels are compiled with amdgpu_num_vgpr(224), so the compiler allocates none of them (it sees
//     them as clobbers only) and a fragment in flight across compiler-generated code cannot be copied or overwritten (AccVGPRs
//     would do too, but one AccVGPR in a clobber list makes this compiler split the 256 registers 128 + 128: 164 B of spills);
//   * true double buffering: the reads of group g+1 are issued between the MFMAs of group g, the wait in front of a group's
//     MFMAs is for ITS reads only; the last group of a chunk reads the first group of the NEXT chunk (the ring hand-over that
//     publishes chunk c+1 lies inside or before chunk c for every wavefront), so a chunk starts with its fragments in registers;
//   * the two tiles of a group interleaved (hh0 hh1 hl0 hl1 lh0 lh1): no MFMA is the accumulator successor of its predecessor.
// Per accumulator the products are added in the order hh, hl, lh of k-step s as before: results are bit-identical to chunk_mfma.
// Software wait states the compiler cannot see: `s_nop 1` in front of a chunk's first MFMA (the B fragments come from VALU
// instructions), MFMA_DRAIN behind the last MFMA of a GEMM (its accumulators are read by VALU instructions next).
// ---------------------------------------------------------------------------------------------------------------------------
#define DANBO_A_CLOBBERS                                                                                                       \
    "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", \
        "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "memory"
#define DANBO_MFMA_DRAIN "s_nop 15\n\ts_nop 5\n\t"

// the four fragments (pieces 4 G .. 4 G + 3) of group G of the chunk at LDS byte address `base` (lane * 16 included) -> buffer 0
__device__ __forceinline__ void agroup_prefetch0(unsigned base) {
    asm volatile("ds_read_b128 v[224:227], %0\n\tds_read_b128 v[228:231], %0 offset:1024\n\t"
                 "ds_read_b128 v[232:235], %0 offset:2048\n\tds_read_b128 v[236:239], %0 offset:3072" ::"v"(base) : DANBO_A_CLOBBERS);
}

// One group: wait for its fragments (buffer G & 1), 6 MFMAs on two in-place accumulators, and between them the 4 reads of the
// NEXT group (G + 1 of this chunk, or group 0 of the next chunk: `nbase`, offset 0) into the other buffer.
//   H0/L0/H1/L1: this group's fragment registers; N0..N3: the other buffer's; FIRST: the accumulators start from zero;
//   HEAD: first group of a chunk (wait state behind the VALU that made the B fragments); DRAIN: last MFMA of a GEMM.
// (the next group's 4 reads behind the group's first MFMA; in front of the MFMAs or spread behind the second and fourth: the same
// 2.00 - 2.03 ms per launch, round 5)
#define DANBO_GROUP_BODY(H0, L0, H1, L1, N0, N1, N2, N3, C0, C1)                                                                \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                                                 \
    "v_mfma_f32_16x16x32_f16 %[a0], " H0 ", %[bh], " C0 "\n\t"                                                                 \
    "ds_read_b128 " N0 ", %[nb] offset:%[o0]\n\t"                                                                              \
    "ds_read_b128 " N1 ", %[nb] offset:%[o1]\n\t"                                                                              \
    "ds_read_b128 " N2 ", %[nb] offset:%[o2]\n\t"                                                                              \
    "ds_read_b128 " N3 ", %[nb] offset:%[o3]\n\t"                                                                              \
    "v_mfma_f32_16x16x32_f16 %[a1], " H1 ", %[bh], " C1 "\n\t"                                                                 \
    "v_mfma_f32_16x16x32_f16 %[a0], " H0 ", %[bl], %[a0]\n\t"                                                                  \
    "v_mfma_f32_16x16x32_f16 %[a1], " H1 ", %[bl], %[a1]\n\t"                                                                  \
    "v_mfma_f32_16x16x32_f16 %[a0], " L0 ", %[bh], %[a0]\n\t"                                                                  \
    "v_mfma_f32_16x16x32_f16 %[a1], " L1 ", %[bh], %[a1]\n\t"

template <int G, bool FIRST, bool HEAD, bool DRAIN, int NG = 8, int PAR = 0>
__device__ __forceinline__ void group_mfma(f32x4& a0, f32x4& a1, const half8& bh, const half8& bl, unsigned cbase, unsigned nbase) {
    constexpr int NO = G == NG - 1 ? 0 : (G + 1) * 4096;      // byte offset of the next group's first fragment
    const unsigned nb = G == NG - 1 ? nbase : cbase;
#define DANBO_GROUP_ASM(BODY, ACC_CONSTRAINT)                                                                                   \
    asm volatile(BODY : [a0] ACC_CONSTRAINT(a0), [a1] ACC_CONSTRAINT(a1)                                                        \
                 : [bh] "v"(bh), [bl] "v"(bl), [nb] "v"(nb), [o0] "n"(NO), [o1] "n"(NO + 1024), [o2] "n"(NO + 2048), [o3] "n"(NO + 3072) \
                 : DANBO_A_CLOBBERS)
#define DANBO_GROUP_VARIANT(PRE, POST)                                                                                          \
    if (((G + PAR) & 1) == 0) {                                                                                                \
        if (FIRST) DANBO_GROUP_ASM(PRE DANBO_GROUP_BODY("v[224:227]", "v[228:231]", "v[232:235]", "v[236:239]", "v[240:243]", "v[244:247]", "v[248:251]", "v[252:255]", "0", "0") POST, "=&v"); \
        else DANBO_GROUP_ASM(PRE DANBO_GROUP_BODY("v[224:227]", "v[228:231]", "v[232:235]", "v[236:239]", "v[240:243]", "v[244:247]", "v[248:251]", "v[252:255]", "%[a0]", "%[a1]") POST, "+v"); \
    } else {                                                                                                                   \
        if (FIRST) DANBO_GROUP_ASM(PRE DANBO_GROUP_BODY("v[240:243]", "v[244:247]", "v[248:251]", "v[252:255]", "v[224:227]", "v[228:231]", "v[232:235]", "v[236:239]", "0", "0") POST, "=&v"); \
        else DANBO_GROUP_ASM(PRE DANBO_GROUP_BODY("v[240:243]", "v[244:247]", "v[248:251]", "v[252:255]", "v[224:227]", "v[228:231]", "v[232:235]", "v[236:239]", "%[a0]", "%[a1]") POST, "+v"); \
    }
    if (HEAD) { DANBO_GROUP_VARIANT("s_nop 1\n\t", "") }
    else if (DRAIN) { DANBO_GROUP_VARIANT("", DANBO_MFMA_DRAIN) }
    else { DANBO_GROUP_VARIANT("", "") }
#undef DANBO_GROUP_VARIANT
#undef DANBO_GROUP_ASM
}

// LDS byte address of the ring (dynamic shared memory starts at the workgroup's LDS base) + lane * 16, re-derived where it is used
__device__ __forceinline__ unsigned ring_lane_addr() {
    extern __shared__ __attribute__((aligned(16))) char smem_ring[];
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem_ring + lane_off16();
}

// chunk_mfma with the hand-scheduled groups.  `lds_ring`: LDS byte address of the ring + lane * 16 (a VGPR).  On entry buffer 0
// holds (or is receiving) group 0 of this chunk; on exit, group 0 of the next one.  LAST: the GEMM's last chunk.
template <int NCH, int NACC, bool VIEW, bool FIRST, bool LAST, int WAIT = 4, class Extra = NoExtra, int WAIT_ALT = WAIT, int NT = 16, int PAR = 0>
__device__ __forceinline__ void chunk_mfma2(f32x4 (&acc)[NACC], Pipe& p, unsigned lds_ring, const half8& b0h, const half8& b0l,
                                            const half8& b1h, const half8& b1l, const Extra& extra = Extra(), bool alt = false) {
    static_assert(NACC == (VIEW ? 8 : 16), "dense layers: 16 output tiles; view layer: 2 k-steps of 8");
    // NT < 16 (the training backward's PE-ordered outputs: 13 tiles): the chunk's last pairs are padding; a group both of whose
    // tiles are padding is skipped (the group before it then reads the next chunk's first fragments)
    // PAR: which fragment buffer holds group 0 on entry.  A chunk of 8 groups leaves the next chunk's group 0 where it found its
    // own (buffer PAR); a chunk of 7 groups in the OTHER buffer: the caller alternates PAR over such chunks (an even number of them
    // in a row, so that the GEMM behind them starts at PAR = 0 again).
    constexpr int NG = (NT + 1) / 2;
    static_assert(NG == 8 || (NG == 7 && !VIEW), "16 tiles, or 13 / 14 in a dense layer");
    if (!p.early) pipe_handover<NCH, WAIT, Extra, WAIT_ALT>(p, extra, alt);
    const unsigned cbase = lds_ring + (unsigned)p.cons_slot * CHUNK_BYTES;
    p.cons_slot = p.cons_slot + 1 == RING_SLOTS ? 0 : p.cons_slot + 1;
    const unsigned nbase = lds_ring + (unsigned)p.cons_slot * CHUNK_BYTES;
    group_mfma<0, FIRST, true, false, NG, PAR>(acc[0], acc[1], b0h, b0l, cbase, nbase);
    group_mfma<1, FIRST, false, false, NG, PAR>(acc[2], acc[3], b0h, b0l, cbase, nbase);
    group_mfma<2, FIRST, false, false, NG, PAR>(acc[4], acc[5], b0h, b0l, cbase, nbase);
    group_mfma<3, FIRST, false, false, NG, PAR>(acc[6], acc[7], b0h, b0l, cbase, nbase);
    if (p.early) pipe_handover<NCH, WAIT, Extra, WAIT_ALT>(p, extra, alt);
    if (VIEW) {
        group_mfma<4, false, false, false, 8, PAR>(acc[0], acc[1], b1h, b1l, cbase, nbase);
        group_mfma<5, false, false, false, 8, PAR>(acc[2], acc[3], b1h, b1l, cbase, nbase);
        group_mfma<6, false, false, false, 8, PAR>(acc[4], acc[5], b1h, b1l, cbase, nbase);
        group_mfma<7, false, false, LAST, 8, PAR>(acc[6], acc[7], b1h, b1l, cbase, nbase);
    } else {
        group_mfma<4, FIRST, false, false, NG, PAR>(acc[NACC == 16 ? 8 : 0], acc[NACC == 16 ? 9 : 1], b0h, b0l, cbase, nbase);
        group_mfma<5, FIRST, false, false, NG, PAR>(acc[NACC == 16 ? 10 : 2], acc[NACC == 16 ? 11 : 3], b0h, b0l, cbase, nbase);
        if (NG == 8) {
            group_mfma<6, FIRST, false, false, NG, PAR>(acc[NACC == 16 ? 12 : 4], acc[NACC == 16 ? 13 : 5], b0h, b0l, cbase, nbase);
            group_mfma<7, FIRST, false, LAST, NG, PAR>(acc[NACC == 16 ? 14 : 6], acc[NACC == 16 ? 15 : 7], b0h, b0l, cbase, nbase);
        } else {
            group_mfma<6, FIRST, false, LAST, NG, PAR>(acc[NACC == 16 ? 12 : 4], acc[NACC == 16 ? 13 : 5], b0h, b0l, cbase, nbase);
            if (FIRST) acc[14] = acc[15] = f32x4{0.f, 0.f, 0.f, 0.f};      // defined values (never read)
        }
    }
}

__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo) { split8_mix(v, hi, lo); }

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
    // + 2 wait states: a store of more than 8 bytes reads its data registers AFTER issue, and a VALU instruction that overwrites one
    // of them within 2 wait states changes what is stored (gfx940+; the compiler inserts these for its own stores, it cannot see
    // that this statement is one: round 5 met `v_mul v147, ...` straight behind `global_store_dwordx4 ..., v[146:149]`)
    asm volatile("global_store_dwordx4 %0, %1, %2 offset:%3\n\ts_nop 1" ::"v"(lane_off), "v"(v), "s"(base), "i"(OFF) : "memory");
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

__device__ __forceinline__ void prefetch_rows(const TileSrc& t, int /*lane*/) {
    int row0 = t.next_row0;
    asm volatile("" : "+s"(row0));  // addresses are formed here, not hoisted out of the layer loop and spilled
    const int lane = (int)(lane_off16() >> 4);      // re-derived for the same reason
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
}  // namespace danbo
