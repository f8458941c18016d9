// K3 in the 32x32x16 form: the DANBO density/colour MLP of k_mlp16.hip (same arithmetic: fp32-accurate products as hi*hi + hi*lo +
// lo*hi on fp16 MFMAs with fp32 accumulation, the same 74 weight chunks through the same LDS ring) on v_mfma_f32_32x32x16_f16 with
// ONE wavefront per SIMD and 32 samples per wavefront.  gfx950 only.
//
// Why: k_pe_mlp16 runs two wavefronts of 16 samples per SIMD; each reads every A fragment of the weight stream for itself (16 KB of
// ds_read_b128 per k-step of 32 and wavefront) and owns half a register file (acc 64 + prev 64 + fragments).  Here a wavefront owns
// the whole file -- 256 AccVGPRs = two banks of 8 result tiles [32 features x 32 samples], 256 VGPRs for the rest -- and every A
// fragment feeds twice the flops: half the LDS reads per flop, one instead of two copies of the epilogue's scalar overhead.  With no
// partner wavefront to cover its latencies the k-substep is hand-interleaved: each group of 6 MFMAs carries, between its MFMAs, the
// ds_reads of the next group's fragments and a slice of the NEXT k-substep's epilogue (AccVGPR reads of the previous layer's outputs
// from the other bank, bias fma, ReLU, hi/lo split).  The same stream carries the ring's refill (LDS-DMA pieces in the first groups of
// a chunk), the hand-over's wait + barrier (behind the first MFMA of a chunk), the next layer's first fragment (under a layer's last
// k-substep) and -- in layer 0 -- the sines and cosines of the encoding's next fragment: the matrix pipe waits for VALU work only
// at tile boundaries (colour head, prologue).  Measured: DESIGN.md section 3 "K3, round 6"; a wavefront's own timeline:
// tools/micro_mlp32.py --trace.
//
// (the view layer's accumulators: compiler-allocated operands in a0 .. a63; a[64:127] receives the tile's view constants meanwhile)
//
// Registers the assembly names itself (mlp32_regs.inc lists them as clobbers, so the compiler keeps nothing there across a block):
//   a[0:127] / a[128:255]  result banks: a layer accumulates into one and reads its input (the previous layer's result) from the other
//   v[224:239], v[240:255] A-fragment double buffer (hi0 lo0 hi1 lo1 of a two-tile group), as in mlp16_core.hpp
//   v[216:223], v[208:215] B-fragment double buffer (hi, lo) of k-substeps U even / odd
//   v[200:207] t0..t7 (epilogue values), v[192:199] bias, v[184:191] alpha weights (view layer)
// Layout: result tile T of a layer, lane (m = lane % 32, g = lane / 32), register r holds feature 32 T + 8 (r / 4) + 4 g + r % 4 of
// sample m; k-substep U of the next layer takes registers 8 (U % 2) .. + 7 of tile U / 2 = features 16 U + {0..3, 8..11} + 4 g: the
// B-fragment k-slots 8 g + e the pack kernel (k_mlp16.hip, form 32) permutes the weight columns to.
#include <cstdlib>
#include "mlp16_core.hpp"
#include "mlp32_regs.inc"

namespace danbo {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int M32_THREADS = 256;
constexpr int M32_NCH = 74;
constexpr int M32_W = 256, M32_VW = 128;
constexpr int M32_TABLE_FLOATS = 8 * M32_W + M32_W + 3 * M32_VW + 4 + 12;     // biases, alpha_w, rgb_w, (alpha_b, rgb_b), winv [9]
constexpr int M32_STAGE_H = 2048, M32_STAGE = 2048 + 256;      // per wavefront: h [32][16] floats, list [64] ints
constexpr int M32_LDS_BYTES = RING_SLOTS * CHUNK_BYTES + M32_TABLE_FLOATS * 4 + 4 * M32_STAGE;

struct Mlp32Args {
    const float* h;
    const int32_t* list;
    const int32_t* count;
    int n_cap;
    int S;
    const char* packed;
    const float* pts_b[8];
    const float* alpha_w;
    const float* alpha_b;
    const float* cview;
    const float* rgb_w;
    const float* rgb_b;
    float* raw_out;
    float* aux_out;
};

struct Pipe32 {
    const char* packed;
    char* ring;
    int issue_chunk, issue_slot, cons_slot, wave;
    // lane * 16 and the ring's LDS address + lane * 16: formed ONCE per kernel and kept (the two-wavefront kernel re-derives them where
    // they are used to save registers; here the ~ 25 instructions of address arithmetic per hand-over are issued by a wavefront that
    // has no partner, while the matrix pipe waits: 140 cycles per chunk by the wavefront trace)
    unsigned l16, ring_lane;
};

// EVERY LDS-DMA load of this kernel is an asm statement that writes M0 itself.  The groups below set M0 for the piece they carry, and
// the compiler does not see an asm statement's write to M0 ("m0" is not accepted as a clobber): with builtin loads in the kernel it
// hoisted / merged its own M0 initialisations across the groups, and the staging loads of the next tile's rows went to wherever the
// last piece had pointed M0 -- into the ring (found as wrong colours beside correct densities).  With no builtin left the compiler
// never relies on M0.
// four 1 KB pieces: wave-uniform global base + 32-bit lane offset, LDS destination `dst` (the immediate offset applies to both)
__device__ __forceinline__ void lds_dma_4k(const char* src, unsigned voff, unsigned dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                 ::"v"(voff), "s"(src), "s"(dst) : "memory");
}
// one load of 16 / 4 bytes per lane from per-lane addresses
__device__ __forceinline__ void lds_dma_lanes16(const void* lane_src, unsigned dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(lane_src), "s"(dst) : "memory");
}
__device__ __forceinline__ void lds_dma_lanes4(const void* lane_src, unsigned dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" ::"v"(lane_src), "s"(dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }

// every wavefront loads 8 of the 32 pieces of a chunk
__device__ __forceinline__ void pipe32_issue(Pipe32& p) {
    const char* src = p.packed + (size_t)p.issue_chunk * CHUNK_BYTES + p.wave * 8192;     // wave-uniform
    const unsigned dst = lds_addr(p.ring + p.issue_slot * CHUNK_BYTES + p.wave * 8192);
    const unsigned l16 = lane_off16();
    lds_dma_4k(src, l16, dst);
    lds_dma_4k(src, l16 + 4096u, dst + 4096u);
    p.issue_chunk = p.issue_chunk + 1 == M32_NCH ? 0 : p.issue_chunk + 1;
    p.issue_slot = p.issue_slot + 1 == RING_SLOTS ? 0 : p.issue_slot + 1;
}

// hand-over #c (in front of chunk c): my share of chunk c+1 has landed (<= 8 + EXTRA younger loads outstanding), barrier: everybody's
// has and everybody is past chunk c-1; refill that slot with chunk c+3.  See pipe_handover (mlp16_core.hpp) for the accounting.
template <int EXTRA, class Extra>
__device__ __forceinline__ void pipe32_handover(Pipe32& p, const Extra& extra) {
    wait_vm<8 + EXTRA>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    extra();
    pipe32_issue(p);
}

// A wavefront WITH rows spreads its 8 loads of chunk c + 3 over the 8 groups of chunk c (one LDS-DMA instruction behind each group's
// second MFMA): issued in one piece by the hand-over they cost ~ 500 cycles per chunk in which the matrix pipe has nothing to do --
// there is no partner wavefront on the SIMD to cover them.  The hand-over itself is then a wait and a barrier.
struct Dma32 {
    const char* src;      // this wavefront's 8 KB of the chunk to load (wave-uniform)
    unsigned dst;         // LDS byte address of the same 8 KB in the slot being refilled (M0)
    unsigned v0, v1;      // lane * 16, lane * 16 + 4096
};
// (the hand-over's wait and barrier sit INSIDE the chunk's first group, behind its first MFMA -- M32_SYNC below: what has to lie
// behind the barrier is the refill of the slot of chunk c - 1 (the group's LDS-DMA sites) and the prefetch of chunk c + 1's first
// fragments by the chunk's LAST group, not the reads of chunk c itself, which the hand-over in front of chunk c - 1 published)
template <int EXTRA, class Extra>
__device__ __forceinline__ Dma32 pipe32_sync(Pipe32& p, const Extra& extra) {
    extra();
    Dma32 d;
    d.src = p.packed + (size_t)p.issue_chunk * CHUNK_BYTES + p.wave * 8192;
    d.dst = lds_addr(p.ring + p.issue_slot * CHUNK_BYTES + p.wave * 8192);
    d.v0 = p.l16;
    d.v1 = d.v0 + 4096u;
    p.issue_chunk = p.issue_chunk + 1 == M32_NCH ? 0 : p.issue_chunk + 1;
    p.issue_slot = p.issue_slot + 1 == RING_SLOTS ? 0 : p.issue_slot + 1;
    return d;
}
__device__ __forceinline__ void idle_tile32(Pipe32& p) {
#pragma unroll 1
    for (int c = 0; c < M32_NCH; ++c) pipe32_handover<0>(p, NoExtra());
    p.cons_slot = (p.cons_slot + M32_NCH) % RING_SLOTS;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the hand-interleaved groups.  Operands (numbers are template constants, printed into the register names):
//   ca, cb   first AccVGPR of the group's two result tiles        ha / na  this group's / the next group's A-fragment buffer
//   bq / nq  B-fragment buffer of this / the next k-substep        pb       first AccVGPR of the next k-substep's 8 inputs
//   nb + o0  LDS address of the next group's first fragment        ba + bo  LDS address of the next k-substep's bias (aa: alpha weights)
//   w        exact inverse of the previous layer's pack scale      al       this lane's part of the density logit (view layer)
// ---------------------------------------------------------------------------------------------------------------------------
#define M32_DRAIN "s_nop 15\n\ts_nop 15\n\ts_nop 7\n\t"
#define M32_ACC_A "a[%[ca]:%[ca]+15]"
#define M32_ACC_B "a[%[cb]:%[cb]+15]"
#define M32_MF(ACC, AOFF, B, C) "v_mfma_f32_32x32x16_f16 " ACC ", v[%[ha]+" AOFF ":%[ha]+" AOFF "+3], " B ", " C "\n\t"
#define M32_BH "v[%[bq]:%[bq]+3]"
#define M32_BL "v[%[bq]+4:%[bq]+7]"
#define M32_READS                                                \
    "ds_read_b128 v[%[na]:%[na]+3], %[nb] offset:%[o0]\n\t"       \
    "ds_read_b128 v[%[na]+4:%[na]+7], %[nb] offset:%[o0]+1024\n\t" \
    "ds_read_b128 v[%[na]+8:%[na]+11], %[nb] offset:%[o0]+2048\n\t" \
    "ds_read_b128 v[%[na]+12:%[na]+15], %[nb] offset:%[o0]+3072\n\t"
// six MFMAs on two in-place accumulators (hh0 hh1 hl0 hl1 lh0 lh1: per accumulator the order hh, hl, lh of k_pe_mlp16); E0 sits in front
// of the next group's fragment reads (what it reads from LDS is older than they are), E1..E5 behind the following MFMAs
#define M32_GROUP_ON(ACC0, ACC1, BH, BL, C0, C1, READS, E0, E1, E2, E3, E4, E5) \
    M32_GROUP_DMA(ACC0, ACC1, BH, BL, C0, C1, READS, , E0, E1, E2, E3, E4, E5)
#ifdef M32_EXP_NOLGKM     // timing experiment (wrong results): what the groups wait for their fragments
#define M32_HEAD_WAIT ""
#else
#define M32_HEAD_WAIT "s_waitcnt lgkmcnt(0)\n\t"
#endif
#define M32_GROUP_DMA(ACC0, ACC1, BH, BL, C0, C1, READS, DMA, E0, E1, E2, E3, E4, E5) \
    M32_DMA_M0 M32_HEAD_WAIT                                                     \
    M32_MF(ACC0, "0", BH, C0) M32_SYNC E0 READS                                  \
    M32_MF(ACC1, "8", BH, C1) M32_DMA_SITE("0") E1                               \
    M32_MF(ACC0, "0", BL, ACC0) M32_DMA_SITE("1") E2                             \
    M32_MF(ACC1, "8", BL, ACC1) M32_DMA_SITE("2") E3                             \
    M32_MF(ACC0, "4", BH, ACC0) M32_DMA_SITE("3") E4                             \
    M32_MF(ACC1, "12", BH, ACC1) E5
// The ring refill rides on the groups: the 8 LDS-DMA pieces a wavefront owes per chunk are issued M32_DMA_PER_GROUP at a time by the
// FIRST groups of the chunk (behind MFMAs 2 .. 5 of the group, M0 -- the LDS destination -- written at the group's head; a single
// wavefront issues one instruction per 4 cycles, 8 per MFMA): early, because the hand-over in front of chunk c + 2 waits for them --
// spread evenly over the chunk the last piece was one chunk old there, and the hand-overs were 13 % of the wavefront's cycles.
// Which sites of a group are live is decided by the ASSEMBLER (.if on a template constant): one asm text for every group.
#ifndef M32_DMA_PER_GROUP
#define M32_DMA_PER_GROUP 2
#endif
#ifdef M32_EXP_NODMA      // timing experiment (wrong results)
#define M32_DMA_M0 ""
#define M32_DMA_SITE(I) ""
#else
#define M32_DMA_M0 ".if %[gon0]\n\ts_mov_b32 m0, %[gm]\n\t.endif\n\t"
#define M32_DMA_SITE(I) ".if %[gon" I "]\n\tglobal_load_lds_dwordx4 %[gv], %[gs] offset:%[gq" I "]\n\t.endif\n\t"
#endif
// group Q (0 .. 7) of a chunk carries pieces P Q .. P Q + P - 1 (P = M32_DMA_PER_GROUP) if those exist
#define M32_DMA_PIECE(I) (M32_DMA_PER_GROUP * Q + (I))
#define M32_DMA_ON(I) (((I) < M32_DMA_PER_GROUP && M32_DMA_PIECE(I) < 8) ? 1 : 0)
// the hand-over of chunk c, in the chunk's first group (Q == 0): my share of chunk c + 1 has landed (<= VMW younger loads outstanding),
// everybody's has and everybody is past chunk c - 1
#define M32_SYNC ".if %[syn]\n\ts_waitcnt vmcnt(%[vmw])\n\ts_barrier\n\t.endif\n\t"
#define M32_DMA_OPERANDS                                                                                                      \
    [syn] "n"(Q == 0 ? 1 : 0), [vmw] "n"(VMW),                                                                                \
    [gm] "s"(M32_DMA_PIECE(0) >= 4 ? d.dst + 4096u : d.dst), [gv] "v"(M32_DMA_PIECE(0) >= 4 ? d.v1 : d.v0), [gs] "s"(d.src),    \
    [gon0] "n"(M32_DMA_ON(0)), [gon1] "n"(M32_DMA_ON(1)), [gon2] "n"(M32_DMA_ON(2)), [gon3] "n"(M32_DMA_ON(3)),                  \
    [gq0] "n"((M32_DMA_PIECE(0) & 3) * 1024), [gq1] "n"((M32_DMA_PIECE(1) & 3) * 1024), [gq2] "n"((M32_DMA_PIECE(2) & 3) * 1024), \
    [gq3] "n"((M32_DMA_PIECE(3) & 3) * 1024)
#define M32_GROUP(BH, BL, C0, C1, READS, E0, E1, E2, E3, E4, E5) \
    M32_GROUP_ON(M32_ACC_A, M32_ACC_B, BH, BL, C0, C1, READS, E0, E1, E2, E3, E4, E5)

// epilogue pieces: value e of the next k-substep lives in t_e = v(200 + e)
#define M32_RD(E) "v_accvgpr_read_b32 v20" #E ", a[%[pb]+" #E "]\n\t"
#define M32_FMA(E) "v_fma_f32 v20" #E ", v20" #E ", %[w], v19" M32_BIAS_##E "\n\t"
#define M32_BIAS_0 "2"
#define M32_BIAS_1 "3"
#define M32_BIAS_2 "4"
#define M32_BIAS_3 "5"
#define M32_BIAS_4 "6"
#define M32_BIAS_5 "7"
#define M32_BIAS_6 "8"
#define M32_BIAS_7 "9"
#define M32_MAX(E) "v_max_f32 v20" #E ", 0, v20" #E "\n\t"
#define M32_AL(E) "v_fma_f32 %[al], v20" #E ", v18" M32_AW_##E ", %[al]\n\t"
#define M32_AW_0 "4"
#define M32_AW_1 "5"
#define M32_AW_2 "6"
#define M32_AW_3 "7"
#define M32_AW_4 "8"
#define M32_AW_5 "9"
/* (values 6 and 7: v190, v191 -- written out, "v18" "10" would name v1810) */
#define M32_AL6 "v_fma_f32 %[al], v206, v190, %[al]\n\t"
#define M32_AL7 "v_fma_f32 %[al], v207, v191, %[al]\n\t"
// hi/lo split of the pair (t_2P, t_2P+1) into dword P of the next B fragments (split8_mix, common.hpp)
#define M32_CVT(P, E0_, E1_) "v_cvt_pk_f16_f32 v[%[nq]+" #P "], v20" #E0_ ", v20" #E1_ "\n\t"
#define M32_MIX(P, E0_, E1_)                                                                                                  \
    "v_fma_mixlo_f16 v[%[nq]+4+" #P "], v20" #E0_ ", 1.0, -v[%[nq]+" #P "] op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"               \
    "v_fma_mixhi_f16 v[%[nq]+4+" #P "], v20" #E1_ ", 1.0, -v[%[nq]+" #P "] op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
#define M32_BIAS_READS                                                  \
    "ds_read_b128 v[192:195], %[ba] offset:%[bo]\n\t"                   \
    "ds_read_b128 v[196:199], %[ba] offset:%[bo]+32\n\t"
#define M32_ALPHA_READS                                                 \
    "ds_read_b128 v[184:187], %[aa] offset:%[bo]\n\t"                   \
    "ds_read_b128 v[188:191], %[aa] offset:%[bo]+32\n\t"

#define M32_ASM(TEXT)                                                                                                        \
    asm volatile(TEXT : [al] "+v"(al)                                                                                        \
                 : [ca] "n"(CA), [cb] "n"(CA + 16), [ha] "n"(HA), [na] "n"(HA == 224 ? 240 : 224), [bq] "n"(BQ),              \
                   [nq] "n"(BQ == 216 ? 208 : 216), [pb] "n"(PB), [nb] "v"(nb), [o0] "n"(O0), [ba] "v"(ba), [aa] "v"(aa), [bo] "n"(BO), \
                   [w] "v"(w), M32_DMA_OPERANDS                                                                              \
                 : M32_V_CLOBBERS, M32_A_CLOBBERS, "memory")
#define M32_EMIT(READS, E0, E1, E2, E3, E4, E5)                                                          \
    do {                                                                                                 \
        if constexpr (FIRST) M32_ASM(M32_GROUP(M32_BH, M32_BL, "0", "0", READS, E0, E1, E2, E3, E4, E5)); \
        else M32_ASM(M32_GROUP(M32_BH, M32_BL, M32_ACC_A, M32_ACC_B, READS, E0, E1, E2, E3, E4, E5));    \
    } while (0)

// EPI: 0 none | 1..4 = the four groups of a dense layer's k-substep | 13: 3 + a wait state behind the split (the layer's last group: the
// next MFMA reads the fragment) | 7, 8, 9: 1, 2, 13 with the density logit's 8 fma (layer 7's last k-substep: the view layer's first
// fragment).  PF: prefetch the next group's fragments.
template <int EPI, bool FIRST, bool PF, int CA, int HA, int BQ, int PB, int O0, int BO, int Q, int VMW = 8>
__device__ __forceinline__ void m32_group(float& al, unsigned nb, unsigned ba, unsigned aa, float w, const Dma32& d) {
    static_assert(PF && (M32_DMA_PER_GROUP == 2 || M32_DMA_PER_GROUP == 4), "");
    if constexpr (EPI == 0 || EPI == 4) M32_EMIT(M32_READS, "", "", "", "", "", "");
    else if constexpr (EPI == 1)
        M32_EMIT(M32_READS, M32_BIAS_READS, M32_RD(0) M32_RD(1), M32_RD(2) M32_RD(3), M32_RD(4) M32_RD(5), M32_RD(6) M32_RD(7), "");      // 6 3 2 2 2 0
    else if constexpr (EPI == 2 || EPI == 8)
        M32_EMIT(M32_READS, M32_FMA(0) M32_FMA(1), M32_FMA(2) M32_FMA(3) M32_FMA(4), M32_FMA(5) M32_FMA(6) M32_FMA(7) M32_MAX(0),
                 M32_MAX(1) M32_MAX(2) M32_MAX(3) M32_MAX(4), M32_MAX(5) M32_MAX(6) M32_MAX(7), "");                                       // 6 4 4 4 3 0
    else if constexpr (EPI == 3)
        M32_EMIT(M32_READS, M32_CVT(0, 0, 1) M32_CVT(1, 2, 3), M32_CVT(2, 4, 5) M32_CVT(3, 6, 7), M32_MIX(0, 0, 1), M32_MIX(1, 2, 3),
                 M32_MIX(2, 4, 5), M32_MIX(3, 6, 7));                                                                                       // 6 3 2 2 2 2
    else if constexpr (EPI == 13)
        M32_EMIT(M32_READS, M32_CVT(0, 0, 1) M32_CVT(1, 2, 3), M32_CVT(2, 4, 5) M32_CVT(3, 6, 7), M32_MIX(0, 0, 1), M32_MIX(1, 2, 3),
                 M32_MIX(2, 4, 5), M32_MIX(3, 6, 7) "s_nop 1\n\t");
    else if constexpr (EPI == 7)
        M32_EMIT(M32_READS, M32_BIAS_READS M32_ALPHA_READS, M32_RD(0) M32_RD(1), M32_RD(2) M32_RD(3), M32_RD(4) M32_RD(5), M32_RD(6) M32_RD(7), "");
    else {
        static_assert(EPI == 9, "");
        M32_EMIT(M32_READS, M32_AL(0) M32_AL(1), M32_AL(2) M32_AL(3) M32_AL(4), M32_AL(5) M32_AL6 M32_AL7,
                 M32_CVT(0, 0, 1) M32_CVT(1, 2, 3) M32_CVT(2, 4, 5) M32_CVT(3, 6, 7), M32_MIX(0, 0, 1) M32_MIX(1, 2, 3),
                 M32_MIX(2, 4, 5) M32_MIX(3, 6, 7) "s_nop 1\n\t");
    }
}

// The view layer's groups: their two accumulators are OPERANDS (compiler variables in AccVGPRs), not pinned registers -- the colour
// head that follows is compiler-generated code, and this compiler takes every AccVGPR it does not know to be live for its own
// values (it read the head's colour weights into a[0:59]).  EPI 5 / 6: the two groups of a k-substep (6 also without a prefetch:
// never -- the tile's last group has no epilogue); the table reads of EPI 5 are OLDER than the fragment reads: lgkmcnt(4) = tables landed.
#define M32_VIEW_A_CLOBBERS M32_A64_CLOBBERS
template <int EPI, bool FIRST, bool PF, int HA, int BQ, int PB, int O0, int BO, int Q, int VMW = 8>
__device__ __forceinline__ void m32_view_group(f32x16& c0, f32x16& c1, float& al, unsigned nb, unsigned ba, unsigned aa, float w, const Dma32& d) {
#define M32_VIEW_ASM(TEXT, CONSTRAINT)                                                                                       \
    asm volatile(TEXT : [al] "+v"(al), [c0] CONSTRAINT(c0), [c1] CONSTRAINT(c1)                                              \
                 : [ha] "n"(HA), [na] "n"(HA == 224 ? 240 : 224), [bq] "n"(BQ), [nq] "n"(BQ == 216 ? 208 : 216), [pb] "n"(PB), \
                   [nb] "v"(nb), [o0] "n"(O0), [ba] "v"(ba), [aa] "v"(aa), [bo] "n"(BO), [w] "v"(w), M32_DMA_OPERANDS         \
                 : M32_V_CLOBBERS, M32_VIEW_A_CLOBBERS, "memory")
#define M32_VIEW_EMIT(READS, E0, E1, E2, E3, E4, E5)                                                                          \
    do {                                                                                                                     \
        if constexpr (FIRST) M32_VIEW_ASM(M32_GROUP_ON("%[c0]", "%[c1]", M32_BH, M32_BL, "0", "0", READS, E0, E1, E2, E3, E4, E5), "=&a"); \
        else M32_VIEW_ASM(M32_GROUP_ON("%[c0]", "%[c1]", M32_BH, M32_BL, "%[c0]", "%[c1]", READS, E0, E1, E2, E3, E4, E5), "+a"); \
    } while (0)
    static_assert(EPI == 0 || (PF && (EPI == 5 || EPI == 6)), "");
    if constexpr (EPI == 0 && !PF) M32_VIEW_EMIT("", "", "", "", "", "", "");
    else if constexpr (EPI == 0) M32_VIEW_EMIT(M32_READS, "", "", "", "", "", "");
    else if constexpr (EPI == 5)
        M32_VIEW_EMIT(M32_READS, M32_BIAS_READS M32_ALPHA_READS, M32_RD(0) M32_RD(1) M32_RD(2) M32_RD(3), M32_RD(4) M32_RD(5) M32_RD(6) M32_RD(7),
                      "s_waitcnt lgkmcnt(4)\n\t" M32_FMA(0) M32_FMA(1) M32_FMA(2) M32_FMA(3), M32_FMA(4) M32_FMA(5) M32_FMA(6) M32_FMA(7),
                      M32_MAX(0) M32_MAX(1) M32_MAX(2) M32_MAX(3));
    else
        M32_VIEW_EMIT(M32_READS, M32_MAX(4) M32_MAX(5) M32_MAX(6) M32_MAX(7), M32_AL(0) M32_AL(1) M32_AL(2) M32_AL(3),
                      M32_AL(4) M32_AL(5) M32_AL6 M32_AL7, M32_CVT(0, 0, 1) M32_CVT(1, 2, 3) M32_CVT(2, 4, 5) M32_CVT(3, 6, 7),
                      M32_MIX(0, 0, 1) M32_MIX(1, 2, 3), M32_MIX(2, 4, 5) M32_MIX(3, 6, 7) "s_nop 1\n\t");
#undef M32_VIEW_EMIT
#undef M32_VIEW_ASM
}

// the same group with the B fragments in compiler registers (the positional-encoding k-substeps).  EPI 1, 2, 13: the groups of layer 0's
// LAST k-substep carry the first fragment of layer 1 (from tile 0 of the bank they accumulate into: final since the k-substep's
// first group), as the last k-substep of a dense layer does
template <bool FIRST, int CA, int HA, int O0, int Q, int VMW = 8, int EPI = 0>
__device__ __forceinline__ void m32_group_pe(const half8& xh, const half8& xl, unsigned nb, const Dma32& d, unsigned ba = 0u, float w = 0.f) {
#define M32_PE_ASM(C0, C1, E0, E1, E2, E3, E4, E5)                                                                           \
    asm volatile("s_nop 1\n\t" M32_GROUP("%[xh]", "%[xl]", C0, C1, M32_READS, E0, E1, E2, E3, E4, E5)                          \
                 :: [ca] "n"(CA), [cb] "n"(CA + 16), [ha] "n"(HA), [na] "n"(HA == 224 ? 240 : 224), [xh] "v"(xh), [xl] "v"(xl),   \
                    [nb] "v"(nb), [o0] "n"(O0), [nq] "n"(216), [pb] "n"((CA / 128) * 128), [ba] "v"(ba), [bo] "n"(0), [w] "v"(w), \
                    M32_DMA_OPERANDS                                                                                         \
                 : M32_V_CLOBBERS, M32_A_CLOBBERS, "memory")
#define M32_PE_EMIT(E0, E1, E2, E3, E4, E5)                                                      \
    do {                                                                                         \
        if constexpr (FIRST) M32_PE_ASM("0", "0", E0, E1, E2, E3, E4, E5);                        \
        else M32_PE_ASM(M32_ACC_A, M32_ACC_B, E0, E1, E2, E3, E4, E5);                            \
    } while (0)
    if constexpr (EPI == 0) M32_PE_EMIT("", "", "", "", "", "");
    else if constexpr (EPI == 1) M32_PE_EMIT(M32_BIAS_READS, M32_RD(0) M32_RD(1), M32_RD(2) M32_RD(3), M32_RD(4) M32_RD(5), M32_RD(6) M32_RD(7), "");
    else if constexpr (EPI == 2)
        M32_PE_EMIT(M32_FMA(0) M32_FMA(1), M32_FMA(2) M32_FMA(3) M32_FMA(4), M32_FMA(5) M32_FMA(6) M32_FMA(7) M32_MAX(0),
                    M32_MAX(1) M32_MAX(2) M32_MAX(3) M32_MAX(4), M32_MAX(5) M32_MAX(6) M32_MAX(7), "");
    else {
        static_assert(EPI == 13, "");
        M32_PE_EMIT(M32_CVT(0, 0, 1) M32_CVT(1, 2, 3), M32_CVT(2, 4, 5) M32_CVT(3, 6, 7), M32_MIX(0, 0, 1), M32_MIX(1, 2, 3), M32_MIX(2, 4, 5),
                    M32_MIX(3, 6, 7) "s_nop 1\n\t");
    }
#undef M32_PE_EMIT
#undef M32_PE_ASM
}

// ---------------------------------------------------------------------------------------------------------------------------
// Layer 0 with the encoding's VALU work under its MFMAs.  The 48 sine / cosine pairs of a row (8 channels x 6 levels per lane)
// and the splits of the 13 B fragments are ~ 1 400 VALU instructions per tile; made in one piece in front of the layer they cost
// 3.7 % of the launch (no partner wavefront covers them).  Here fragment U + 1 is made WHILE k-substep U runs: the 24 MFMAs of a
// k-substep are single asm statements, and behind MFMA m sits slice m of the next fragment's work -- compiler-generated code, held
// in its slot by empty volatile asm statements on the values that cross the slot's borders (volatile statements keep their order,
// so the slice can move neither above the MFMA in front of it nor below the one behind it).  The arithmetic is pe_sincos' (common.hpp),
// operation for operation: the fragments are bit-identical to pe32_ksub's.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int pe_t(int U, int e) { return (8 * U + e) % 13; }
constexpr int pe_c(int U, int e) { return (8 * U + e) / 13; }
constexpr bool pe_is_job(int U, int e) { return 8 * U + e < 104 && (pe_t(U, e) & 1) != 0; }      // a sine: one sincos job (its cosine is the next value)
constexpr int pe_njobs(int U) { int n = 0; for (int e = 0; e < 8; ++e) n += pe_is_job(U, e) ? 1 : 0; return n; }
constexpr int pe_job_e(int U, int n) { int k = 0; for (int e = 0; e < 8; ++e) if (pe_is_job(U, e)) { if (k == n) return e; ++k; } return 0; }

struct PeJob { float a, k, r, r2, sp, cp, sn, cs; };
struct PeFrag {
    float v8[8];
    float cs_in, cs_out;          // the cosine of a pair whose sine is a fragment's last value: first value of the next fragment
    unsigned h[4], l[4];
};
#define M32_FENCE2(A, B) asm volatile("" : "+v"(A), "+v"(B))
#define M32_FENCE3(A, B, C) asm volatile("" : "+v"(A), "+v"(B), "+v"(C))
#define M32_FENCE5(A, B, C, D, E) asm volatile("" : "+v"(A), "+v"(B), "+v"(C), "+v"(D), "+v"(E))

template <int STAGE>
__device__ __forceinline__ void pe_job_stage(PeJob& j, float x, float scale, float& out_s, float& out_c) {
    if constexpr (STAGE == 0) {
        asm volatile("" : "+v"(x));
        j.a = x * scale;
        j.k = rintf(j.a * 0.636619772367581343f);
        j.r = fmaf(j.k, -1.57079601287841796875f, j.a);
        M32_FENCE2(j.k, j.r);
    } else if constexpr (STAGE == 1) {
        j.r = fmaf(j.k, -3.1391647326017846353352069854736328125e-7f, j.r);
        j.r = fmaf(j.k, -5.390302529957764765544681040410068817436695098876953125e-15f, j.r);
        j.r2 = j.r * j.r;
        M32_FENCE3(j.k, j.r, j.r2);
    } else if constexpr (STAGE == 2) {
        j.sp = fmaf(j.r2, 2.6083159809786593541502952575683593750e-6f, -1.981069071916863322258e-4f);
        j.sp = fmaf(j.sp, j.r2, 8.33307858556509017944e-3f);
        j.sp = fmaf(j.sp, j.r2, -1.66666597127914428711e-1f);
        j.cp = fmaf(j.r2, 2.44331571593647822737693786621e-5f, -1.38873163610696792602539062500e-3f);
        M32_FENCE5(j.k, j.r, j.r2, j.sp, j.cp);
    } else if constexpr (STAGE == 3) {
        j.sn = fmaf(j.sp * j.r2, j.r, j.r);
        j.cp = fmaf(j.cp, j.r2, 4.16666455566883087158203125e-2f);
        j.cp = fmaf(j.cp, j.r2, -0.5f);
        j.cs = fmaf(j.cp, j.r2, 1.0f);
        M32_FENCE3(j.k, j.sn, j.cs);
    } else {
        const int q = (int)j.k;
        const float s0 = (q & 1) ? j.cs : j.sn;
        const float c0 = (q & 1) ? j.sn : j.cs;
        out_s = (q & 2) ? -s0 : s0;
        out_c = ((q + 1) & 2) ? -c0 : c0;
        M32_FENCE2(out_s, out_c);
    }
}

// slice M (0 .. 23) of fragment UN: slots 5 n .. 5 n + 4 carry job n, slot 20 the values that are no job's and half of the split,
// slot 21 the other half
template <int UN, int M>
__device__ __forceinline__ void pe_slot(const float (&hv)[8], PeJob& j, PeFrag& f, half8& xh, half8& xl) {
    constexpr int NJ = pe_njobs(UN);
    static_assert(NJ <= 4, "four jobs of five slots");
    if constexpr (M < 20) {
        constexpr int n = M / 5, st = M % 5;
        if constexpr (n < NJ) {
            constexpr int e = pe_job_e(UN, n), c = pe_c(UN, e), lvl = (pe_t(UN, e) - 1) >> 1;
            if constexpr (e + 1 < 8) pe_job_stage<st>(j, hv[c], (float)(1 << lvl), f.v8[e], f.v8[e + 1]);
            else pe_job_stage<st>(j, hv[c], (float)(1 << lvl), f.v8[e], f.cs_out);
        }
    } else if constexpr (M == 20) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int jj = 8 * UN + e, t = jj % 13;
            if (jj >= 104) f.v8[e] = 0.f;
            else if (t == 0) f.v8[e] = hv[jj / 13];
            else if (e == 0 && !(t & 1)) f.v8[0] = f.cs_in;
        }
        f.cs_in = f.cs_out;
#define M32_SPLIT_PAIR(P)                                                                                                        \
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(f.h[P]) : "v"(f.v8[2 * (P)]), "v"(f.v8[2 * (P) + 1]));                   \
        asm volatile("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(f.l[P]) : "v"(f.v8[2 * (P)]), "v"(f.h[P])); \
        asm volatile("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(f.l[P]) : "v"(f.v8[2 * (P) + 1]), "v"(f.h[P]));
        M32_SPLIT_PAIR(0) M32_SPLIT_PAIR(1)
    } else if constexpr (M == 21) {
        M32_SPLIT_PAIR(2) M32_SPLIT_PAIR(3)
#undef M32_SPLIT_PAIR
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        xh = __builtin_bit_cast(half8, u32x4{f.h[0], f.h[1], f.h[2], f.h[3]});
        xl = __builtin_bit_cast(half8, u32x4{f.l[0], f.l[1], f.l[2], f.l[3]});
    }
}
// fragment UN in one piece (the tile's first)
template <int UN, int M = 0>
__device__ __forceinline__ void pe_fragment(const float (&hv)[8], PeJob& j, PeFrag& f, half8& xh, half8& xl) {
    pe_slot<UN, M>(hv, j, f, xh, xl);
    if constexpr (M + 1 < 22) pe_fragment<UN, M + 1>(hv, j, f, xh, xl);
}

// MFMA I (0 .. 5) of a group of the encoding's k-substep, as a statement of its own (m32_group_pe cut in six)
template <int I, bool FIRST, int CA, int HA, int O0, int Q, int VMW = 8>
__device__ __forceinline__ void m32_pe_mfma(const half8& xh, const half8& xl, unsigned nb, const Dma32& d) {
#define M32_PE1_ASM(TEXT)                                                                                                    \
    asm volatile(TEXT :: [ca] "n"(CA), [cb] "n"(CA + 16), [ha] "n"(HA), [na] "n"(HA == 224 ? 240 : 224), [xh] "v"(xh), [xl] "v"(xl), \
                 [nb] "v"(nb), [o0] "n"(O0), M32_DMA_OPERANDS : M32_V_CLOBBERS, M32_A_CLOBBERS, "memory")
    if constexpr (I == 0) {
        if constexpr (FIRST) M32_PE1_ASM("s_nop 1\n\t" M32_DMA_M0 M32_HEAD_WAIT M32_MF(M32_ACC_A, "0", "%[xh]", "0") M32_SYNC M32_READS);
        else M32_PE1_ASM("s_nop 1\n\t" M32_DMA_M0 M32_HEAD_WAIT M32_MF(M32_ACC_A, "0", "%[xh]", M32_ACC_A) M32_SYNC M32_READS);
    } else if constexpr (I == 1) {
        if constexpr (FIRST) M32_PE1_ASM(M32_MF(M32_ACC_B, "8", "%[xh]", "0") M32_DMA_SITE("0"));
        else M32_PE1_ASM(M32_MF(M32_ACC_B, "8", "%[xh]", M32_ACC_B) M32_DMA_SITE("0"));
    } else if constexpr (I == 2) M32_PE1_ASM(M32_MF(M32_ACC_A, "0", "%[xl]", M32_ACC_A) M32_DMA_SITE("1"));
    else if constexpr (I == 3) M32_PE1_ASM(M32_MF(M32_ACC_B, "8", "%[xl]", M32_ACC_B) M32_DMA_SITE("2"));
    else if constexpr (I == 4) M32_PE1_ASM(M32_MF(M32_ACC_A, "4", "%[xh]", M32_ACC_A) M32_DMA_SITE("3"));
    else M32_PE1_ASM(M32_MF(M32_ACC_B, "12", "%[xh]", M32_ACC_B));
#undef M32_PE1_ASM
}

// k-substep U of layer 0 (bank 0) with fragment U + 1 made under it: MFMA m, slice m, MFMA m + 1, ...
template <int U, int M = 0>
__device__ __forceinline__ void m32_pe_ksub_sliced(half8 (&xh)[13], half8 (&xl)[13], const float (&hv)[8], PeJob& j, PeFrag& f,
                                                   unsigned cbase, unsigned nbase, const Dma32& d) {
    constexpr int G = M / 6, I = M % 6, H = U & 1;
    constexpr int O0 = G < 3 ? (H * 4 + G + 1) * 4096 : (H ? 0 : 4 * 4096);
    m32_pe_mfma<I, U == 0, 32 * G, (G & 1) ? 240 : 224, O0, H * 4 + G>(xh[U], xl[U], (G == 3 && H) ? nbase : cbase, d);
    pe_slot<U + 1, M>(hv, j, f, xh[U + 1], xl[U + 1]);
    if constexpr (M + 1 < 24) m32_pe_ksub_sliced<U, M + 1>(xh, xl, hv, j, f, cbase, nbase, d);
}

// the first group's fragments of the chunk at `base` (LDS address, lane * 16 included) -> buffer 0
__device__ __forceinline__ void m32_prefetch0(unsigned base) {
    asm volatile("ds_read_b128 v[224:227], %0\n\tds_read_b128 v[228:231], %0 offset:1024\n\t"
                 "ds_read_b128 v[232:235], %0 offset:2048\n\tds_read_b128 v[236:239], %0 offset:3072" ::"v"(base)
                 : M32_V_CLOBBERS, "memory");
}

// one float of a table that is written before the first barrier and never again, by a read the compiler does not track (a tracked
// read -- or a global load, which the "memory" clobbers of the groups would make it repeat per group -- waits with vmcnt(0): the ring)
__device__ __forceinline__ float m32_lds_f32(unsigned addr) {
    float x;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(addr) : "memory");
    return x;
}

// ---------------------------------------------------------------------------------------------------------------------------
// k-substeps and layers
// ---------------------------------------------------------------------------------------------------------------------------
// dense layer, k-substep U (chunk U / 2, half U % 2): 4 groups; the epilogue of k-substep U + 1 rides on them.  On the LAST one (U = 15)
// rides the first fragment of the NEXT layer (NEXT 1; 2: of the view layer, with the density logit): it is made from tile 0 of the
// bank this layer accumulates into, which is final since this k-substep's first group -- groups 1 .. 3 carry the three epilogue
// stages with the next layer's bias row `ban` and scale `wn`.  No layer ever ends with a drained matrix pipe (the stand-alone layer end
// -- drain, 40 instructions, LDS latency -- was 900 cycles, nine times per tile: 5.5 % of it, by the wavefront trace).
template <int BANK, int U, bool FIRST, int NEXT = 1>
__device__ __forceinline__ void m32_dense_ksub(float& al, unsigned cbase, unsigned nbase, unsigned ba, float w, const Dma32& d,
                                               unsigned ban = 0u, float wn = 0.f, unsigned aa = 0u) {
    constexpr int H = U & 1, BQ = H ? 208 : 216;
    if constexpr (U < 15) {
        constexpr int PB = (1 - BANK) * 128 + 8 * (U + 1), BO = 64 * (U + 1);
        m32_group<1, FIRST, true, BANK * 128 + 0, 224, BQ, PB, (H * 4 + 1) * 4096, BO, H * 4 + 0>(al, cbase, ba, ba, w, d);
        m32_group<2, FIRST, true, BANK * 128 + 32, 240, BQ, PB, (H * 4 + 2) * 4096, BO, H * 4 + 1>(al, cbase, ba, ba, w, d);
        m32_group<3, FIRST, true, BANK * 128 + 64, 224, BQ, PB, (H * 4 + 3) * 4096, BO, H * 4 + 2>(al, cbase, ba, ba, w, d);
        // the chunk's last group reads the first group of the NEXT chunk (published by the hand-over in front of this one)
        m32_group<4, FIRST, true, BANK * 128 + 96, 240, BQ, PB, H ? 0 : 4 * 4096, BO, H * 4 + 3>(al, H ? nbase : cbase, ba, ba, w, d);
    } else {
        constexpr int PB = BANK * 128;
        m32_group<0, false, true, BANK * 128 + 0, 224, BQ, PB, 5 * 4096, 0, 4>(al, cbase, ban, aa, wn, d);
        m32_group<NEXT == 2 ? 7 : 1, false, true, BANK * 128 + 32, 240, BQ, PB, 6 * 4096, 0, 5>(al, cbase, ban, aa, wn, d);
        m32_group<NEXT == 2 ? 8 : 2, false, true, BANK * 128 + 64, 224, BQ, PB, 7 * 4096, 0, 6>(al, cbase, ban, aa, wn, d);
        m32_group<NEXT == 2 ? 9 : 13, false, true, BANK * 128 + 96, 240, BQ, PB, 0, 0, 7>(al, nbase, ban, aa, wn, d);
    }
}

struct RingPos {
    unsigned cbase, nbase;
};
__device__ __forceinline__ RingPos m32_next_chunk(Pipe32& p, unsigned lds_ring) {
    RingPos r;
    r.cbase = lds_ring + (unsigned)p.cons_slot * CHUNK_BYTES;
    p.cons_slot = p.cons_slot + 1 == RING_SLOTS ? 0 : p.cons_slot + 1;
    r.nbase = lds_ring + (unsigned)p.cons_slot * CHUNK_BYTES;
    return r;
}

// 256 -> 256 layer into bank BANK from the other bank's result.  first: the accumulators start from zero (false: the skip layer,
// whose encoding part has been accumulated already).  On entry B buffer 0 holds k-substep 0 (made under the previous layer's last
// k-substep); on exit it holds the next layer's (bias row `ban`, scale `wn`; NEXT 2: the view layer's, alpha weights at `aa`).
template <int BANK, int NEXT = 1>
__device__ __forceinline__ void m32_dense_layer(Pipe32& p, float& al, unsigned ba, float w, bool first, unsigned ban, float wn, unsigned aa = 0u) {
#define M32_CHUNK(C)                                                                            \
    {                                                                                           \
        const Dma32 d = pipe32_sync<0>(p, NoExtra());                                           \
        const RingPos r = m32_next_chunk(p, p.ring_lane);                                       \
        m32_dense_ksub<BANK, 2 * (C), false>(al, r.cbase, r.nbase, ba, w, d);                   \
        m32_dense_ksub<BANK, 2 * (C) + 1, false, NEXT>(al, r.cbase, r.nbase, ba, w, d, ban, wn, aa); \
    }
    {
        const Dma32 d = pipe32_sync<0>(p, NoExtra());
        const RingPos r = m32_next_chunk(p, p.ring_lane);
        if (first) m32_dense_ksub<BANK, 0, true>(al, r.cbase, r.nbase, ba, w, d);
        else m32_dense_ksub<BANK, 0, false>(al, r.cbase, r.nbase, ba, w, d);
        m32_dense_ksub<BANK, 1, false>(al, r.cbase, r.nbase, ba, w, d);
    }
    M32_CHUNK(1) M32_CHUNK(2) M32_CHUNK(3) M32_CHUNK(4) M32_CHUNK(5) M32_CHUNK(6) M32_CHUNK(7)
#undef M32_CHUNK
}

// the 13 k-substeps of the positional encoding (7 chunks, the second half of the last one is padding) into bank BANK, from zero
template <int BANK>
__device__ __forceinline__ void m32_pe_layer(Pipe32& p, const half8 (&xh)[13], const half8 (&xl)[13]) {
#define M32_PE_KSUB(U, FIRST_, LASTOFF, LASTBASE, Q0)                                                            \
    m32_group_pe<FIRST_, BANK * 128 + 0, 224, (((U) & 1) * 4 + 1) * 4096, (Q0) + 0>(xh[U], xl[U], r.cbase, d);   \
    m32_group_pe<FIRST_, BANK * 128 + 32, 240, (((U) & 1) * 4 + 2) * 4096, (Q0) + 1>(xh[U], xl[U], r.cbase, d);  \
    m32_group_pe<FIRST_, BANK * 128 + 64, 224, (((U) & 1) * 4 + 3) * 4096, (Q0) + 2>(xh[U], xl[U], r.cbase, d);  \
    m32_group_pe<FIRST_, BANK * 128 + 96, 240, LASTOFF, (Q0) + 3>(xh[U], xl[U], LASTBASE, d);
#define M32_PE_CHUNK(C)                                                                                          \
    {                                                                                                            \
        const Dma32 d = pipe32_sync<0>(p, NoExtra());                                                            \
        const RingPos r = m32_next_chunk(p, p.ring_lane);                                                   \
        M32_PE_KSUB(2 * (C), (C) == 0, 4 * 4096, r.cbase, 0)                                                     \
        M32_PE_KSUB(2 * (C) + 1, false, 0, r.nbase, 4)                                                           \
    }
    M32_PE_CHUNK(0) M32_PE_CHUNK(1) M32_PE_CHUNK(2) M32_PE_CHUNK(3) M32_PE_CHUNK(4) M32_PE_CHUNK(5)
    {
        const Dma32 d = pipe32_sync<0>(p, NoExtra());
        const RingPos r = m32_next_chunk(p, p.ring_lane);
        M32_PE_KSUB(12, false, 0, r.nbase, 0)
    }
#undef M32_PE_CHUNK
#undef M32_PE_KSUB
}

// layer 0: the same 13 k-substeps into bank 0, each with the next fragment made under it (fragment 0: in front, in one piece)
__device__ __forceinline__ void m32_pe_layer0(Pipe32& p, half8 (&xh)[13], half8 (&xl)[13], const float (&hv)[8], PeJob& j, PeFrag& f,
                                              unsigned ba0, float w0) {
#define M32_PE0_CHUNK(C)                                                                                         \
    {                                                                                                            \
        const Dma32 d = pipe32_sync<0>(p, NoExtra());                                                            \
        const RingPos r = m32_next_chunk(p, p.ring_lane);                                                   \
        m32_pe_ksub_sliced<2 * (C)>(xh, xl, hv, j, f, r.cbase, r.nbase, d);                                      \
        m32_pe_ksub_sliced<2 * (C) + 1>(xh, xl, hv, j, f, r.cbase, r.nbase, d);                                  \
    }
    M32_PE0_CHUNK(0) M32_PE0_CHUNK(1) M32_PE0_CHUNK(2) M32_PE0_CHUNK(3) M32_PE0_CHUNK(4) M32_PE0_CHUNK(5)
#undef M32_PE0_CHUNK
    {
        const Dma32 d = pipe32_sync<0>(p, NoExtra());
        const RingPos r = m32_next_chunk(p, p.ring_lane);
        // ... and the first fragment of layer 1 under the last k-substep's groups 1 .. 3 (bias row and scale of layer 0)
        m32_group_pe<false, 0, 224, 1 * 4096, 0>(xh[12], xl[12], r.cbase, d);
        m32_group_pe<false, 32, 240, 2 * 4096, 1, 8, 1>(xh[12], xl[12], r.cbase, d, ba0, w0);
        m32_group_pe<false, 64, 224, 3 * 4096, 2, 8, 2>(xh[12], xl[12], r.cbase, d, ba0, w0);
        m32_group_pe<false, 96, 240, 0, 3, 8, 13>(xh[12], xl[12], r.nbase, d, ba0, w0);
    }
}

// view layer (256 -> 128: four result tiles `accv` from bank 1): k-substep U = 2 groups, 4 k-substeps per chunk
template <int U, int VMW = 8>
__device__ __forceinline__ void m32_view_ksub(f32x16 (&accv)[4], float& al, unsigned cbase, unsigned nbase, unsigned ba, unsigned aa, float w,
                                              const Dma32& d) {
    constexpr int UC = U & 3, BQ = (U & 1) ? 208 : 216, PB = 128 + 8 * (U + 1), BO = 64 * (U + 1);
    constexpr bool EPI = U < 15, FIRST = U == 0;
    m32_view_group<EPI ? 5 : 0, FIRST, true, 224, BQ, PB, (UC * 2 + 1) * 4096, BO, 2 * UC, VMW>(accv[0], accv[1], al, cbase, ba, aa, w, d);
    // the tile's very last group prefetches nothing: the fragment buffers are dead across the head and the next tile's prologue
    if constexpr (U == 15) m32_view_group<0, false, false, 240, BQ, PB, 0, BO, 2 * UC + 1>(accv[2], accv[3], al, cbase, ba, aa, w, d);
    else m32_view_group<6, FIRST, true, 240, BQ, PB, UC == 3 ? 0 : (UC * 2 + 2) * 4096, BO, 2 * UC + 1>(accv[2], accv[3], al, UC == 3 ? nbase : cbase, ba, aa, w, d);
}

// the 8 positional-encoding values of k-substep U held by this lane: slot j = 8 U + e = 13 c + t of the lane's channel c (kk = 8 g + c):
// t = 0: x, t = 1 + 2 l: sin(2^l x), t = 2 + 2 l: cos(2^l x); j >= 104 is padding (pe_kstep of k_mlp16.hip with 8 channels per lane)
template <int U>
__device__ __forceinline__ void pe32_ksub(const float (&hv)[8], float& cs_keep, float (&v8)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int j = 8 * U + e, c = j / 13, t = j % 13;
        float val = 0.f;
        if (j < 104) {
            if (t == 0) val = hv[c];
            else if (t & 1) {
                float sn;
                pe_sincos(hv[c] * (float)(1 << ((t - 1) >> 1)), &sn, &cs_keep);
                val = sn;
            } else val = cs_keep;
        }
        v8[e] = val;
    }
}

struct TileSrc32 {
    const float* h;
    const int32_t* list;
    const char* dummy;
    int next_row0, n;
    char* stage;
};
// the next tile's 32 rows of this wavefront (blended features, list entries) -> its staging area: three LDS-DMA loads
struct StageRows32 {
    const TileSrc32& t;
    __device__ __forceinline__ void operator()() const {
        int row0 = t.next_row0;
        asm volatile("" : "+s"(row0));
        const int lane = (int)(lane_off16() >> 4);
        const int r0 = min(row0 + (lane >> 2), t.n - 1), r1 = min(row0 + 16 + (lane >> 2), t.n - 1);
        const unsigned st = lds_addr(t.stage);
        lds_dma_lanes16(t.h + (size_t)r0 * DANBO_H_STRIDE + 4 * (lane & 3), st);
        lds_dma_lanes16(t.h + (size_t)r1 * DANBO_H_STRIDE + 4 * (lane & 3), st + 1024u);
        const int rl = min(row0 + (lane & 31), t.n - 1);
        const void* src_l = t.list ? (const void*)(t.list + rl) : (const void*)(t.dummy + 4 * lane);
        lds_dma_lanes4(src_l, st + (unsigned)M32_STAGE_H);
    }
};
// ... and THIS tile's per-ray view constants (16 x 16 bytes per lane: features 32 T + 8 Q + 4 g .. + 3 of the lane's ray) into
// a[64:127], four chunks before the colour head needs them: the result bank they land in is dead during the view layer (whose
// accumulators the compiler keeps in a0 .. a63: the view groups clobber everything above), and loads issued at the head cost their
// whole L2 / HBM latency there (2 % of the launch).  19 loads for the next hand-over's count.
struct StageView32 {
    const TileSrc32& t;
    const float* cvb;
    __device__ __forceinline__ void operator()() const {
        StageRows32{t}();
#define M32_CVL(K, OFF) "global_load_dwordx4 a[" #K ":" #K "+3], %0, off offset:" #OFF "\n\t"
        asm volatile(M32_CVL(64, 0) M32_CVL(68, 32) M32_CVL(72, 64) M32_CVL(76, 96) M32_CVL(80, 128) M32_CVL(84, 160) M32_CVL(88, 192)
                     M32_CVL(92, 224) M32_CVL(96, 256) M32_CVL(100, 288) M32_CVL(104, 320) M32_CVL(108, 352) M32_CVL(112, 384)
                     M32_CVL(116, 416) M32_CVL(120, 448) M32_CVL(124, 480)
                     ::"v"(cvb) : M32_A64_CLOBBERS, "memory");
#undef M32_CVL
    }
};

#ifdef M32_TRACE           // dev variant (tools/ab/build_variant.sh ... -DM32_TRACE): s_memtime stamps of wavefront 0 of workgroup 0, 16 per tile
__device__ long long* g_m32_trace = nullptr;
#define M32_STAMP(I) do { if (g_m32_trace && blockIdx.x == 0 && wave == 0 && rnd < 4) { const long long t_ = clock64(); if ((lane_off16() >> 4) == 0) g_m32_trace[rnd * 16 + (I)] = t_; } } while (0)
#else
#define M32_STAMP(I) do { } while (0)
#endif
__global__ __launch_bounds__(M32_THREADS, 1) void k_pe_mlp32(Mlp32Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem + RING_SLOTS * CHUNK_BYTES);  // [8][256]
    float* s_aw = s_bias + 8 * M32_W;                                           // [256]
    float* s_rgbw = s_aw + M32_W;                                               // [3][128]
    float* s_misc = s_rgbw + 3 * M32_VW;                                        // alpha_b, rgb_b[3]
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // exact inverses of the nine matrices' pack scales (trailer of the packed buffer) -> LDS
    if (tid < 9) s_misc[4 + tid] = reinterpret_cast<const float*>(a.packed + (size_t)M32_NCH * CHUNK_BYTES)[tid];
    for (int i = tid; i < 8 * M32_W; i += M32_THREADS) s_bias[i] = a.pts_b[i >> 8][i & 255];
    if (tid < M32_W) s_aw[tid] = a.alpha_w[tid];
    for (int i = tid; i < 3 * M32_VW; i += M32_THREADS) s_rgbw[i] = a.rgb_w[i];
    if (tid < 4) s_misc[tid] = tid == 0 ? a.alpha_b[0] : a.rgb_b[tid - 1];

    const int n = resolve_count(a.count, a.n_cap);
    // rounds as in k_pe_mlp16: row groups of 32 (one wavefront's samples), 4 per workgroup and full round, the partial round spread
    // over all workgroups (`gpw` groups each; wavefronts without one only keep the ring's hand-overs going)
    const int G = (n + 31) >> 5;
    const int per_round = 4 * (int)gridDim.x;
    const int full = G / per_round, rem = G - full * per_round;
    const int gpw = (rem + (int)gridDim.x - 1) / (int)gridDim.x;
    const int my_rounds = full + ((int)blockIdx.x * gpw < rem ? 1 : 0);
    if (my_rounds == 0) return;
    auto round_base = [&](int r) { return r < full ? (r * (int)gridDim.x + (int)blockIdx.x) * 4 : full * per_round + (int)blockIdx.x * gpw; };
    auto round_waves = [&](int r) { return r < full ? 4 : min(gpw, G - round_base(r)); };

    Pipe32 p;
    p.packed = a.packed; p.ring = smem; p.issue_chunk = 0; p.issue_slot = 0; p.cons_slot = 0; p.wave = wave;
    p.l16 = (unsigned)(tid & 63) << 4;
    p.ring_lane = lds_addr(smem) + p.l16;
    pipe32_issue(p);
    pipe32_issue(p);
    pipe32_issue(p);
    TileSrc32 src;
    src.h = a.h; src.list = a.list; src.dummy = a.packed; src.n = n;
    src.stage = smem + RING_SLOTS * CHUNK_BYTES + M32_TABLE_FLOATS * 4 + wave * M32_STAGE;
    src.next_row0 = (round_base(0) + wave) * 32;
    StageRows32{src}();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int rnd = 0; rnd < my_rounds; ++rnd) {
        const int grp_i = round_base(rnd) + wave;
        if (wave >= round_waves(rnd)) {
            idle_tile32(p);
            continue;
        }
        // ------------------------------------------------------------------ inputs (staged during the previous tile's view layer)
        int zero_t = 0;
        asm volatile("" : "+s"(zero_t));
        const int lane_t = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero_t));
        const int m = lane_t & 31, g_t = lane_t >> 5;
        const int row0 = grp_i * 32 + m;
        const bool row_ok = row0 < n;
        const float* sh = reinterpret_cast<const float*>(src.stage) + m * DANBO_H_STRIDE + 8 * g_t;
        const int staged_dst = reinterpret_cast<const int*>(src.stage + M32_STAGE_H)[m];
        int dst = row_ok ? (a.list ? staged_dst : row0) : -1;
        asm volatile("" : "+v"(dst));       // materialised now: the staging area is overwritten during this tile's view layer
        float hv[8];
        {
            const float4 h0 = *reinterpret_cast<const float4*>(sh), h1 = *reinterpret_cast<const float4*>(sh + 4);
            hv[0] = h0.x; hv[1] = h0.y; hv[2] = h0.z; hv[3] = h0.w; hv[4] = h1.x; hv[5] = h1.y; hv[6] = h1.z; hv[7] = h1.w;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) hv[c] = row_ok ? hv[c] : 0.f;
        if (g_t == 1) hv[7] = 0.f;                                   // channel 15 is padding
        src.next_row0 = (round_base(rnd + 1) + wave) * 32;
        // the encoding's B fragments are kept for the skip layer; fragment 0 here, fragment U + 1 under k-substep U of layer 0
        half8 xh[13], xl[13];
        PeJob pe_job;
        PeFrag pe_frag;
        pe_frag.cs_in = pe_frag.cs_out = 0.f;
        pe_fragment<0>(hv, pe_job, pe_frag, xh[0], xl[0]);
        float al = 0.f;
        f32x16 accv[4];          // the view layer's result tiles (features 32 T + 8 (r / 4) + 4 g + r % 4)
        const unsigned tab = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)s_bias;
        const unsigned g16 = (lane_off16() >> 9) << 4;        // 16 g bytes: the lane group's 4 floats inside a block of 8 features
        const unsigned winv_at = tab + (unsigned)(M32_TABLE_FLOATS - 12) * 4u;
#define M32_WINV(L_) m32_lds_f32(winv_at + 4u * (unsigned)(L_))
        // the ray of this lane's sample and the address of its view constants (dst / S formed here, per tile: not hoisted out of the tile
        // loop and spilled, not between two layers where the matrix pipe would wait for it)
        int zero_v = 0, S_ = a.S;
        asm volatile("" : "+s"(zero_v), "+s"(S_));
        const int g_v = (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero_v)) >> 5);
        const int ray = dst >= 0 ? dst / S_ : 0;
        const float* cvb = a.cview ? a.cview + (size_t)ray * M32_VW + 4 * g_v : reinterpret_cast<const float*>(a.packed);
        M32_STAMP(0);
        m32_prefetch0(p.ring_lane + (unsigned)p.cons_slot * CHUNK_BYTES);
        // layer 0: the encoding into bank 0 (and, under its last k-substep, the first fragment of layer 1)
        m32_pe_layer0(p, xh, xl, hv, pe_job, pe_frag, tab + g16, M32_WINV(0));
        M32_STAMP(1);
        M32_STAMP(2);
#pragma unroll 1
        for (int pr = 0; pr < 3; ++pr) {
            // odd layer L = 2 pr + 1: bank 1 <- bank 0 (L = 5: the encoding first); even layer L + 1: bank 0 <- bank 1; each makes the
            // next layer's first fragment under its last k-substep
            const int L = 2 * pr + 1;
            const unsigned ba_in = tab + (unsigned)(L - 1) * 1024u + g16;
            const float w_in = M32_WINV(L - 1), w_l = M32_WINV(L), w_l1 = M32_WINV(L + 1);
            if (pr == 2) m32_pe_layer<1>(p, xh, xl);
            m32_dense_layer<1>(p, al, ba_in, w_in, pr != 2, ba_in + 1024u, w_l);
            m32_dense_layer<0>(p, al, ba_in + 1024u, w_l, true, ba_in + 2048u, w_l1);
        }
        // (outside the loop: the view layer's accumulators are compiler variables, and every statement of the loop clobbers every AccVGPR)
        {
            // layer 7: bank 1 <- bank 0; under its last k-substep the view layer's first fragment and the density logit's first 16 terms
            const unsigned ba_out = tab + 7u * 1024u + g16, aa = tab + 8u * 1024u + g16;
            const float w_6 = M32_WINV(6), w_7 = M32_WINV(7);
            M32_STAMP(3);
            m32_dense_layer<1, 2>(p, al, tab + 6u * 1024u + g16, w_6, true, ba_out, w_7, aa);
            M32_STAMP(4);
            // view layer (feature_linear merged into views_linears.0): accv <- bank 1, + the density logit
            const StageView32 stage{src, cvb};
#define M32_VCHUNK(C, EXTRA_, STAGE_)                                                              \
            {                                                                                      \
                const Dma32 d = pipe32_sync<EXTRA_>(p, STAGE_);                                    \
                const RingPos r = m32_next_chunk(p, p.ring_lane);                             \
                m32_view_ksub<4 * (C), 8 + (EXTRA_)>(accv, al, r.cbase, r.nbase, ba_out, aa, w_7, d);               \
                m32_view_ksub<4 * (C) + 1>(accv, al, r.cbase, r.nbase, ba_out, aa, w_7, d);           \
                m32_view_ksub<4 * (C) + 2>(accv, al, r.cbase, r.nbase, ba_out, aa, w_7, d);           \
                m32_view_ksub<4 * (C) + 3>(accv, al, r.cbase, r.nbase, ba_out, aa, w_7, d);           \
            }
            M32_VCHUNK(0, 19, stage)           // the 19 staging loads are issued in front of this chunk's hand-over: younger than what
            M32_VCHUNK(1, 19, NoExtra())       // it and the next one wait for, older than the refills that follow
            M32_VCHUNK(2, 0, NoExtra())
            M32_STAMP(5);
            M32_VCHUNK(3, 0, NoExtra())
            M32_STAMP(6);
#undef M32_VCHUNK
        }
        asm volatile(M32_DRAIN : "+a"(accv[0]), "+a"(accv[1]), "+a"(accv[2]), "+a"(accv[3])::"memory");
#ifdef M32_EXP_NOHEAD      // timing experiment (wrong results): the colour head
        if (dst >= 0 && (lane_off16() >> 9) == 0) reinterpret_cast<float4*>(a.raw_out)[dst] = make_float4(accv[0][0], accv[1][0], accv[2][0], al + accv[3][0]);
        continue;
#endif
        // ------------------------------------------------------------------ colour head + output
        int zero_h = 0;
        asm volatile("" : "+s"(zero_h));
        const int lane_h = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero_h));
        const int g = lane_h >> 5;
        const int row = grp_i * 32 + (lane_h & 31);
        asm volatile("" : "+v"(dst));
        // the view constants, landed in a[64:127] during the view layer (the hand-overs of its last two chunks waited for them)
        f32x4 cv[16];
#define M32_CVR(T)                                                                                                              \
        asm volatile("v_accvgpr_read_b32 %0, a[64+16*" #T "+0]\n\tv_accvgpr_read_b32 %1, a[64+16*" #T "+1]\n\t"                 \
                     "v_accvgpr_read_b32 %2, a[64+16*" #T "+2]\n\tv_accvgpr_read_b32 %3, a[64+16*" #T "+3]\n\t"                 \
                     "v_accvgpr_read_b32 %4, a[64+16*" #T "+4]\n\tv_accvgpr_read_b32 %5, a[64+16*" #T "+5]\n\t"                 \
                     "v_accvgpr_read_b32 %6, a[64+16*" #T "+6]\n\tv_accvgpr_read_b32 %7, a[64+16*" #T "+7]\n\t"                 \
                     "v_accvgpr_read_b32 %8, a[64+16*" #T "+8]\n\tv_accvgpr_read_b32 %9, a[64+16*" #T "+9]\n\t"                 \
                     "v_accvgpr_read_b32 %10, a[64+16*" #T "+10]\n\tv_accvgpr_read_b32 %11, a[64+16*" #T "+11]\n\t"             \
                     "v_accvgpr_read_b32 %12, a[64+16*" #T "+12]\n\tv_accvgpr_read_b32 %13, a[64+16*" #T "+13]\n\t"             \
                     "v_accvgpr_read_b32 %14, a[64+16*" #T "+14]\n\tv_accvgpr_read_b32 %15, a[64+16*" #T "+15]"                  \
                     : "=v"(cv[4 * T][0]), "=v"(cv[4 * T][1]), "=v"(cv[4 * T][2]), "=v"(cv[4 * T][3]), "=v"(cv[4 * T + 1][0]),   \
                       "=v"(cv[4 * T + 1][1]), "=v"(cv[4 * T + 1][2]), "=v"(cv[4 * T + 1][3]), "=v"(cv[4 * T + 2][0]),           \
                       "=v"(cv[4 * T + 2][1]), "=v"(cv[4 * T + 2][2]), "=v"(cv[4 * T + 2][3]), "=v"(cv[4 * T + 3][0]),           \
                       "=v"(cv[4 * T + 3][1]), "=v"(cv[4 * T + 3][2]), "=v"(cv[4 * T + 3][3])::"memory");
        M32_CVR(0) M32_CVR(1) M32_CVR(2) M32_CVR(3)
#undef M32_CVR
        float pr_ = 0.f, pg_ = 0.f, pb_ = 0.f;
        float* aux = (a.aux_out && dst >= 0) ? a.aux_out + (size_t)row * (M32_VW + 1) + 4 * g : nullptr;
        const float winv_v = M32_WINV(8);
#undef M32_WINV
#define M32_HEAD(T, Q)                                                                                                          \
        {                                                                                                                       \
            constexpr int nn = 32 * (T) + 8 * (Q);                                                                              \
            const float a0 = accv[T][4 * (Q) + 0], a1 = accv[T][4 * (Q) + 1], a2 = accv[T][4 * (Q) + 2], a3 = accv[T][4 * (Q) + 3]; \
            f32x4 c4 = cv[4 * (T) + (Q)];                                                                                       \
            if (!a.cview) c4 = f32x4{0.f, 0.f, 0.f, 0.f};                                                                       \
            if (aux) *reinterpret_cast<float4*>(aux + nn) = make_float4(a0 * winv_v, a1 * winv_v, a2 * winv_v, a3 * winv_v);    \
            const float x0 = fmaxf(fmaf(a0, winv_v, c4[0]), 0.f), x1 = fmaxf(fmaf(a1, winv_v, c4[1]), 0.f);                     \
            const float x2 = fmaxf(fmaf(a2, winv_v, c4[2]), 0.f), x3 = fmaxf(fmaf(a3, winv_v, c4[3]), 0.f);                     \
            const float4 wr = *reinterpret_cast<const float4*>(s_rgbw + 0 * M32_VW + nn + 4 * g);                               \
            const float4 wg = *reinterpret_cast<const float4*>(s_rgbw + 1 * M32_VW + nn + 4 * g);                               \
            const float4 wb = *reinterpret_cast<const float4*>(s_rgbw + 2 * M32_VW + nn + 4 * g);                               \
            pr_ = fmaf(x0, wr.x, pr_); pr_ = fmaf(x1, wr.y, pr_); pr_ = fmaf(x2, wr.z, pr_); pr_ = fmaf(x3, wr.w, pr_);         \
            pg_ = fmaf(x0, wg.x, pg_); pg_ = fmaf(x1, wg.y, pg_); pg_ = fmaf(x2, wg.z, pg_); pg_ = fmaf(x3, wg.w, pg_);         \
            pb_ = fmaf(x0, wb.x, pb_); pb_ = fmaf(x1, wb.y, pb_); pb_ = fmaf(x2, wb.z, pb_); pb_ = fmaf(x3, wb.w, pb_);         \
        }
#define M32_HEAD4(T) M32_HEAD(T, 0) M32_HEAD(T, 1) M32_HEAD(T, 2) M32_HEAD(T, 3)
        M32_HEAD4(0) M32_HEAD4(1) M32_HEAD4(2) M32_HEAD4(3)
#undef M32_HEAD4
#undef M32_HEAD
        // the other lane group's half (lane ^ 32): v_permlane32_swap, selected by the lane group re-derived above
        auto other = [&](float x) {
            const unsigned u = __builtin_bit_cast(unsigned, x);
            const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            return __builtin_bit_cast(float, g ? r[0] : r[1]);
        };
        const float r_ = pr_ + other(pr_) + s_misc[1];
        const float g_ = pg_ + other(pg_) + s_misc[2];
        const float b_ = pb_ + other(pb_) + s_misc[3];
        const float al_ = al + other(al) + s_misc[0];
        M32_STAMP(7);
        if (g == 0 && dst >= 0) {
            reinterpret_cast<float4*>(a.raw_out)[dst] = make_float4(r_, g_, b_, al_);
            if (a.aux_out) a.aux_out[(size_t)row * (M32_VW + 1) + M32_VW] = al_;
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

}  // namespace danbo

using namespace danbo;

// danbo_pe_mlp16_fwd's contract on weights packed by danbo_mlp32_pack
#ifdef M32_TRACE
extern "C" int danbo_dev_m32_trace(long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_m32_trace), &buf, sizeof(buf)); }
#endif
extern "C" int danbo_pe_mlp32_fwd(const float* h, const int32_t* list, const int32_t* count, int n, int S,
                                   const void* packed32, const float* const* pts_b, const float* alpha_w,
                                   const float* alpha_b, const float* cview, const float* rgb_w,
                                   const float* rgb_b, float* raw_out, float* aux_out, void* stream) {
    DANBO_CHECK_ARG(n >= 0 && S > 0 && h && packed32 && pts_b && raw_out);
    if (n == 0) return 0;
    Mlp32Args a;
    a.h = h; a.list = list; a.count = count; a.n_cap = n; a.S = S; a.packed = reinterpret_cast<const char*>(packed32);
    for (int i = 0; i < 8; ++i) a.pts_b[i] = pts_b[i];
    a.alpha_w = alpha_w; a.alpha_b = alpha_b; a.cview = cview;
    a.rgb_w = rgb_w; a.rgb_b = rgb_b; a.raw_out = raw_out; a.aux_out = aux_out;
    DANBO_ENSURE_LDS(k_pe_mlp32, M32_LDS_BYTES);
    const int groups = ceil_div(n, 32);
    const int grid = groups < num_cu() ? groups : num_cu();
    hipLaunchKernelGGL(k_pe_mlp32, dim3(grid), dim3(M32_THREADS), M32_LDS_BYTES, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
