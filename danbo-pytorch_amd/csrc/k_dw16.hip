// Weight / bias gradients of the dense layers of the training step, all layers in ONE launch:
//     dW_l [N, K] = dz_l^T [N x rows] * x_l [rows x K],      db_l [N] = sum_rows dz_l
// (reference: what loss.backward() computes for every nn.Linear of core/networks/nerf.py:176-209, core/trainer.py:563-576),
// with fp32-accurate products on the fp16 matrix cores (hi/lo split as in k_linear16.hip, fp32 accumulate).  gfx950 only.
//
// The reduction index of this GEMM is the ROW of both operands, i.e. the slow index of both row-major buffers, while an MFMA
// fragment wants 8 consecutive reduction indices per lane.  So 32 rows of a tile's 128 gradient columns and 256 input columns
// go through LDS *transposed*: a thread loads the same 4 columns of two consecutive rows (2 x 16 B), splits them into fp16
// hi / lo and writes row PAIRS as 32-bit words at [column][row] (column stride 72 B, see DW_CSTRIDE).
// The workgroup is SPECIALISED: wavefronts 4-7 are producers (global loads into two register sets, i.e. two 32-row steps
// ahead -- one load instruction = 128 contiguous bytes of 8 rows --, hi/lo split in three instructions per value pair,
// transposed LDS stores, bias-gradient column sums), wavefronts 0-3 -- one per SIMD -- are consumers
// (each owns 64 x 128 of the 128 x 256 output tile = 32 accumulator tiles; 24 fragment reads feed 96 MFMAs per step).  The
// converted tile is double-buffered in LDS, one barrier per step: the producers fill buffer (s + 1) & 1 while the consumers
// multiply out of buffer s & 1, so HBM latency, the VALU conversion and the MFMAs overlap inside ONE workgroup per CU.  (The
// unspecialised forms -- every wavefront loads, converts, then multiplies: 256 x 256 tiles at one workgroup per CU, 128 x 256
// at two, 128 x 128 at three -- all measured 0.49-0.52 ms: their loads were only in flight during the MFMA phase.  This form:
// 0.39 ms, with the MFMAs compiled out the same -- the producers set the pace, at 3.4 TB/s of HBM reads, 1.33 GB per launch by
// the FETCH_SIZE counter against 1.0 GB if every operand row were read once.)
// The two 128-column halves of a layer's gradient read the same input rows: they are neighbours in the work-item order and
// blockIdx is permuted so that both run on the same XCD at the same time -- the second read of the inputs is an L2 hit.
// Rows are split over `slices` workgroups per tile; partial tiles go to a scratch buffer and a second kernel sums the
// slices in a fixed order (deterministic, no float atomics), undoes the power-of-two pre-scale of the gradient operand
// (in_maxabs of danbo_linear16_ex) and writes the gradients in nn.Linear layout.
// Algorithmic traffic per row and layer: 4 (N + K) bytes; flops 2 N K (x 3 MFMA products).
#include <type_traits>
#include "common.hpp"

namespace danbo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ half8 dw_frag(const char* p) {   // 16 bytes at an 8-byte aligned LDS address
    const half4 a = *reinterpret_cast<const half4*>(p), b = *reinterpret_cast<const half4*>(p + 8);
    return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
}

constexpr int DW_TN = 128, DW_TK = 256, DW_ROWS = 32, DW_THREADS = 512, DW_PRODUCERS = 256;
// bytes per column in LDS: 32 rows x 2 B + 8 B padding.  18 words: the 32-bit transposed stores of a producer wavefront (8 row pairs
// x 8 groups of 4 columns) fall into 32 different banks per half-wave, and the 16 columns a fragment read touches start in 16
// different even banks; the price is 8-byte alignment, i.e. fragments are read as two 8-byte halves
constexpr int DW_CSTRIDE = 72;
constexpr int DW_A_BYTES = DW_TN * DW_CSTRIDE;           // one of hi / lo
constexpr int DW_B_BYTES = DW_TK * DW_CSTRIDE;
constexpr int DW_BUF_BYTES = 2 * DW_A_BYTES + 2 * DW_B_BYTES;               // one converted 32-row step
constexpr int DW_LDS_BYTES = 2 * DW_BUF_BYTES + DW_TN * 4;                  // double-buffered + column sums
constexpr int DW_MAX_LAYERS = 13;

struct DwLayer {
    const float* dy;       // [rows, ldy] gradient with respect to the layer's pre-activation
    const float* x1;       // [rows, ld1]
    const float* x2;       // [rows, ld2] or nullptr
    const float* dy_maxabs;  // device scalar: max |dy| (power-of-two pre-scale), or nullptr
    float* gw;             // nn.Linear.weight.grad [N, K1 + K2] rows < split_n ...
    float* gw2;            // ... and rows >= split_n (feature_linear | alpha_linear evaluated as one layer), or nullptr
    float* gb;
    float* gb2;
    int ldy, ld1, ld2, N, K1, K2, split_n;
    int frag;              // bit 0: dy, 1: x1 (then K2 == 0) in the fragment order of k_linear16 ([rows/16][C/32][2][4 q][16 n][4])
    int gw_ld, gw_col0;    // the gradient goes to gw[n * gw_ld + gw_col0 + k] (a layer whose inputs are handled as two layers)
    int x1_pe;             // x1 = the fused trunk's encoding buffer: slot k -> column pe_slot_column(k), padding slots dropped
    int K1p, Kv;           // K1 rounded up to 4; virtual width K1p + K2
    int tile0, nt_k;       // first tile of this layer, tiles along K
    long part_off;         // floats: this layer's [slices][N * Kv + N] partial block
};

struct DwArgs {
    DwLayer l[DW_MAX_LAYERS];
    int n_layers, n_tiles, slices, M;
    const int32_t* count;  // device row count (or nullptr: M)
    float* part;
};

__device__ __forceinline__ int dw_rows_per_slice(int M, int slices) {
    const int r = (M + slices - 1) / slices;
    return (r + DW_ROWS - 1) / DW_ROWS * DW_ROWS;
}

__device__ __forceinline__ void dw_pow2_scale(float maxabs, float& s, float& inv) {
    const unsigned E = (__builtin_bit_cast(unsigned, maxabs) >> 23) & 255u;
    unsigned se = (E == 0u || E == 255u) ? 127u : 257u - E;    // max * s in [8, 16)
    se = se < 1u ? 1u : (se > 253u ? 253u : se);
    s = __builtin_bit_cast(float, se << 23);
    inv = __builtin_bit_cast(float, (254u - se) << 23);
}

// (a, b) -> hi = {fp16(a), fp16(b)}, lo = {fp16(a - hi.x), fp16(b - hi.y)} in three instructions: one packed convert and two
// mixed-precision fmas that subtract the fp16 half from the fp32 value and round once (a - fp16(a) is exact in fp32, so this is
// bit for bit the four-instruction form (_Float16)(a - (float)(_Float16)a))
__device__ __forceinline__ void dw_split2(float a, float b, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_fma_mixlo_f16 %1, %2, 1.0, -%0 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(hi), "=&v"(lo)
        : "v"(a), "v"(b));
}

// two rows x 4 columns -> hi / lo halves, stored as row pairs
template <bool SCALE>
__device__ __forceinline__ void dw_store4(char* hi_base, char* lo_base, int col, int rp, const f32x4& r0, const f32x4& r1, float sc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h, l;
        dw_split2(SCALE ? r0[j] * sc : r0[j], SCALE ? r1[j] * sc : r1[j], h, l);
        *reinterpret_cast<unsigned*>(hi_base + (col + j) * DW_CSTRIDE + rp * 4) = h;
        *reinterpret_cast<unsigned*>(lo_base + (col + j) * DW_CSTRIDE + rp * 4) = l;
    }
}

// The producer half of k_dw16 for one combination of operand layouts (compile-time: with run-time layout tests in the load /
// convert loops the compiler serialises the loads of a step behind vmcnt(0) / vmcnt(1) waits).
struct DwCtx {
    char* smem; float* s_db;
    int tid, lane, row0, row_end, nsteps, n0, v0, tk;
    float sc;
};
__device__ __forceinline__ void dw_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <int T> __device__ __forceinline__ void dw_issue(const char* const (&pa)[2][2], const char* const (&pb)[4][2]);
template <int T> __device__ __forceinline__ void dw_take(f32x4 (&a)[2][2], f32x4 (&bb)[4][2]);
#include "k_dw16_regs.inc"
template <int N> __device__ __forceinline__ void dw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool FA, bool FB>
__device__ __forceinline__ void dw_producer(const DwLayer& L, const DwCtx& c) {
    char* const smem = c.smem;
    float* const s_db = c.s_db;
    const int tid = c.tid, lane = c.lane, row0 = c.row0, row_end = c.row_end, nsteps = c.nsteps, n0 = c.n0, v0 = c.v0, tk = c.tk;
    const float sc = c.sc;
    auto wg_sync = []() { dw_sync(); };
    {
        // ---- producers.  A thread owns row PAIR p of the 32-row step -- rows p and p + 16, one 32-bit word of hi halves and
        // one of lo halves per column -- and a few groups of 4 columns.  Which (p, columns) a lane gets depends on the
        // operand's layout, so that one load instruction of a wavefront is as contiguous as the layout allows:
        //   row-major      : 8 row pairs x 8 column groups  -> 128 contiguous bytes of 8 consecutive rows
        //                    (with 16 pairs x 4 groups it was sixteen 64-byte pieces: 0.43 -> 0.39 ms)
        //   fragment order : 16 row pairs x 4 column groups -> ONE contiguous KB, the [4 q][16 n][4] block of a k-step half
        // Any pairing works as long as both operands use the same one: the sum over rows does not care.
        const int ptid = tid & (DW_PRODUCERS - 1);
        const int pw = ptid >> 6;
        constexpr bool fragA = FA, fragB = FB;
        const int pA = fragA ? (lane & 15) : (lane & 7) + 8 * (pw & 1);
        const int pB = fragB ? (lane & 15) : (lane & 7) + 8 * (pw & 1);
        int colA[2], colB[4];                         // first of the thread's 4 columns, relative to the tile
#pragma unroll
        for (int u = 0; u < 2; ++u) colA[u] = fragA ? 16 * (pw + 4 * u) + 4 * (lane >> 4) : 4 * ((lane >> 3) + 8 * (pw >> 1) + 16 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) colB[u] = fragB ? 16 * (pw + 4 * u) + 4 * (lane >> 4) : 4 * ((lane >> 3) + 8 * (pw >> 1) + 16 * u);
        struct Stage { f32x4 a[2][2], b[4][2]; };    // [group][row of the pair]
        // Every load is issued for every thread in every step (past the last column: the group at column 0 -- no branch, no
        // zero-initialised alternative, so two steps of loads stay in flight); what must not count is zeroed when the values
        // are converted, and only in tiles / steps that have such columns / rows (uniform branches).  The 12 addresses are
        // 64-bit pointers advanced by 32 rows per step: steps are requested in order.
        unsigned a_live = 0, b_live = 0;                 // 4 bits per group, one per column
        float db_acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const char* pa[2][2];
        const char* pb[4][2];
        unsigned a_stride, b_stride[4];
        // address of (row, 4 columns from `col`) of a fragment-order buffer of width C
        auto frag_ptr = [&](const float* base, int C, int row, int col) {
            const long g = row >> 4;
            const int nn = row & 15, cg = col >> 2;
            return reinterpret_cast<const char*>(base + ((g * (C >> 5) + (cg >> 3)) * 2 + ((cg >> 2) & 1)) * 256 + (nn + 16 * (cg & 3)) * 4);
        };
        a_stride = fragA ? (unsigned)(L.N >> 5) * 4096u : (unsigned)L.ldy * 4u * DW_ROWS;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col = n0 + colA[u];
            const int cc = col < L.N ? col : 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) a_live |= (col + j < L.N ? 1u : 0u) << (4 * u + j);
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int row = row0 + pA + 16 * w;
                pa[u][w] = fragA ? frag_ptr(L.dy, L.N, row, cc) : reinterpret_cast<const char*>(L.dy + (long)row * L.ldy + cc);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int vc = v0 + colB[u];                 // virtual column: [x1 padded to K1p | x2]
            const bool first = vc < L.K1p || L.x2 == nullptr;
            const int c = first ? (vc < L.K1p ? vc : 0) : (vc < L.Kv ? vc - L.K1p : 0);
            const int ld = first ? L.ld1 : L.ld2;
            b_stride[u] = fragB ? (unsigned)(L.K1 >> 5) * 4096u : (unsigned)ld * 4u * DW_ROWS;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool live = first ? (vc + j < L.K1) : (vc + j < L.Kv);
                b_live |= (live ? 1u : 0u) << (4 * u + j);
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int row = row0 + pB + 16 * w;
                pb[u][w] = fragB ? frag_ptr(L.x1, L.K1, row, c)
                                 : reinterpret_cast<const char*>((first ? L.x1 : L.x2) + (long)row * ld + c);
            }
        }
        const bool cols_full = __all((a_live == 0xffu) && (b_live == 0xffffu));   // per wavefront
        int next_row = row0;                            // first row of the step the next load_step requests
        // The loads are inline asm into fixed registers (k_dw16_regs.inc) and are waited for BY COUNT: vmcnt retires in issue order,
        // every step issues the same 12 loads per lane, so "at most 12 y outstanding" means everything but the y youngest steps has
        // arrived.  (With compiler-tracked loads every convert ended in vmcnt(0) -- the requests of the steps ahead were drained
        // at every step and the kernel ran at one memory latency per step, MFMAs or not.)
        auto issue = [&](auto stage) {
            if (next_row + DW_ROWS <= row_end) {
                dw_issue<decltype(stage)::value>(pa, pb);
            } else {
                // the slice's last, partial step: rows past the end read the last row again (row-major; a fragment-order buffer is
                // padded to whole 128-row tiles and read in place); zeroed in convert
                const char* qa[2][2];
                const char* qb[4][2];
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const long backA = (long)max(next_row + pA + 16 * w - (row_end - 1), 0);
                    const long backB = (long)max(next_row + pB + 16 * w - (row_end - 1), 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) qa[u][w] = pa[u][w] - (fragA ? 0 : backA * (L.ldy * 4));
#pragma unroll
                    for (int u = 0; u < 4; ++u) qb[u][w] = pb[u][w] - (fragB ? 0 : backB * (long)(b_stride[u] / DW_ROWS));
                }
                dw_issue<decltype(stage)::value>(qa, qb);
            }
#pragma unroll
            for (int w = 0; w < 2; ++w) {
#pragma unroll
                for (int u = 0; u < 2; ++u) pa[u][w] += a_stride;
#pragma unroll
                for (int u = 0; u < 4; ++u) pb[u][w] += b_stride[u];
            }
            next_row += DW_ROWS;
        };
        // registers -> LDS buffer `buf` (transposed, split); the bias gradient on the way
        auto convert = [&](Stage& st, int buf, int r) {
            char* const base = smem + buf * DW_BUF_BYTES;
            if (!cols_full || r + DW_ROWS > row_end) {      // edge tile / last step of the slice: zero what does not exist
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const bool okA = r + pA + 16 * w < row_end, okB = r + pB + 16 * w < row_end;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) st.a[u][w][j] = okA && ((a_live >> (4 * u + j)) & 1u) ? st.a[u][w][j] : 0.f;
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) st.b[u][w][j] = okB && ((b_live >> (4 * u + j)) & 1u) ? st.b[u][w][j] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                dw_store4<true>(base, base + DW_A_BYTES, colA[u], pA, st.a[u][0], st.a[u][1], sc);
                if (tk == 0) {   // the 16 lanes of a DPP row hold the 32 rows of the same 4 columns
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x = st.a[u][0][j] + st.a[u][1][j];
                        // row-major: 8 consecutive lanes hold 16 rows of the same 4 columns: after shifts by 1, 2, 4 lane 7 of the
                        // eight has their sum (a window of 8: lanes 8-15 of a DPP row do not see lanes 0-7); fragment order: the
                        // 16 lanes of a DPP row hold all 32 rows, one more shift and lane 15 has the sum
                        DANBO_DPP_STEP(dpp_add_, 0.f, 0x111, 0xf) DANBO_DPP_STEP(dpp_add_, 0.f, 0x112, 0xf)
                        DANBO_DPP_STEP(dpp_add_, 0.f, 0x114, 0xf)
                        if (fragA) DANBO_DPP_STEP(dpp_add_, 0.f, 0x118, 0xf)
                        if (fragA ? (lane & 15) == 15 : (lane & 7) == 7) db_acc[4 * u + j] += x;   // column sums stay in registers
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                dw_store4<false>(base + 2 * DW_A_BYTES, base + 2 * DW_A_BYTES + DW_B_BYTES, colB[u], pB, st.b[u][0], st.b[u][1], 1.f);
        };
        // TWO steps in flight: step k lives in register set k & 1 and is converted into LDS buffer k & 1 while the consumers
        // multiply step k - 1; its registers are re-used for step k + 2.  1 + nsteps barriers, as the consumers.
        using S0 = std::integral_constant<int, 0>;
        using S1 = std::integral_constant<int, 1>;
        issue(S0{});
        if (nsteps > 1) issue(S1{});
        auto step = [&](auto stage, int k) {
            if (k < nsteps) {
                if (k + 1 < nsteps) dw_wait_vm<12>();                 // step k + 1 has been requested behind step k
                else dw_wait_vm<0>();
                Stage st;
                dw_take<decltype(stage)::value>(st.a, st.b);
                if (k + 2 < nsteps) issue(stage);
                convert(st, k & 1, row0 + k * DW_ROWS);
            }
            wg_sync();
        };
        for (int k = 0;;) {
            step(S0{}, k); if (++k > nsteps) break;
            step(S1{}, k); if (++k > nsteps) break;
        }
        if (tk == 0 && (fragA ? (lane & 15) == 15 : (lane & 7) == 7)) {   // row-major: two wavefronts (row pairs 0-7 / 8-15) per column
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(s_db + colA[u] + j, db_acc[4 * u + j]);
        }
    }
}

__global__ __launch_bounds__(DW_THREADS, 1) void k_dw16(DwArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_db = reinterpret_cast<float*>(smem + 2 * DW_BUF_BYTES);

    // work item: blockIdx permuted so that items 2j and 2j + 1 (the two gradient-column halves of a layer's tile) get
    // workgroup ids 8 apart, i.e. the same XCD back to back
    const int id = (int)blockIdx.x, xcd = id & 7, kk = id >> 3;
    const int item = ((((kk >> 1) << 3) + xcd) << 1) + (kk & 1);
    // The tiles of a slice are padded to an EVEN count: with the training step's 21 tiles the two halves of a layer's tile were
    // neighbours (same XCD, same moment) in every other slice only; the other half of the pairs ran on different XCDs and read the
    // shared input rows twice from HBM (1.33 GB per launch; 1.07 GB now, against 1.0 GB if every operand were read once).
    const int tiles_even = (a.n_tiles + 1) & ~1;
    if (item >= tiles_even * a.slices) return;
    const int sl = item / tiles_even, tile = item % tiles_even;
    if (tile >= a.n_tiles) return;

    const int M = resolve_count(a.count, a.M);
    const int rps = dw_rows_per_slice(M, a.slices);
    const int row0 = sl * rps;
    if (row0 >= M) return;                       // empty slice: the reduction only reads slices that exist
    const int row_end = min(row0 + rps, M);

    // which layer / tile (gradient-column tile fastest)
    int li = 0;
    while (li + 1 < a.n_layers && tile >= a.l[li + 1].tile0) ++li;
    const DwLayer& L = a.l[li];
    const int t = tile - L.tile0;
    const int nt_n = (L.N + DW_TN - 1) / DW_TN;
    const int tn = t % nt_n, tk = t / nt_n;
    const int n0 = tn * DW_TN, v0 = tk * DW_TK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = tid >= DW_THREADS - DW_PRODUCERS;
    float sc = 1.f, inv = 1.f;
    if (L.dy_maxabs != nullptr) dw_pow2_scale(*L.dy_maxabs, sc, inv);

    const int nsteps = (row_end - row0 + DW_ROWS - 1) / DW_ROWS;
    for (int i = tid; i < DW_TN; i += DW_THREADS) s_db[i] = 0.f;
    __syncthreads();
    float* part = a.part + L.part_off + (long)sl * ((long)L.N * L.Kv + L.N);
    // Both roles execute the same sequence of barriers (1 + nsteps).  They are separate regions of the kernel so that neither
    // carries the other's registers: 96 staging registers here, 128 accumulators + 40 fragment registers there.
    auto wg_sync = []() { dw_sync(); };

    if (producer) {
        const DwCtx ctx{smem, s_db, tid, lane, row0, row_end, nsteps, n0, v0, tk, sc};
        switch (L.frag & 3) {
            case 0: dw_producer<false, false>(L, ctx); break;
            case 1: dw_producer<true, false>(L, ctx); break;
            case 2: dw_producer<false, true>(L, ctx); break;
            default: dw_producer<true, true>(L, ctx); break;
        }
    } else {
        // ---- consumers: wave tile 64 gradient columns x 128 input columns (4 wavefronts: 2 x 2)
        const int wn = (wave >> 1) * 64, wk = (wave & 1) * 128;
        const int m = lane & 15, q = lane >> 4;
        const bool wave_live = (n0 + wn < L.N) && (v0 + wk < L.Kv);
        f32x4 acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto multiply = [&](int buf) {
            const char* const s_ah = smem + buf * DW_BUF_BYTES;
            const char* const s_al = s_ah + DW_A_BYTES;
            const char* const s_bh = s_ah + 2 * DW_A_BYTES;
            const char* const s_bl = s_bh + DW_B_BYTES;
            half8 ah[4], al[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = dw_frag(s_ah + (wn + 16 * i + m) * DW_CSTRIDE + q * 16);
                al[i] = dw_frag(s_al + (wn + 16 * i + m) * DW_CSTRIDE + q * 16);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const half8 bh = dw_frag(s_bh + (wk + 16 * j + m) * DW_CSTRIDE + q * 16);
                const half8 bl = dw_frag(s_bl + (wk + 16 * j + m) * DW_CSTRIDE + q * 16);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[i], bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[i], bh, acc[i][j], 0, 0, 0);
                }
            }
        };
        wg_sync();
        for (int s = 0; s < nsteps; s += 2) {
            if (wave_live) multiply(0);
            wg_sync();
            if (s + 1 >= nsteps) break;
            if (wave_live) multiply(1);
            wg_sync();
        }
        // partial tile: lane (m, q) of accumulator tile (i, j) holds dW[n0 + wn + 16 i + 4 q + e][v0 + wk + 16 j + m]
        if (wave_live) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int vc = v0 + wk + 16 * j + m;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int n = n0 + wn + 16 * i + 4 * q + e;
                        if (n < L.N && vc < L.Kv) part[(long)n * L.Kv + vc] = acc[i][j][e];
                    }
                }
        }
    }
    // column sums of the gradient (bias gradient), once per gradient-column tile
    if (tk == 0) {
        __syncthreads();
        for (int i = tid; i < DW_TN; i += DW_THREADS)
            if (n0 + i < L.N) part[(long)L.N * L.Kv + n0 + i] = s_db[i];
    }
}

// pts_linears column of slot k of the fused trunk's `pe` buffer (= danbo_trunk_pe_column, k_mlp16.hip), -1: padding
__device__ __forceinline__ int pe_slot_column(int k) {
    const int ks = k >> 5, e = 4 * ((k >> 4) & 1) + (k & 3), q = (k >> 2) & 3;
    const int j = 8 * ks + e, c = j / 13, t = j % 13, kk = q + 4 * c;
    return (j < 52 && kk < FEAT) ? FEAT * t + kk : -1;
}

// sums the slices in order, scales back, writes nn.Linear layout
__global__ __launch_bounds__(256) void k_dw16_reduce(DwArgs a) {
    const int M = resolve_count(a.count, a.M);
    const int rps = dw_rows_per_slice(M, a.slices);
    const int live = M > 0 ? (M + rps - 1) / rps : 0;
    const DwLayer& L = a.l[blockIdx.y];
    float sc = 1.f, inv = 1.f;
    if (L.dy_maxabs != nullptr) dw_pow2_scale(*L.dy_maxabs, sc, inv);
    const int K = L.K1 + L.K2;
    const long per = (long)L.N * L.Kv + L.N;
    const long total = (long)L.N * K + L.N;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const bool is_b = idx >= (long)L.N * K;
        int n, k = 0;
        long src;
        if (is_b) {
            n = (int)(idx - (long)L.N * K);
            src = (long)L.N * L.Kv + n;
        } else {
            n = (int)(idx / K);
            k = (int)(idx % K);
            src = (long)n * L.Kv + (k < L.K1 ? k : k - L.K1 + L.K1p);
        }
        float s = 0.f;
        const float* p = a.part + L.part_off + src;
        int i = 0;
        for (; i + 4 <= live; i += 4) {          // four slices in flight (same summation order as the plain loop)
            const float v0 = p[(long)i * per], v1 = p[(long)(i + 1) * per], v2 = p[(long)(i + 2) * per], v3 = p[(long)(i + 3) * per];
            s += v0; s += v1; s += v2; s += v3;
        }
        for (; i < live; ++i) s += p[(long)i * per];
        if (is_b) {
            float* g = n < L.split_n ? L.gb : L.gb2;
            if (g != nullptr) g[n < L.split_n ? n : n - L.split_n] = s;   // the bias sums use the unscaled gradient
        } else {
            float* g = n < L.split_n ? L.gw : L.gw2;
            const int col = L.x1_pe ? pe_slot_column(k) : k;
            if (g != nullptr && col >= 0) g[(long)(n < L.split_n ? n : n - L.split_n) * L.gw_ld + L.gw_col0 + col] = s * inv;
        }
    }
}

}  // namespace danbo

using namespace danbo;

extern "C" long danbo_dw16_scratch_floats(const DanboDwLayer* layers, int n_layers, int slices) {
    if (!layers || n_layers < 1 || n_layers > DW_MAX_LAYERS || slices < 1) return -1;
    long total = 0;
    for (int i = 0; i < n_layers; ++i) {
        const long K1p = (layers[i].K1 + 3) & ~3;
        total += (long)slices * ((long)layers[i].N * (K1p + layers[i].K2) + layers[i].N);
    }
    return total;
}

extern "C" int danbo_dw16(const DanboDwLayer* layers, int n_layers, int M, const int32_t* count, int slices, float* scratch,
                          void* stream) {
    DANBO_CHECK_ARG(layers && n_layers >= 1 && n_layers <= DW_MAX_LAYERS && M >= 0 && slices >= 1 && slices <= 1024 && scratch);
    if (M == 0) return 0;
    DwArgs a;
    a.n_layers = n_layers;
    a.slices = slices;
    a.M = M;
    a.count = count;
    a.part = scratch;
    int tile = 0;
    long off = 0;
    for (int i = 0; i < n_layers; ++i) {
        const DanboDwLayer& s = layers[i];
        DANBO_CHECK_ARG(s.dy && s.x1 && s.N >= 1 && s.K1 >= 1 && s.K2 >= 0 && (s.K2 == 0 || s.x2));
        DANBO_CHECK_ARG(s.frag >= 0 && s.frag <= 3 && (uintptr_t)s.x1 % 16 == 0 && (uintptr_t)s.x2 % 16 == 0 && (uintptr_t)s.dy % 16 == 0);
        DANBO_CHECK_ARG((s.frag & 2) ? (s.K1 % 32 == 0 && s.K2 == 0) : (s.ld1 % 4 == 0 && s.ld1 >= ((s.K1 + 3) & ~3)));
        DANBO_CHECK_ARG(s.K2 == 0 || (s.ld2 % 4 == 0 && s.ld2 >= s.K2));
        DANBO_CHECK_ARG(!s.x1_pe || ((s.frag & 2) && s.K1 == DANBO_TRUNK_PE_WIDTH && s.K2 == 0 && s.gw_ld >= s.gw_col0 + 195));
        DANBO_CHECK_ARG(s.gw_ld == 0 || s.x1_pe || (s.gw_ld >= s.gw_col0 + s.K1 + s.K2 && s.gw_col0 >= 0 && !s.gw2));
        DANBO_CHECK_ARG((s.frag & 1) ? s.N % 32 == 0 : (s.ldy % 4 == 0 && s.ldy >= s.N));
        DwLayer& L = a.l[i];
        L.dy = s.dy; L.x1 = s.x1; L.x2 = s.x2; L.dy_maxabs = s.dy_maxabs;
        L.gw = s.gw; L.gw2 = s.gw2; L.gb = s.gb; L.gb2 = s.gb2;
        L.ldy = s.ldy; L.ld1 = s.ld1; L.ld2 = s.ld2; L.N = s.N; L.K1 = s.K1; L.K2 = s.K2; L.frag = s.frag;
        L.gw_ld = s.gw_ld > 0 ? s.gw_ld : s.K1 + s.K2;
        L.gw_col0 = s.gw_ld > 0 ? s.gw_col0 : 0;
        L.x1_pe = s.x1_pe;
        L.split_n = s.gw2 || s.gb2 ? s.split_n : 0x7fffffff;
        L.K1p = (s.K1 + 3) & ~3;
        L.Kv = L.K1p + s.K2;
        L.tile0 = tile;
        L.nt_k = (L.Kv + DW_TK - 1) / DW_TK;
        tile += ((s.N + DW_TN - 1) / DW_TN) * L.nt_k;
        L.part_off = off;
        off += (long)slices * ((long)s.N * L.Kv + s.N);
    }
    a.n_tiles = tile;
    DANBO_ENSURE_LDS(k_dw16, DW_LDS_BYTES);
    const int items = ((tile + 1) & ~1) * slices;
    hipLaunchKernelGGL(k_dw16, dim3((items + 15) / 16 * 16), dim3(DW_THREADS), DW_LDS_BYTES, (hipStream_t)stream, a);
    hipLaunchKernelGGL(k_dw16_reduce, dim3(128, n_layers), dim3(256), 0, (hipStream_t)stream, a);
    DANBO_LAUNCH_RET();
}
