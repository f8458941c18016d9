// danbo_group_rows: reorder the compacted in-volume rows so that rows whose samples lie inside the same bone volumes are
// neighbours.  No reference counterpart -- the reference evaluates every sample against every bone
// (core/networks/gnn_backbone.py:787-828, danbo.py:299-300); this only serves the launch that follows.
//
// Why: k_assign16 (csrc/k_assign16.hip) gives a wavefront 32 consecutive rows and evaluates, for all 32, every bone that is valid
// for at least one of them, plus the features of those bones' tree neighbours.  In the order the cull kernel compacts (ray-major
// pieces) a wavefront's rows run from the front bone of a ray to its back bone and across limbs: 2.86 bones evaluated for 1.65
// valid per row (tools/diag/assign_wave_stats.py, bench frame).  Grouped by (lowest valid bone, second lowest valid bone) inside
// windows of 16 384 rows: 1.80 (by the whole set: 1.71).  Every row's result is independent of its neighbours (a skipped bone has
// p_j = 0 exactly), so the order is free -- K2's h rows and K3's scatter go through the same list.
//
// One workgroup per window, in place, a counting sort over 600 bins = 24 x 25 (lowest, second lowest or none) -- neighbouring bins
// share their lowest bone, so a wavefront that straddles two bins evaluates a small union.  No atomics: every wavefront counts
// into its OWN row of the LDS histogram (the leader lane of each distinct key of a 64-row vector adds the ballot's population),
// ranks inside a vector come from the ballot, and bin starts from two scans (over the 16 wavefronts of a bin, over the bins).
// The output order is a pure function of the input, and stable (rows of a bin keep their input order).  (First version: one LDS
// hash table of the window's distinct sets with atomicCAS / atomicAdd per wavefront and set -- 27 us alone, 86 us beside the
// view-constant kernel on the side stream: sixteen wavefronts queueing on the same few LDS words.)
#include "common.hpp"

namespace danbo {

constexpr int GR_THREADS = 1024;
constexpr int GR_WAVES = GR_THREADS / 64;
constexpr int GR_PER_THREAD = 16;
constexpr int GR_WINDOW = GR_THREADS * GR_PER_THREAD;   // 16 384 rows
constexpr int GR_BINS = J * (J + 1);                    // 600

__device__ __forceinline__ int gr_bin(uint32_t b) {
    b &= (1u << J) - 1u;
    if (b == 0u) return 0;                              // (not a listed row; harmless)
    const int low = __builtin_ctz(b);
    const uint32_t rest = b & (b - 1u);
    return low * (J + 1) + (rest ? __builtin_ctz(rest) : J);
}

__global__ __launch_bounds__(GR_THREADS) void k_group_rows(const uint32_t* __restrict__ bits, int32_t* __restrict__ list,
                                                           const int32_t* __restrict__ count, int n_cap) {
    __shared__ int s_cnt[GR_WAVES][GR_BINS];   // rows of (wavefront, bin); then the first position of the wavefront's rows of the bin
    __shared__ int s_tot[GR_BINS];             // rows of the bin; then its first position in the window
    __shared__ int s_wsum[GR_WAVES];
    const int n = resolve_count(count, n_cap);
    const int win0 = blockIdx.x * GR_WINDOW;
    if (win0 >= n) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < GR_WAVES * GR_BINS; i += GR_THREADS) (&s_cnt[0][0])[i] = 0;

    int ms[GR_PER_THREAD];
    uint32_t bk[GR_PER_THREAD];            // all 16 gathers in flight at once
#pragma unroll
    for (int k = 0; k < GR_PER_THREAD; ++k) {
        // a wavefront owns 1 024 CONSECUTIVE rows (16 vectors of 64): position inside a bin = (wavefront, vector, lane) is then
        // the input order -- the sort is stable, samples of a ray stay together, and with them the rarer third / fourth bones
        // of a bin (an interleaved assignment scattered those over the whole bin: K2 157 instead of 135 us)
        const int row = win0 + wave * (64 * GR_PER_THREAD) + k * 64 + lane;
        ms[k] = row < n ? list[row] : -1;
    }
#pragma unroll
    for (int k = 0; k < GR_PER_THREAD; ++k) bk[k] = ms[k] >= 0 ? bits[ms[k]] : 0u;
    __syncthreads();

    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    int bin[GR_PER_THREAD], rank[GR_PER_THREAD];
#pragma unroll
    for (int k = 0; k < GR_PER_THREAD; ++k) {
        const bool live = ms[k] >= 0;
        bin[k] = gr_bin(bk[k]);
        rank[k] = 0;
        unsigned long long todo = __ballot(live);
        while (todo != 0ull) {                          // one turn per distinct bin of this 64-row vector (one to three)
            const int leader = (int)__builtin_ctzll(todo);
            const int key = __builtin_amdgcn_readlane(bin[k], leader);
            const unsigned long long same = __ballot(live && bin[k] == key);
            int base = 0;
            if (lane == leader) {                       // this wavefront's own counter: plain read-modify-write
                base = s_cnt[wave][key];
                s_cnt[wave][key] = base + (int)__popcll(same);
            }
            base = __builtin_amdgcn_readlane(base, leader);
            if (live && bin[k] == key) rank[k] = base + (int)__popcll(same & lt_mask);
            todo &= ~same;
        }
    }
    __syncthreads();
    // per bin: exclusive scan over the wavefronts, total
    if (tid < GR_BINS) {
        int run = 0;
#pragma unroll
        for (int w = 0; w < GR_WAVES; ++w) {
            const int c = s_cnt[w][tid];
            s_cnt[w][tid] = run;
            run += c;
        }
        s_tot[tid] = run;
    }
    __syncthreads();
    // exclusive scan of the 600 totals: inclusive scan inside each wavefront, then the wavefront sums
    int incl = 0, mine = 0;
    if (tid < 640) {
        mine = tid < GR_BINS ? s_tot[tid] : 0;
        incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_wsum[wave] = incl;
    }
    __syncthreads();
    if (tid < GR_BINS) {
        int before = 0;
        for (int w = 0; w < wave; ++w) before += s_wsum[w];
        s_tot[tid] = before + incl - mine;              // first position of the bin
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GR_PER_THREAD; ++k)
        if (ms[k] >= 0) list[win0 + s_tot[bin[k]] + s_cnt[wave][bin[k]] + rank[k]] = ms[k];
}

}  // namespace danbo

using namespace danbo;

extern "C" int danbo_group_rows(const uint32_t* valid_bits, int32_t* list, const int32_t* count, int n_cap, void* stream) {
    DANBO_CHECK_ARG(valid_bits && list && n_cap >= 0);
    if (n_cap == 0) return 0;
    hipLaunchKernelGGL(k_group_rows, dim3(ceil_div(n_cap, GR_WINDOW)), dim3(GR_THREADS), 0, (hipStream_t)stream, valid_bits, list, count,
                       n_cap);
    DANBO_LAUNCH_RET();
}
