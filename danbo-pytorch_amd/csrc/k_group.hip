// danbo_group_rows: reorder the compacted in-volume rows so that rows whose samples lie inside the SAME SET of bone volumes are
// neighbours.  No reference counterpart -- the reference evaluates every sample against every bone
// (core/networks/gnn_backbone.py:787-828, danbo.py:299-300); this only serves the launch that follows.
//
// Why: k_assign16 (csrc/k_assign16.hip) gives a wavefront 32 consecutive rows and evaluates, for all 32, every bone that is valid
// for at least one of them, plus the features of those bones' tree neighbours.  In the order the cull kernel compacts (ray-major
// pieces) a wavefront's rows run from the front bone of a ray to its back bone and across limbs: 2.86 bones evaluated for 1.65
// valid per row, 4.5 feature bone-pairs where 3.3 would do (tools/diag/assign_wave_stats.py, bench frame).  Grouped by bone set
// inside windows of 16 384 rows: 1.71 and 3.36.  Every row's result is independent of its neighbours (a skipped bone has p_j = 0
// exactly), so the order is free -- K2's h rows and K3's scatter go through the same list.
//
// One workgroup per window, in place: rows -> (bits) -> slot of an LDS hash table of the window's distinct bit sets (typically
// 20 .. 60) + rank inside the set by an LDS atomic -> sets ordered by bit-reversed value (sets that share their lowest bone end
// up next to each other: a wavefront that straddles two sets evaluates a small union) -> write back.  A window with more
// distinct sets than the table holds is left as it is.
#include "common.hpp"

namespace danbo {

constexpr int GR_THREADS = 1024;
constexpr int GR_PER_THREAD = 16;
constexpr int GR_WINDOW = GR_THREADS * GR_PER_THREAD;   // 16 384 rows
constexpr int GR_SLOTS = 2048;                          // hash table (power of two)
constexpr int GR_MAX_SETS = 1024;                       // distinct bit sets of a window the ordering handles
constexpr uint32_t GR_EMPTY = 0xffffffffu;              // (a row's word is never 0xffffffff: 24 bones)

__global__ __launch_bounds__(GR_THREADS) void k_group_rows(const uint32_t* __restrict__ bits, int32_t* __restrict__ list,
                                                           const int32_t* __restrict__ count, int n_cap) {
    __shared__ uint32_t s_key[GR_SLOTS];     // bit set of the slot
    __shared__ int s_cnt[GR_SLOTS];          // rows of the set, then its first position in the window
    __shared__ int s_dense[GR_MAX_SETS];     // the used slots
    __shared__ int s_nsets, s_overflow;
    const int n = resolve_count(count, n_cap);
    const int win0 = blockIdx.x * GR_WINDOW;
    if (win0 >= n) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < GR_SLOTS; i += GR_THREADS) { s_key[i] = GR_EMPTY; s_cnt[i] = 0; }
    if (tid == 0) { s_nsets = 0; s_overflow = 0; }
    __syncthreads();

    int ms[GR_PER_THREAD], slot[GR_PER_THREAD], rank[GR_PER_THREAD];
#pragma unroll
    for (int k = 0; k < GR_PER_THREAD; ++k) {
        const int row = win0 + k * GR_THREADS + tid;
        ms[k] = row < n ? list[row] : -1;
    }
    const int lane = tid & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t bk[GR_PER_THREAD];            // all 16 gathers in flight at once (one per loop turn was 16 dependent HBM round trips)
#pragma unroll
    for (int k = 0; k < GR_PER_THREAD; ++k) bk[k] = ms[k] >= 0 ? bits[ms[k]] : GR_EMPTY;
#pragma unroll
    for (int k = 0; k < GR_PER_THREAD; ++k) {
        slot[k] = -1;
        rank[k] = 0;
        const bool live = ms[k] >= 0;
        const uint32_t b = bk[k];
        // One table insertion and ONE counter update per wavefront and distinct set (64 neighbouring rows hold one to three sets:
        // per-row atomics on the same few LDS words serialised -- 61 us for the bench frame's 650 k rows).
        unsigned long long todo = __ballot(live);
        while (todo != 0ull) {
            const int leader = (int)__builtin_ctzll(todo);
            const uint32_t key = (uint32_t)__builtin_amdgcn_readlane((int)b, leader);
            const unsigned long long same = __ballot(live && b == key);
            int sl = -1, base = 0;
            if (lane == leader) {
                uint32_t h = (key * 2654435761u) >> (32 - 11);          // Fibonacci hash -> [0, 2048)
                for (int probe = 0; probe < GR_SLOTS; ++probe) {
                    const uint32_t old = atomicCAS(&s_key[h], GR_EMPTY, key);
                    if (old == GR_EMPTY) {                              // first rows of a new set: register the slot
                        const int d = atomicAdd(&s_nsets, 1);
                        if (d < GR_MAX_SETS) s_dense[d] = (int)h; else s_overflow = 1;
                    }
                    if (old == GR_EMPTY || old == key) { sl = (int)h; break; }
                    h = (h + 1) & (GR_SLOTS - 1);
                }
                if (sl >= 0) base = atomicAdd(&s_cnt[sl], (int)__popcll(same)); else s_overflow = 1;
            }
            sl = __builtin_amdgcn_readlane(sl, leader);
            base = __builtin_amdgcn_readlane(base, leader);
            if (live && b == key) {
                slot[k] = sl;
                rank[k] = base + (int)__popcll(same & lt_mask);
            }
            todo &= ~same;
        }
    }
    __syncthreads();
    if (s_overflow) return;          // (workgroup-uniform) too many distinct sets: the window keeps the cull order
    // first position of every set: the rows of all sets that sort before it (bit-reversed value, ties impossible: keys distinct)
    const int nsets = s_nsets;
    int first = 0;
    uint32_t mykey = 0;
    if (tid < nsets) {
        mykey = __brev(s_key[s_dense[tid]]);
        for (int i = 0; i < nsets; ++i) {
            const int s = s_dense[i];
            if (__brev(s_key[s]) < mykey) first += s_cnt[s];
        }
    }
    __syncthreads();
    if (tid < nsets) s_cnt[s_dense[tid]] = first;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GR_PER_THREAD; ++k)
        if (ms[k] >= 0) list[win0 + s_cnt[slot[k]] + rank[k]] = ms[k];
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
