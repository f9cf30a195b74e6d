// Trace words: the read-out of a short discrete trace without a lineage walk.
//
// The posterior read-out (reference stats_printer.hpp:88-120 over the final particles' traces) needs, for every predict hit t, the
// state x_t of every final particle's ANCESTOR at generation t.  kernels.hpp finds it by walking anc[] backwards: T dependent gathers
// per particle, ~1.5 us a hop -- 25-30 us of the 164 us headline run, behind the last step instead of under it.  When a whole trace
// fits a machine word (hmm<T>: 2 bits a state, T <= 16) the particle can carry it: the step kernel gathers its ancestor's WORD where
// it gathered the ancestor's state byte (the same dependent load; the state is the word's top field), ors its own state in and stores
// the word.  After the last step every particle holds its full trace, and the read-out is one streaming pass:
//     n[t][s][c] = #{ i : x_t(trace_i) = s, x_{T-1}(trace_i) = c }          (integers: exact, order-free)
//     P(x_t = s | y) = sum_c e_c n[t][s][c] / sum_c e_c n_c                  (e_c: the last step's three weights, step_counts.hpp)
// The particle store (values[], anc[]) is written as before: cpprob_hip_copy_paths / _ancestors do not change.
//
// trace_readout_kernel: a lane counts, for its particles, the pairs with x_t in {1, 2} (x_t = 0 follows from the class sizes): sixteen
// 32-bit accumulators of six 5-bit counters, flushed through 15-bit packed wave sums; a workgroup's 6 T counts go into one of
// kTraceSlots counter sets by atomic adds (a line per counter), the workgroup that arrives last collects and clears them, and writes
// the statistics: no partials array, no finalize launch.
#pragma once
#include "step_fixed.hpp"

namespace cph {

constexpr int kTraceMaxT = 16;                  // 2 bits a state in a 32-bit word
constexpr int kTraceKeys = 6;                   // (x_t, x_{T-1}) pairs with x_t in {1, 2}: key = 3 (x_t - 1) + x_{T-1}
constexpr int kTraceSlots = 2;                  // counter sets the workgroups spread their atomic adds over (same-call A/B, read-out us:
                                                // 8 sets 14.9, 4 13.3, 2 12.4, 1 12.9 -- the last workgroup's collect is one exchange per set)
constexpr int kTraceLine = 32;                  // 32-bit words between two counters: a 128-byte line each
constexpr int kTraceBatch = 4;                  // tiles a workgroup has in flight: 16 particles a lane between two flushes (5-bit counters hold 31)
constexpr size_t kTraceCounterWords = (size_t)kTraceSlots * kTraceMaxT * kTraceKeys * kTraceLine;

struct TraceReadoutArgs {
    const uint32_t* trace; int64_t n; int T;
    CountsFinal f;
    uint32_t* counters;                          // [kTraceSlots][kTraceMaxT * kTraceKeys] lines, zero between two launches
    unsigned long long* arrive;                  // zero between two launches
    double* stats;                               // [T][3]
    int raw; double n_local;                     // one shard of a joint population: un-normalised sums over THIS shard's n_local particles (the ranks'
                                                 // sums are all-reduced and divided by the joint normaliser: group.hpp), no bookkeeping here
};

__global__ __launch_bounds__(kThreads) void trace_readout_kernel(TraceReadoutArgs a)
{
    __shared__ uint32_t s_pack[kWaves][3 * kTraceMaxT];          // per wave and t: fields {0, 3}, {1, 4}, {2, 5} as 15-bit packed sums
    __shared__ uint32_t s_cnt[kTraceMaxT * kTraceKeys];          // the workgroup's counts
    __shared__ unsigned long long s_n[kTraceMaxT * kTraceKeys];  // (last workgroup) the population's counts
    __shared__ int s_last;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    if (a.f.bookkeep && !a.raw && blockIdx.x == 0 && wv == 0) {
        const Cnt2 tl = hier_total(a.f.h);
        if (tid == 0) counts_final_bookkeep(a.f, (double)tl.n0, (double)tl.n1);
    }
    for (int i = tid; i < kTraceMaxT * kTraceKeys; i += kThreads) s_cnt[i] = 0;
    __syncthreads();

    const int sh_last = 2 * (a.T - 1);
    const int64_t ntiles = (a.n + kTile - 1) / kTile;
    // (workgroup-uniform trip count: the flush holds barriers)
    for (int64_t tile0 = blockIdx.x; tile0 < ntiles; tile0 += (int64_t)gridDim.x * kTraceBatch) {
        U4 w[kTraceBatch];
#pragma unroll
        for (int b = 0; b < kTraceBatch; ++b) {                  // the batch's loads travel together
            const int64_t tile = tile0 + (int64_t)b * gridDim.x;
            const int64_t j0 = (tile < ntiles ? tile : 0) * kTile + (int64_t)tid * kPPT;
            w[b] = *reinterpret_cast<const U4*>(a.trace + j0);
        }
        uint32_t acc[kTraceMaxT];
#pragma unroll
        for (int t = 0; t < kTraceMaxT; ++t) acc[t] = 0;
#pragma unroll
        for (int b = 0; b < kTraceBatch; ++b) {
            const int64_t tile = tile0 + (int64_t)b * gridDim.x;
            const int64_t j0 = tile * kTile + (int64_t)tid * kPPT;
#pragma unroll
            for (int k = 0; k < kPPT; ++k) {
                const uint32_t v = w[b][k];
                // the particle's class field, three fields up: x_t = 2 leaves it there, x_t = 1 brings it down, x_t = 0 drops it
                const uint32_t base = (tile < ntiles && j0 + k < a.n) ? 1u << (15u + 5u * ((v >> sh_last) & 3u)) : 0u;
#pragma unroll
                for (int t = 0; t < kTraceMaxT; ++t) acc[t] += base >> (30u - 15u * ((v >> (2 * t)) & 3u));        // (t >= T: state 0)
            }
        }
        // 5-bit fields -> three words of two 15-bit fields: room for the sum over 64 lanes x 16
#pragma unroll
        for (int t = 0; t < kTraceMaxT; ++t) {
            const uint32_t p0 = wave_sum_u32(acc[t] & 0x000f801fu), p1 = wave_sum_u32((acc[t] >> 5) & 0x000f801fu), p2 = wave_sum_u32((acc[t] >> 10) & 0x000f801fu);
            if (lane == 0) { s_pack[wv][3 * t] = p0; s_pack[wv][3 * t + 1] = p1; s_pack[wv][3 * t + 2] = p2; }
        }
        __syncthreads();
        if (tid < kTraceMaxT * kTraceKeys) {
            const int t = tid / kTraceKeys, key = tid - t * kTraceKeys;
            const int word = key < 3 ? key : key - 3, shift = key < 3 ? 0 : 15;
            uint32_t s = 0;
#pragma unroll
            for (int w2 = 0; w2 < kWaves; ++w2) s += (s_pack[w2][3 * t + word] >> shift) & 0x7fffu;
            s_cnt[tid] += s;
        }
        __syncthreads();
    }

    // the workgroup's counts into one of the counter sets; performed before its arrival is counted
    if (tid < a.T * kTraceKeys && s_cnt[tid] != 0)
        atomicAdd(a.counters + ((size_t)(blockIdx.x % kTraceSlots) * kTraceMaxT * kTraceKeys + tid) * kTraceLine, s_cnt[tid]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(a.arrive, 1ull) == (unsigned long long)gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!s_last) return;

    // last to arrive: every workgroup's adds have been performed.  Collect and clear (read where the adds were performed).
    if (tid < a.T * kTraceKeys) {
        unsigned long long n = 0;
#pragma unroll
        for (int s = 0; s < kTraceSlots; ++s) n += atomicExch(a.counters + ((size_t)s * kTraceMaxT * kTraceKeys + tid) * kTraceLine, 0u);
        s_n[tid] = n;
    }
    if (tid == 0) *a.arrive = 0;
    __syncthreads();
    if (tid < a.T * 3) {
        const int t = tid / 3, s = tid - 3 * t;
        const double e0 = a.f.e[0], e1 = a.f.e[1], e2 = a.f.e[2];
        // the class sizes -- x_{T-1} = c pairs with itself -- and the normaliser as counts_final_bookkeep takes it
        const unsigned long long* last = s_n + (a.T - 1) * kTraceKeys;
        const double N1 = (double)last[1], N2 = (double)last[5], N0 = (a.raw ? a.n_local : a.f.n_pop) - N1 - N2;
        const double W = fma(N2, e2, fma(N1, e1, __dmul_rn(N0, e0)));
        const unsigned long long* row = s_n + t * kTraceKeys;
        double n0, n1, n2;                                        // particles of class 0 / 1 / 2 whose trace held s at t
        if (s == 0) { n0 = N0 - (double)(row[0] + row[3]); n1 = N1 - (double)(row[1] + row[4]); n2 = N2 - (double)(row[2] + row[5]); }
        else { n0 = (double)row[3 * (s - 1)]; n1 = (double)row[3 * (s - 1) + 1]; n2 = (double)row[3 * (s - 1) + 2]; }
        const double num = fma(n2, e2, fma(n1, e1, __dmul_rn(n0, e0)));
        a.stats[t * 3 + s] = a.raw ? num : num / W;
    }
}

}  // namespace cph
