// SMC step on FIXED-POINT weights: continuous-weight models and ESS-triggered schedules, systematic resampling.
//
// What step_counts.hpp does for table weights, for any weight: the linear weight of particle i of generation t is the INTEGER
//        q_i = min(rint(exp(lw_i - R_t) * 2^32), 2^32 - 1),
// taken against a reference R_t >= max_i lw_i that every workgroup knows before it has seen a single weight:
//        R_t = B_t                   when every particle enters step t at log-weight 0 (t = 0, or step t-1 resampled),
//        R_t = M_{t-1} + B_t         otherwise (weights carried),
// B_t = the host-known upper bound of the step's incremental log-weight (the Gaussian emission's density at its mode; the
// largest table value), M_{t-1} = the exact maximum of generation t-1's log-weights (a maximum is order-free).  Sums of integers
// are exact in any order, so
//        the inclusive CDF  C_k = sum_{i <= k} q_i                                   (exact, 64-bit),
//        the first output   G_k = ceil(fma(double(C_k), N / double(C_N), -u0)),      ancestor of output j = min{k : G_k > j},
//        the normaliser     W = C_N * 2^-32,   ESS = (C_N * 2^-16)^2 / sum_i fix_square(q_i)    (squares of 24-bit weights, 32 bits each:
//                                                                                             exact in 64 bits up to 2^28 particles)
// are the same integers however tiles, wavefronts and shards are laid out: the resampling decision, every ancestor and the
// evidence of a sharded run equal the single-GPU run's bit for bit, and no workgroup has to re-derive a floating-point CDF from
// every tile partial (kernels.hpp: the fused prologue) or wait for a normalisation launch (scan_partials_kernel).  The CPU
// restatement the parity tests compare with states the same arithmetic.
//
// The prefix masses live in the 64-ary hierarchy of step_counts.hpp (same layout, same rotation of three copies): the word of a
// tile / block is its mass S (levels >= 1: | arrivals << 56), and the line of a block also holds Q = sum fix_square(q) and the
// order key of M = max lw (atomic add / add / max: all order-free); tiles keep their Q and M in two arrays beside level 0.
// Resolution: weights below 2^-33 of the reference are zero -- 23 nats under the heaviest admissible particle.
#pragma once
#include "step_counts.hpp"
#include "strata_cut.hpp"

namespace cph {

// (fixed-point weights, 64-bit wavefront scans, the mass hierarchy, search, walk and decision: cpprob/detail/fixed_mass.hpp)
// Both parts in every wavefront (the exchange scope's packing, whose output tiles sit anywhere in the shard).  Stratified: the
// outputs' uniforms are staged here first (behind a barrier of its own).
template <int RS = kFixSystematic>
__device__ __forceinline__ void ancestors_fixed(const FHier& f, const FixedCdf& fc, const uint32_t* __restrict__ qprev, int64_t n, int nb, bool last_shard,
                                                double gj_first, int n_out, int guess, int32_t (&anc)[kPPT], FixedLdsT<RS>& L)
{
    if constexpr (RS == kFixStratified) { stratified_stage(L, fc.seed, fc.draw, fc.uid0 + (uint64_t)gj_first); __syncthreads(); }
    const FLocated loc = fixed_locate<RS>(f, fc, nb, gj_first, n_out, guess, nullptr);
    const U4 z = {0u, 0u, 0u, 0u};
    fixed_walk<RS>(fc, qprev, n, nb, last_shard, gj_first, n_out, loc, guess, false, z, z, z, anc, L);
}

// all-gathered totals of the ranks: 3 words per rank {S, Q, key(M)} (they travel as 24 bytes, whatever the collective calls them)
struct FixedRanks { uint64_t S, Q; double M; uint64_t before; };
__device__ __forceinline__ FixedRanks fixed_ranks(const uint64_t* __restrict__ all, int world, int rank)
{
    const int lane = lane_id();
    uint64_t s = 0, q = 0, m = 0;
    if (lane < world) { s = all[3 * lane]; q = all[3 * lane + 1]; m = all[3 * lane + 2]; }
    FixedRanks r;
    r.S = wave_sum_u64(s); r.Q = wave_sum_u64(q); r.M = dkey_inv(wave_max_u64(m)); r.before = wave_sum_u64(lane < rank ? s : 0ull);
    return r;
}

struct FixedFound { FLocated loc; double inv, ref; uint64_t base, S, own; int64_t l0, l1; int resample, w0, w1; };     // the searching wavefront's hand-over

#ifndef CPPROB_HAND_OVER_FIXED
#define CPPROB_HAND_OVER_FIXED 1
#endif
#ifndef CPPROB_PARK_DRAWS
#define CPPROB_PARK_DRAWS 1
#endif
template <class Model>
struct StepFixedArgs {
    ModelParams mp; const double* obs; int t, T; int64_t n, ld, rs;
    uint64_t seed, pid0;
    typename Model::store_t* values; int32_t* anc;
    FHier f;                                   // generation t-1's masses (read); generation t's are written one copy further
    const uint32_t* q_prev; uint32_t* q_next;  // [ld] integer weights, ping-pong
    const double* logw_prev; double* logw_next;
    double u0;                                 // systematic offset of the resampling before step t (Philox, evaluated on the host)
    double bound_prev, bound;                  // B_{t-1}, B_t
    double ess_frac; int may_carry;            // may_carry = 0: every step resamples (known on the host): no log-weight ever carries
    int prefetch;                              // fetch the three likely source tiles' weights at kernel entry: a round trip saved where every
                                               // step resamples; off where steps may not (12 wasted bytes per particle on those, twelve
                                               // registers on all: -1 % at 1.25e6 particles of the LGSSM, same-call A/B)
    StepCtrl* ctrl; double n_pop; double* ess_trace; int32_t* resampled;
    const uint64_t* all_totals; int world, rank;       // one shard of a joint population (exchange scope): every rank's {S, Q, key(M)} of generation t-1
    const int64_t* annex_base;
    int row_w, row_r;
    // multinomial resampling: exclusive prefix masses of generation t-1's tiles ([nb + 1]: fixed_tile_prefix_kernel) and the
    // in-tile inclusive prefix at the end of every lane's four particles ([ld / 4]), read for generation t-1 / written for generation t
    const uint64_t* tile_prefix; const uint64_t* lane_prefix_prev; uint64_t* lane_prefix_next;
    const uint32_t* strata_offs; int strata_k; // multinomial, strata form: [2^k + 1] first output of every stratum's thresholds at this step (multinomial_strata_kernel)
    CutView cut;                               // ... of one shard of a joint population: what the ranks' boundaries cut (strata_cut.hpp), written by the exchange that preceded this step
};

// ---- multinomial resampling on integer masses (thesis Alg. 1 p.36: a_j ~ Categorical(W)) ---------------------------------------
// Output j draws its own 53-bit uniform u = b 2^-53 (the words draw_u01_53 takes: pair id & 1 of block id >> 1) and its threshold is
// the INTEGER tau_j = floor(u C_N) = mulhi64(b << 11, C_N); its ancestor is min{k : C_k > tau_j}.  No rounding anywhere, tau_j < C_N
// always.  The thresholds are unsorted, so every output searches on its own: the tile by a binary search of the tiles' exclusive
// prefix masses, the lane inside the tile by a binary search of the lanes' inclusive prefixes (what the producing step left beside
// the weights: 2 bytes a particle), the particle among the lane's four weights.
// Exclusive prefix masses of the tiles of one generation: out[c] = mass of tiles 0 .. c-1, out[nb] = C_N.  One workgroup.
constexpr int kTilePrefixThreads = 1024;
__global__ __launch_bounds__(kTilePrefixThreads) void fixed_tile_prefix_kernel(const uint64_t* __restrict__ tile_mass, int nb, uint64_t* __restrict__ out)
{
    __shared__ uint64_t s_w[kTilePrefixThreads / kWave];
    const int tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    uint64_t carry = 0;
    for (int base = 0; base < nb; base += kTilePrefixThreads) {
        const int i = base + tid;
        const uint64_t v = i < nb ? (tile_mass[i] & kMassMask) : 0ull;
        const uint64_t incl = wave_incl_scan_u64(v);
        if (lane == kWave - 1) s_w[wv] = incl;
        __syncthreads();
        uint64_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kTilePrefixThreads / kWave; ++w) { const uint64_t x = s_w[w]; if (w < wv) off += x; tot += x; }
        if (i < nb) out[i] = carry + off + incl - v;
        carry += tot;
        __syncthreads();
    }
    if (tid == 0) out[nb] = carry;
}

// Ancestors of the four outputs whose ids start at `uid` (a multiple of four).  S = C_N > 0.
__device__ __forceinline__ void fixed_multinomial_ancestors(uint64_t S, uint64_t seed, uint64_t draw, uint64_t uid, const uint64_t* __restrict__ tile_prefix, int nb,
                                                            const uint64_t* __restrict__ lane_prefix, const uint32_t* __restrict__ q, int32_t (&anc)[kPPT])
{
    static_assert(kPPT == 4, "two Philox blocks a lane");
    const u32x4 b0 = draw_block(seed, uid >> 1, draw), b1 = draw_block(seed, (uid >> 1) + 1, draw);
    uint64_t tau[kPPT];
    tau[0] = __umul64hi(bits53(b0.x, b0.y) << 11, S); tau[1] = __umul64hi(bits53(b0.z, b0.w) << 11, S);
    tau[2] = __umul64hi(bits53(b1.x, b1.y) << 11, S); tau[3] = __umul64hi(bits53(b1.z, b1.w) << 11, S);
    // the tile: the largest c with prefix[c] <= tau (prefix[0] = 0 <= tau < prefix[nb] = S); four searches side by side
    int lo[kPPT], hi[kPPT];
    uint64_t at[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) { lo[k] = 0; hi[k] = nb; at[k] = 0; }
    const int iters = nb > 1 ? 32 - __builtin_clz((unsigned)(nb - 1)) : 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const int mid = (lo[k] + hi[k]) >> 1;
            const uint64_t v = tile_prefix[mid];
            if (v <= tau[k]) { lo[k] = mid; at[k] = v; } else hi[k] = mid;
        }
    }
    // the lane: the first l with P[l] > rem (P[255] = the tile's mass > rem)
    int a[kPPT], b[kPPT];
    uint64_t rem[kPPT], below[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) { a[k] = 0; b[k] = kThreads - 1; rem[k] = tau[k] - at[k]; below[k] = 0; }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const int mid = (a[k] + b[k]) >> 1;
            const uint64_t v = lane_prefix[(int64_t)lo[k] * kThreads + mid];
            if (v > rem[k]) b[k] = mid; else { a[k] = mid + 1; below[k] = v; }
        }
    }
    static_assert(kThreads == 256, "eight halvings of a tile's 256 lanes");
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        const int l = a[k] < kThreads ? a[k] : kThreads - 1;                 // (a population without mass: nothing to find)
        const int64_t i0 = (int64_t)lo[k] * kTile + (int64_t)l * kPPT;
        const U4 w = *reinterpret_cast<const U4*>(q + i0);
        uint64_t r = rem[k] - below[k];
        int j = 0;
        if (r >= w[0]) { r -= w[0]; j = 1; if (r >= w[1]) { r -= w[1]; j = 2; if (r >= w[2]) j = 3; } }
        anc[k] = (int32_t)(i0 + j);
    }
}

// ---- multinomial resampling, STRATA form: the form that runs by default --------------------------------------------------------
// The literal form above is N searches of the whole population at random addresses (19 dependent scattered loads an output at 10^6
// particles: 6 - 8 x the systematic step, 25 x at 10^7).  The same law -- N iid uniform thresholds on [0, C_N) -- generated in nearly
// sorted order keeps an output's ancestor next to it, as under systematic resampling:
//      N iid uniforms  =  how many fall into each of K = 2^k equal strata, (m_w) ~ Multinomial(N; 1/K .. 1/K),
//                         and iid uniforms inside each stratum.
//   1  The counts do not depend on the weights: multinomial_strata_kernel draws them for every step of a run in ONE launch at the
//      run's start -- a binary tree over the strata, the n thresholds of a node go left with probability 1/2 each: left = popcount of
//      the first n bits of the node's own Philox stream, an exact Binomial(n, 1/2) in integers.  o_w = m_0 + .. + m_w-1.
//   2  The step launch: output s in [o_w, o_w+1) draws the 53-bit uniform of OUTPUT s (draw kResampleDrawBase2 + step);
//          tau_s = B_w + floor(v_s (B_w+1 - B_w)),   B_w = floor(C_N w / K),   ancestor = min{k : C_k > tau_s}.
//      An output tile's thresholds lie in two or three neighbouring strata, i.e. in the two or three source tiles around its own
//      index: each of them rebuilds its prefix masses in LDS and the outputs search there.  No atomics, no launch between two steps.
// K = the smallest power of two >= four times the number of tiles (strata_levels).  Integers throughout (the CPU restatement: orc_resample_fixed_multinomial_strata).
__device__ __forceinline__ uint64_t strata_bound(uint64_t S, uint64_t w, int k)
{
    if (k == 0) return w ? S : 0ull;
    return (__umul64hi(S, w) << (64 - k)) | ((S * w) >> k);
}

// The counts are spread over the chip in two parts for k > 6 (the sum of independent multinomial counts is multinomial):
//   top     the outputs are dealt to G = clamp(K / 128, 1, 64) groups of consecutive outputs; every group sends its own through the top
//           six levels of a tree of ITS OWN (workgroup = (group, step); stream of node `node` of group g: block group
//           1 << 63 | g << 40 | node << 32 | chunk) and adds its counts of the 64 level-6 nodes to the step's totals (64 atomics a workgroup);
//   bottom  workgroup (i, step) splits the total of level-6 node i down the remaining k - 6 levels (streams by heap index) and writes
//           the first outputs of its K / 64 strata: its own prefix sums on top of the nodes' before it.
// k <= 6 (<= 64 tiles): one workgroup a step does the whole tree.
constexpr int kStrataTop = 6;
__host__ __device__ inline int strata_groups(int k)
{
    if (k <= kStrataTop) return 1;
    const int64_t g = ((int64_t)1 << k) / 128;
    return (int)(g < 1 ? 1 : (g > 64 ? 64 : g));
}
struct StrataArgs { uint64_t seed; int t0, k; uint32_t n_out; uint32_t* offs; uint32_t* top; uint32_t* top_clear; };   // offs: [steps][2^k + 1], top: [steps][64]; blockIdx.y = step t0 + y

// popcount of the bits [128 chunk, 128 chunk + 128) of a node's stream, cut at its n-th bit
__device__ __forceinline__ uint32_t strata_chunk_bits(uint64_t seed, uint64_t draw, uint64_t node_key, uint64_t chunk, uint64_t n)
{
    const u32x4 r = draw_block(seed, (node_key << 32) | chunk, draw);
    const uint64_t rem = n - chunk * 128;
    if (rem >= 128) return (uint32_t)(__popc(r.x) + __popc(r.y) + __popc(r.z) + __popc(r.w));
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
    uint32_t c = 0, left = (uint32_t)rem;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t take = left >= 32u ? 32u : left;
        const uint32_t m = take == 32u ? 0xffffffffu : ((1u << take) - 1u);
        c += (uint32_t)__popc(w[j] & m);
        left -= take;
    }
    return c;
}

// One workgroup splits the count in cnt[0] down `levels` levels, in place: node i of level l sits at cnt[i (stride >> l)], its heap
// index is (heap0 << l) + i, the n thresholds of a node go left with probability 1/2 each (left = popcount of the first n bits of the
// node's stream).  s_left: kThreads words of LDS.  cnt may be LDS or global memory (read and written by this workgroup alone).
__device__ __forceinline__ void strata_split(uint64_t seed, uint64_t draw, uint64_t key_hi, uint64_t heap0, int levels, uint64_t stride, uint32_t* cnt, uint32_t* s_left)
{
    const int tid = threadIdx.x;
    __syncthreads();
    for (int l = 0; l < levels; ++l) {
        const uint64_t nodes = 1ull << l, span = stride >> l, half = span >> 1;
        if (nodes <= (uint64_t)kThreads) {
            // a group of threads per node, chunks dealt round-robin, the group's popcounts meet in an LDS word
            const int G = kThreads >> l;
            const int i = tid / G, sub = tid % G;
            s_left[tid] = 0u;
            __syncthreads();
            const uint64_t n = cnt[(uint64_t)i * span];
            uint32_t part = 0;
            for (uint64_t ch = (uint64_t)sub; ch * 128 < n; ch += (uint64_t)G) part += strata_chunk_bits(seed, draw, key_hi | ((heap0 << l) + (uint64_t)i), ch, n);
            if (part) atomicAdd(&s_left[i], part);
            __syncthreads();
            if (sub == 0) { const uint32_t left = s_left[i]; cnt[(uint64_t)i * span] = left; cnt[(uint64_t)i * span + half] = (uint32_t)n - left; }
            __syncthreads();
        } else {
            for (uint64_t i = (uint64_t)tid; i < nodes; i += (uint64_t)kThreads) {
                const uint64_t n = cnt[i * span];
                uint32_t left = 0;
                for (uint64_t ch = 0; ch * 128 < n; ++ch) left += strata_chunk_bits(seed, draw, key_hi | ((heap0 << l) + i), ch, n);
                cnt[i * span] = left; cnt[i * span + half] = (uint32_t)n - left;
            }
            __syncthreads();
        }
    }
}

// counts -> first outputs: exclusive prefix sums of cnt[0 .. m) in place, on top of `base`; returns base + the sum (every thread)
__device__ __forceinline__ uint32_t strata_prefix(uint32_t* cnt, uint64_t m, uint32_t base, uint32_t* s_w)
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    uint32_t carry = base;
    for (uint64_t b0 = 0; b0 < m; b0 += (uint64_t)kThreads) {
        const uint64_t i = b0 + (uint64_t)tid;
        const uint32_t v = i < m ? cnt[i] : 0u;
        const uint32_t incl = wave_incl_scan_u32(v);
        if (lane == kWave - 1) s_w[wv] = incl;
        __syncthreads();
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const uint32_t x = s_w[w]; if (w < wv) off += x; tot += x; }
        if (i < m) cnt[i] = carry + off + incl - v;
        carry += tot;
        __syncthreads();
    }
    return carry;
}

// k <= 6: the whole tree of a step in one workgroup.  grid (1, steps)
__global__ __launch_bounds__(kThreads) void multinomial_strata_kernel(StrataArgs a)
{
    __shared__ uint32_t s_left[kThreads];
    __shared__ uint32_t s_w[kWaves];
    const uint64_t K = 1ull << a.k;
    uint32_t* o = a.offs + (size_t)blockIdx.y * (size_t)(K + 1);
    const uint64_t draw = kResampleDrawBase3 + (uint64_t)(a.t0 + (int)blockIdx.y);
    if (threadIdx.x == 0) o[0] = a.n_out;
    strata_split(a.seed, draw, 0, 1, a.k, K, o, s_left);
    const uint32_t tot = strata_prefix(o, K, 0u, s_w);
    if (threadIdx.x == 0) o[K] = tot;
}
// k > 6, top: grid (groups, steps)
__global__ __launch_bounds__(kThreads) void multinomial_strata_top_kernel(StrataArgs a)
{
    __shared__ uint32_t s_left[kThreads];
    __shared__ uint32_t s_cnt[64];
    const int g = (int)blockIdx.x, G = (int)gridDim.x;
    const uint64_t draw = kResampleDrawBase3 + (uint64_t)(a.t0 + (int)blockIdx.y);
    // outputs [g n / G, (g + 1) n / G)   (n < 2^32, G <= 64: the products fit 64 bits)
    const uint64_t n_g = ((uint64_t)a.n_out * (uint64_t)(g + 1)) / (uint64_t)G - ((uint64_t)a.n_out * (uint64_t)g) / (uint64_t)G;
    if (threadIdx.x < 64) s_cnt[threadIdx.x] = threadIdx.x == 0 ? (uint32_t)n_g : 0u;
    strata_split(a.seed, draw, (1ull << 31) | ((uint64_t)g << 8), 1, kStrataTop, 64, s_cnt, s_left);
    if (threadIdx.x < 64 && s_cnt[threadIdx.x]) atomicAdd(a.top + (size_t)blockIdx.y * 64 + threadIdx.x, s_cnt[threadIdx.x]);
}
// k > 6, bottom: grid (64, steps)
__global__ __launch_bounds__(kThreads) void multinomial_strata_bottom_kernel(StrataArgs a)
{
    __shared__ uint32_t s_left[kThreads];
    __shared__ uint32_t s_w[kWaves];
    __shared__ uint32_t s_base[2];
    const int i = (int)blockIdx.x;
    const uint64_t K = 1ull << a.k, sub = K >> kStrataTop;
    uint32_t* o = a.offs + (size_t)blockIdx.y * (size_t)(K + 1) + (size_t)i * sub;
    const uint64_t draw = kResampleDrawBase3 + (uint64_t)(a.t0 + (int)blockIdx.y);
    if (wave_id() == 0) {
        const uint32_t v = a.top[(size_t)blockIdx.y * 64 + threadIdx.x];
        const uint32_t incl = wave_incl_scan_u32(v);
        if ((int)threadIdx.x == i) { s_base[0] = incl - v; s_base[1] = v; }
        // (the totals of the run after this one: cleared here, one word a workgroup)
        if ((int)threadIdx.x == i && a.top_clear) a.top_clear[(size_t)blockIdx.y * 64 + i] = 0u;
    }
    __syncthreads();
    if (threadIdx.x == 0) o[0] = s_base[1];
    strata_split(a.seed, draw, 0, 64 + (uint64_t)i, a.k - kStrataTop, sub, o, s_left);
    const uint32_t end = strata_prefix(o, sub, s_base[0], s_w);
    if (i == 63 && threadIdx.x == 0) o[sub] = end;                        // offs[K] = the number of outputs
}

// Largest tile c whose exclusive prefix mass is <= x (the tile that holds mass unit x), with that prefix: top-down descent.
__device__ __forceinline__ int fhier_locate_mass(const HierTable* __restrict__ ht, int copy, uint64_t x, uint64_t& P)
{
    const int lane = lane_id();
    int blk = 0;
    uint64_t p = 0;
    for (int l = ht->n_lev - 1; l >= 0; --l) {
        const int idx = (blk << 6) + lane;
        uint64_t w = 0;
        const bool in = idx < ht->n_ent[l];
        if (in) w = ht->lvl[copy][l][(int64_t)idx * (l == 0 ? 1 : kHierStride)] & kMassMask;
        const uint64_t incl = wave_incl_scan_u64(w);
        const uint64_t e = p + incl - w;
        const unsigned long long m = __ballot(in && e <= x);
        const int child = m ? (63 - __builtin_clzll(m)) : 0;
        p = read_lane_u64(e, child);
        blk = (blk << 6) + child;
    }
    P = p;
    return blk;
}

// The SEARCH of the strata form (one wavefront): the strata w0 .. w1 of the outputs [s_first, s_last] -- the largest w with
// o_w <= s: a window of 64 offsets around the output tile's own place almost surely holds both -- and the source tiles that hold
// their mass range [B_w0, B_w1+1): probed around the output tile's own index like the systematic search (fixed_locate).
// One SHARD of a population (exchange scope): these nb tiles hold the mass range [before, before + own) of the population's S, and
// only the thresholds inside it are searched here -- the prefixes below are the shard's own, the prefix handed on is the population's.
struct StrataLocated { FLocated loc; int w0, w1; };
__device__ __forceinline__ StrataLocated strata_locate(const FHier& f, const uint32_t* __restrict__ offs, int k, int nb, int w_near, int guess, uint32_t s_first, uint32_t s_last,
                                                       uint64_t S, uint64_t before, uint64_t own, const ProbeWords* first)
{
    const int lane = lane_id();
    StrataLocated r;
    strata_window(offs, k, w_near, s_first, s_last, r.w0, r.w1);
    const uint64_t g_lo = strata_bound(S, (uint64_t)r.w0, k), g_hi = strata_bound(S, (uint64_t)r.w1 + 1, k);
    const uint64_t gl = g_lo > before ? g_lo : before, gh = g_hi < before + own ? g_hi : before + own;
    if (gh <= gl && !(own == S && before == 0)) { r.loc = FLocated{1, 0, 0}; return r; }          // no threshold of these outputs lies in this shard's mass
    const uint64_t x_lo = gl - before, b_hi = gh - before;
    const uint64_t x_hi = b_hi > x_lo ? b_hi - 1 : x_lo;
    int c = 0, c_last = nb;
    uint64_t P = 0;
    auto probe = [&](int at, const ProbeWords& pw) -> bool {
        const int cs = at > 0 ? at - 1 : 0;
        const uint64_t Pc = fhier_prefix_sum(cs, pw.lvl);
        const uint64_t we = (lane < 4 && cs + lane < nb) ? (pw.we & kMassMask) : 0ull;
        const uint64_t incl = wave_incl_scan_u64(we);
        const uint64_t x = Pc + incl - we;                               // lanes 0..4: the prefix at cs + lane
        const bool known = lane < 5 && cs + lane < nb;
        const unsigned long long m = __ballot(known && x <= x_lo);
        const int i_lo = m ? (63 - __builtin_clzll(m)) : -1;
        if ((i_lo >= 0 || cs == 0) && i_lo < 4) {
            const int i = i_lo < 0 ? 0 : i_lo;
            c = cs + i;
            P = read_lane_u64(x, i);
            const unsigned long long mh = __ballot(known && x <= x_hi);
            const int i_hi = mh ? (63 - __builtin_clzll(mh)) : i;
            c_last = (i_hi >= 4 && cs + 5 < nb) ? nb : cs + (i_hi > i ? i_hi : i);
            return true;
        }
        return false;
    };
    if (!(first && probe(guess, *first))) {
        // aim by the mass (tile masses are comparable), then descend from the top
        const double aim = u64_to_double(x_lo) * ((double)nb / u64_to_double(own > 0 ? own : 1));
        const int at = (int)fmin(fmax(aim, 0.0), (double)(nb - 1));
        ProbeWords pw;
        probe_fetch(f.h, at, nb, pw);
        if (!probe(at, pw)) { c = fhier_locate_mass(f.h.table, f.h.copy, x_lo, P); c_last = nb; }
    }
    r.loc = FLocated{c, c_last, P + before};
    return r;
}

// The WALK of the strata form (the whole workgroup): the lane's four outputs take their thresholds, every source tile of the range
// rebuilds its prefix masses in LDS (per wavefront: one scan, one barrier a tile), and the outputs whose threshold lies in the tile
// search them.  uid = the id of the lane's first output (a multiple of four); j0 = its index in the population.
// One shard of a population: j0 = the lane's first output's index in the POPULATION, [before, before + own) = the mass these
// sources hold; mine[i] = output i's threshold lies in it (its ancestor is in anc[i]); the others keep what anc[] held.
__device__ __forceinline__ void strata_walk(const uint32_t* __restrict__ offs, int k, const uint32_t* __restrict__ qprev, int nb, const StrataLocated& sl, uint64_t S,
                                            int64_t j0, uint64_t seed, uint64_t draw, uint64_t uid, int32_t (&anc)[kPPT], FixedLdsT<kFixMultinomial>& L,
                                            uint64_t before, uint64_t own, bool (&mine)[kPPT])
{
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    uint64_t v[kPPT], tau[kPPT];
    strata_bits4(seed, draw, uid, v);
#pragma unroll
    for (int i = 0; i < kPPT; ++i) v[i] <<= 11;
    bool live[kPPT];
#pragma unroll
    for (int i = 0; i < kPPT; ++i) { live[i] = false; tau[i] = 0; }
    const int w0 = __builtin_amdgcn_readfirstlane(sl.w0), w1 = __builtin_amdgcn_readfirstlane(sl.w1);
    {
        int ws[kPPT];
        lane_strata4(offs, k, w0, w1, (uint32_t)j0, ws, live);
#pragma unroll
        for (int i = 0; i < kPPT; ++i) {
            const uint64_t b_lo = strata_bound(S, (uint64_t)ws[i], k), b_hi = strata_bound(S, (uint64_t)ws[i] + 1, k);
            tau[i] = live[i] ? b_lo + __umul64hi(v[i], b_hi - b_lo) : 0ull;
        }
    }
#pragma unroll
    for (int i = 0; i < kPPT; ++i) { live[i] = live[i] && tau[i] >= before && tau[i] - before < own; mine[i] = live[i]; }
    const uint64_t b_end = strata_bound(S, (uint64_t)w1 + 1, k), x_first = strata_bound(S, (uint64_t)w0, k);
    const uint64_t x_hi = b_end > x_first ? b_end - 1 : x_first;
    int c = __builtin_amdgcn_readfirstlane(sl.loc.c);
    const int c_last = __builtin_amdgcn_readfirstlane(sl.loc.c_last);
    uint64_t P = sl.loc.P;
    int it = 0;
    // The usual case -- the probe told the last source tile and there are at most kStrataTiles of them: every tile's prefix masses
    // staged side by side behind ONE barrier (their weights fetched together), then each output searches once, in its own tile.
    if (c_last < nb && c_last - c + 1 <= kStrataTiles) {
        const int nt = c_last - c + 1;
        U4 raw[kStrataTiles];
#pragma unroll
        for (int j = 0; j < kStrataTiles; ++j)
            raw[j] = j < nt ? *reinterpret_cast<const U4*>(qprev + (int64_t)(c + j) * kTile + (int64_t)tid * kPPT) : U4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int j = 0; j < kStrataTiles; ++j) {
            if (j < nt) {
                uint64_t pre[kPPT];
                uint64_t run = 0;
#pragma unroll
                for (int i = 0; i < kPPT; ++i) { run += (uint64_t)raw[j][i]; pre[i] = run; }
                const uint64_t incl = wave_incl_scan_u34(run);
                const uint64_t excl = incl - run;
                uint64_t* Pm = L.mp + (size_t)j * kTile;
                using UL2 = unsigned long long __attribute__((ext_vector_type(2)));
                UL2 a0, a1;
                a0[0] = excl + pre[0]; a0[1] = excl + pre[1]; a1[0] = excl + pre[2]; a1[1] = excl + pre[3];
                *reinterpret_cast<UL2*>(Pm + (size_t)tid * kPPT) = a0;
                *reinterpret_cast<UL2*>(Pm + (size_t)tid * kPPT + 2) = a1;
                if (lane == kWave - 1) L.wtot[j][wv] = incl;
            }
        }
        __syncthreads();
        uint64_t tb[kStrataTiles + 1];                                    // prefix masses at the tiles' starts
        tb[0] = P;
#pragma unroll
        for (int j = 0; j < kStrataTiles; ++j) tb[j + 1] = tb[j] + (j < nt ? L.wtot[j][0] + L.wtot[j][1] + L.wtot[j][2] + L.wtot[j][3] : 0ull);
#pragma unroll
        for (int i = 0; i < kPPT; ++i) {
            if (live[i] && tau[i] >= tb[0] && tau[i] < tb[kStrataTiles]) {
                int j = 0;
#pragma unroll
                for (int jj = 1; jj < kStrataTiles; ++jj) j += tau[i] >= tb[jj] ? 1 : 0;
                uint64_t r = tau[i] - (j == 0 ? tb[0] : (j == 1 ? tb[1] : tb[2]));
                static_assert(kStrataTiles == 3, "tile starts selected by hand");
                const uint64_t w0s = L.wtot[j][0], w1s = L.wtot[j][1], w2s = L.wtot[j][2];
                int seg = 0;
                if (r >= w0s) { r -= w0s; seg = 1; if (r >= w1s) { r -= w1s; seg = 2; if (r >= w2s) { r -= w2s; seg = 3; } } }
                const uint64_t* Ps = L.mp + (size_t)j * kTile + seg * (kTile / kWaves);
                int a = 0, b = kTile / kWaves - 1;
#pragma unroll
                for (int h = 0; h < 8; ++h) { const int mid = (a + b) >> 1; if (Ps[mid] > r) b = mid; else a = mid + 1; }
                anc[i] = (c + j) * kTile + seg * (kTile / kWaves) + (a < kTile / kWaves ? a : kTile / kWaves - 1);
            }
        }
        return;
    }
    while (c < nb && c <= c_last) {
        if (c_last >= nb && __builtin_amdgcn_readfirstlane(P > x_hi ? 1 : 0)) break;          // the last tile is not known from the probe
        const U4 raw = *reinterpret_cast<const U4*>(qprev + (int64_t)c * kTile + (int64_t)tid * kPPT);
        uint64_t pre[kPPT];
        uint64_t run = 0;
#pragma unroll
        for (int i = 0; i < kPPT; ++i) { run += (uint64_t)raw[i]; pre[i] = run; }
        const uint64_t incl = wave_incl_scan_u34(run);
        const uint64_t excl = incl - run;
        uint64_t* Pm = L.mp + (size_t)(it & 1) * kTile;
        {
            using UL2 = unsigned long long __attribute__((ext_vector_type(2)));
            UL2 a0, a1;
            a0[0] = excl + pre[0]; a0[1] = excl + pre[1]; a1[0] = excl + pre[2]; a1[1] = excl + pre[3];
            *reinterpret_cast<UL2*>(Pm + (size_t)tid * kPPT) = a0;
            *reinterpret_cast<UL2*>(Pm + (size_t)tid * kPPT + 2) = a1;
        }
        if (lane == kWave - 1) L.scan[it & 1][wv] = incl;
        __syncthreads();
        uint64_t wsum[kWaves];
#pragma unroll
        for (int w = 0; w < kWaves; ++w) wsum[w] = L.scan[it & 1][w];
        static_assert(kWaves == 4, "four wavefront segments");
        const uint64_t M = wsum[0] + wsum[1] + wsum[2] + wsum[3];
#pragma unroll
        for (int i = 0; i < kPPT; ++i) {
            if (live[i] && tau[i] >= P && tau[i] - P < M) {
                uint64_t r = tau[i] - P;
                int seg = 0;
                if (r >= wsum[0]) { r -= wsum[0]; seg = 1; if (r >= wsum[1]) { r -= wsum[1]; seg = 2; if (r >= wsum[2]) { r -= wsum[2]; seg = 3; } } }
                const uint64_t* Ps = Pm + seg * (kTile / kWaves);
                int a = 0, b = kTile / kWaves - 1;                         // the first j with Ps[j] > r (Ps[255] = the segment's mass > r)
#pragma unroll
                for (int h = 0; h < 8; ++h) { const int mid = (a + b) >> 1; if (Ps[mid] > r) b = mid; else a = mid + 1; }
                static_assert(kTile / kWaves == 256, "eight halvings of a wavefront's 256 particles");
                anc[i] = c * kTile + seg * (kTile / kWaves) + (a < kTile / kWaves ? a : kTile / kWaves - 1);
                live[i] = false;
            }
        }
        P += M;
        ++c; ++it;
    }
}

// Bookkeeping of generation t-1 for the host (one thread): ESS, decision, evidence.  ref_prev = R_{t-1}.
__device__ __forceinline__ void fixed_bookkeep(StepCtrl* c, int t_prev, const FixedDecision& d, double ref_prev, double n_pop, double u0, double* ess_trace,
                                               int32_t* resampled, bool last, double m_prev)
{
    // how far the generation's heaviest particle sat below the reference its weights were taken against (NaN-proof: an empty
    // mass counts as the whole range) -- the host repeats a run whose gap cost too many bits in the floating-point form
    const double gap = (d.W > 0.0) ? ref_prev - m_prev : 1e300;
    c->fix_gap = (t_prev == 0) ? gap : fmax(c->fix_gap, gap);
    if (t_prev == 0) c->first_bad = -1;
    if (gap > kFixGapLimit && c->first_bad < 0) c->first_bad = t_prev;
    c->M = ref_prev; c->W = d.W; c->Q = d.Qd; c->ess = d.ess; c->do_resample = d.resample ? 1 : 0;
    c->cdf_lo = 0.0; c->w_local = d.W; c->scale = 1.0; c->u0 = u0; c->inv_stepw = d.inv * kFixScale; c->lw_after = 0.0; c->inv_global = d.inv * kFixScale;
    double lz = (t_prev == 0) ? 0.0 : c->log_z;
    int nr = (t_prev == 0) ? 0 : c->n_resampled;
    if (c->lz_trace) c->lz_trace[t_prev] = lz;
    if (d.resample || last) lz += ref_prev + log(d.W / n_pop);
    if (d.resample) nr += 1;
    c->log_z = lz; c->n_resampled = nr;
    if (ess_trace) ess_trace[t_prev] = d.ess;
    if (resampled) resampled[t_prev] = d.resample ? 1 : 0;
}

// PAIRED launches (CPPROB_HIP_FLAG_PAIRED_STEP_LAUNCH: an A/B form, measured and NOT the default -- profiles/r05_notes.md): a step is TWO launches back to back -- this body with
// RESAMPLING_ONLY, and smc_step_fixed_carry_body below, the step that resamples nothing -- each of which takes generation t-1's
// decision from its totals in every workgroup's first instructions and ends at once when the step is the other one's.  What that buys:
// the carry form has no search, walk, gather or ancestor row in it (half the registers, eight wavefronts a SIMD) and streams; what it
// costs: one launch of workgroups that read 24 bytes and leave.
template <class Model, bool SHARDED, bool PREFETCH, int RS = kFixSystematic, bool RESAMPLING_ONLY = false>
__device__ __forceinline__ void smc_step_fixed_body(const StepFixedArgs<Model>& a)
{
    using V = typename Model::value_t;
    using S = typename Model::store_t;
    constexpr bool kMulti = RS == kFixMultinomial || RS == kFixMultinomialLiteral;
    static_assert(!(kMulti && PREFETCH), "multinomial resampling: no source tiles to fetch ahead");
    static_assert(!(RS == kFixMultinomialLiteral && SHARDED), "the literal multinomial form serves one population per context");
    __shared__ FixedLdsT<RS> L;
    __shared__ __attribute__((aligned(16))) FixedFound s_found;
    __shared__ uint64_t s_red[3 * kWaves];
    const int tid = threadIdx.x;
    const int nb = (int)gridDim.x;
    const int bid = xcd_contiguous_tile((int)blockIdx.x, nb);
    const int64_t j0 = (int64_t)bid * kTile + (int64_t)tid * kPPT;
    const int t = a.t;
    const bool searcher = wave_id() == 0;
    CPH_STAMP(0);
    // everything the prologue reads is addressed by the launch geometry: fetched here, in one round trip under the random draws
    U4 q_0 = {0u, 0u, 0u, 0u}, q_m1 = q_0, q_p1 = q_0;
    FTotWords tw{};
    ProbeWords pw0{};
    uint64_t rs_ = 0, rq_ = 0, rm_ = 0;
    double lw_carry[kPPT];
    lane_fill(lw_carry, 0.0);
    // the source tile this output tile is expected to start in: its own index
    const int guess = bid;
    // (the count form moves this guess by the offset the previous exchange's plan left -- a dependent load in front of the entry
    //  fetches: here it costs more than the better aim returns: same-call A/B, eight loopback ranks, configs[3] 29.3 -> 28.1 ms,
    //  configs[4] 150.6 -> 143.0; the second probe aims by the miss distance anyway)
    if (t > 0) {
        if (PREFETCH) {
            const int64_t g0 = (int64_t)guess * kTile + (int64_t)tid * kPPT;
            q_0 = *reinterpret_cast<const U4*>(a.q_prev + g0);
            q_m1 = *reinterpret_cast<const U4*>(a.q_prev + (guess > 0 ? g0 - kTile : g0));
            q_p1 = *reinterpret_cast<const U4*>(a.q_prev + (guess + 1 < nb ? g0 + kTile : g0));
        }
        if (searcher) {
            ftot_fetch(a.f, tw);
            probe_fetch(a.f.h, guess, nb, pw0);
            if (SHARDED) { const int r = tid < a.world ? tid : 0; rs_ = a.all_totals[3 * r]; rq_ = a.all_totals[3 * r + 1]; rm_ = a.all_totals[3 * r + 2]; }
        }
    }
    // The variates: drawn here, under the round trip of the entry loads, where they are expensive (Box-Muller in fp64: the latency buys
    // them); behind the walk where they are a few integer operations (the discrete models), so that nothing of them is live across
    // the search and the walk -- measured both ways for both (profiles/r03_notes.md section 3).
    constexpr bool kDrawEarly = !Model::kIsInt;
    typename Model::Rand rnd[kPPT / 4];
    // Early draws where a search / decision follows (t > 0): the searching wavefront does not draw its own -- its chain is the workgroup's
    // serial part and starts at once, under the others' draws; wavefronts 1 and 3 draw one Box-Muller pair each of its lanes' four normals
    // (half a share more each: still ahead of "own draws, then the search") and hand them over through LDS behind the search's barrier.
    constexpr bool kHandOver = kDrawEarly && CPPROB_HAND_OVER_FIXED;
    __shared__ double s_z[kHandOver ? 4 : 1][kHandOver ? kWave : 1];
    const bool hand_over = kHandOver && t > 0 && (a.pid0 & 1ull) == 0;
    if constexpr (kDrawEarly) {
        if (!(hand_over && searcher)) {
#pragma unroll
            for (int q = 0; q < kPPT / 4; ++q) Model::draw4(a.seed, a.pid0 + (uint64_t)j0 + 4 * q, t, rnd[q]);
        }
        if constexpr (kHandOver) {
            if (hand_over && (wave_id() == 1 || wave_id() == 3)) {
                const int half = wave_id() == 1 ? 0 : 1;
                const uint64_t pid = a.pid0 + (uint64_t)((int64_t)bid * kTile + (int64_t)lane_id() * kPPT) + 2 * (uint64_t)half;
                double z0, z1;
                Model::draw2(a.seed, pid, t, z0, z1);
                s_z[2 * half][lane_id()] = z0; s_z[2 * half + 1][lane_id()] = z1;
            }
        }
    }
    // Late draws (the discrete models: a few integer operations whose registers are better not live across the search and the walk) where
    // a decision / search follows: drawn HERE all the same, by the three wavefronts that wait for the searching one (wavefront 2 also its
    // share), and parked in LDS -- no register is live, and the Philox block leaves the chain behind the barrier.
    constexpr bool kPark = !kDrawEarly && CPPROB_PARK_DRAWS && sizeof(typename Model::Rand) == 16 && kPPT == 4;
    __shared__ __attribute__((aligned(16))) typename Model::Rand s_park[kPark ? kThreads : 1];
    const bool park = kPark && t > 0;
    if constexpr (kPark) {
        if (park && !searcher) {
            typename Model::Rand r1;
            Model::draw4(a.seed, a.pid0 + (uint64_t)j0, t, r1);
            s_park[tid] = r1;
            if (wave_id() == 2) {
                Model::draw4(a.seed, a.pid0 + (uint64_t)((int64_t)bid * kTile + (int64_t)lane_id() * kPPT), t, r1);
                s_park[lane_id()] = r1;
            }
        }
    }
    CPH_STAMP(1);

    int32_t anc[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) anc[k] = (int32_t)(j0 + k);
    bool resample = false;
    double ref = a.bound;                                              // R_0 = B_0
    if (t > 0) {
        if (guess == 0) q_m1 = U4{0u, 0u, 0u, 0u};
        if (guess + 1 >= nb) q_p1 = U4{0u, 0u, 0u, 0u};
        if constexpr (!kMulti) {
            int32_t neg[kPPT];
            lane_fill(neg, (int32_t)-1);
            store4(L.slot, (int64_t)tid * kPPT, neg);
        }
        const int64_t rem = a.n - (int64_t)bid * kTile;
        const int n_out = rem < kTile ? (int)rem : kTile;
        const double gj_first = SHARDED ? (double)(a.pid0 + (uint64_t)bid * kTile) : (double)((uint64_t)bid * kTile);
        const bool last_shard = SHARDED ? a.rank + 1 == a.world : true;
        FixedCdf fc;
        fc.u0 = a.u0; fc.n_pop = a.n_pop; fc.base = 0; fc.inv = 0.0;
        // (the resampling's own uniforms: ids are population-wide in a sharded run; a population of its own offsets them by pid0,
        //  which there only selects streams)
        fc.seed = a.seed; fc.draw = kResampleDrawBase + (uint64_t)t; fc.uid0 = SHARDED ? 0 : a.pid0;
        // (stratified: the outputs' uniforms are staged here, under the entry loads, where every step resamples; on a schedule where steps
        //  may not, behind the decision -- a second barrier on the steps that do resample, no Philox block on those that do not)
        if constexpr (RS == kFixStratified) { if (!a.may_carry) stratified_stage(L, fc.seed, fc.draw, fc.uid0 + (uint64_t)gj_first); }
        if (searcher) {
            uint64_t St, Qt; double Mt; uint64_t before = 0;
            const FTot own = ftot_sum(a.f, tw);
            if (SHARDED) {
                const int lane = tid;
                if (lane >= a.world) { rs_ = 0; rq_ = 0; rm_ = 0; }
                St = wave_sum_u64(rs_); Qt = wave_sum_u64(rq_); Mt = dkey_inv(wave_max_u64(rm_)); before = wave_sum_u64(lane < a.rank ? rs_ : 0ull);
            } else { St = own.S; Qt = own.Q; Mt = own.M; }
            const FixedDecision d = fixed_decide(St, Qt, a.n_pop, a.ess_frac, true);        // (generation t-1 is never the last one here)
            fc.inv = d.inv; fc.base = before;
            const double r_t = fixed_reference(d.resample, Mt, a.bound);
            if (bid == 0 && tid == 0 && (!RESAMPLING_ONLY || d.resample)) {          // (paired launches: the books are kept by the launch whose step it is)
                StepCtrl* c = a.ctrl;
                fixed_bookkeep(c, t - 1, d, c->ref_cur, a.n_pop, a.u0, a.ess_trace, a.resampled, false, Mt);
                c->ref_cur = r_t;
            }
            FLocated loc{0, 0, 0};
            int64_t l0 = 0, l1 = a.n;
            int sw0 = 0, sw1 = 0;
            if (d.resample && RS == kFixMultinomial) {
                // (a shard: the strata, their outputs and the thresholds are the POPULATION's; only those inside this shard's mass are searched here)
                const uint64_t gfirst = SHARDED ? a.pid0 + (uint64_t)bid * kTile : (uint64_t)bid * kTile;
                const int64_t nb_pop = SHARDED ? ((int64_t)a.n_pop + kTile - 1) / kTile : (int64_t)nb;
                const StrataLocated sl = strata_locate(a.f, a.strata_offs, a.strata_k, nb, strata_near(a.strata_k, nb_pop, (int64_t)(gfirst / kTile)), bid, (uint32_t)gfirst,
                                                       (uint32_t)(gfirst + (uint64_t)n_out - 1), St, before, SHARDED ? own.S : St, &pw0);
                loc = sl.loc; sw0 = sl.w0; sw1 = sl.w1;
            } else if (d.resample && !kMulti) {
                loc = fixed_locate<RS>(a.f, fc, nb, gj_first, n_out, guess, &pw0);
                if (SHARDED) {
                    // outputs below o_lo / at or beyond o_hi descend from other shards' sources
                    const double o_lo = fc.template first<RS>(0), o_hi = last_shard ? a.n_pop : fc.template first<RS>(own.S);
                    const double sb = (double)a.pid0;
                    l0 = (int64_t)fmin(fmax(o_lo - sb, 0.0), (double)a.n); l1 = (int64_t)fmin(fmax(o_hi - sb, 0.0), (double)a.n);
                }
            }
            if (tid == 0) { s_found.loc = loc; s_found.inv = d.inv; s_found.ref = r_t; s_found.base = before; s_found.S = St; s_found.own = SHARDED ? own.S : St; s_found.l0 = l0; s_found.l1 = l1; s_found.resample = d.resample ? 1 : 0; s_found.w0 = sw0; s_found.w1 = sw1; }
        }
        __syncthreads();                                               // slots reset, search results in place
        if constexpr (kHandOver) {
            if (hand_over && searcher) {
#pragma unroll
                for (int k = 0; k < 4; ++k) rnd[0].z[k] = s_z[k][lane_id()];
            }
        }
    CPH_STAMP(2);
        resample = s_found.resample != 0;
        if (RESAMPLING_ONLY && !resample) return;                      // this step carries its weights: the other launch's
        ref = s_found.ref;
        if (resample && RS == kFixMultinomialLiteral) {
            fixed_multinomial_ancestors(s_found.S, fc.seed, fc.draw, fc.uid0 + (uint64_t)j0, a.tile_prefix, nb, a.lane_prefix_prev, a.q_prev, anc);
        } else if (resample && RS == kFixMultinomial) {
            if constexpr (RS == kFixMultinomial) {
                StrataLocated sl;
                sl.loc = s_found.loc; sl.w0 = s_found.w0; sl.w1 = s_found.w1;
                bool mine[kPPT];
                const uint64_t gj = SHARDED ? a.pid0 + (uint64_t)j0 : (uint64_t)j0;        // the lane's first output in the population
                strata_walk(a.strata_offs, a.strata_k, a.q_prev, nb, sl, s_found.S, (int64_t)gj, fc.seed, kResampleDrawBase2 + (uint64_t)t, a.pid0 + (uint64_t)j0, anc, L,
                            s_found.base, s_found.own, mine);
                if constexpr (SHARDED) {
                    // an output whose threshold lies in another rank's mass: its ancestor arrived as an annex column, in output order
                    bool out = false;
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) out = out || (!mine[k] && j0 + k < a.n);
                    if (__any(out)) {
                        const KeptCtx kc = kept_ctx(a.cut, a.rank, (uint32_t)a.pid0);
                        const int64_t col0 = a.ld + a.annex_base[t - 1];
#pragma unroll
                        for (int k = 0; k < kPPT; ++k)
                            if (!mine[k] && j0 + k < a.n) anc[k] = (int32_t)(col0 + (j0 + k) - (int64_t)kept_before(kc, (uint32_t)(a.pid0 + (uint64_t)(j0 + k))));
                    }
                }
            }
        } else if (resample) {
            fc.inv = s_found.inv; fc.base = s_found.base;
            if constexpr (RS == kFixStratified) { if (a.may_carry) { stratified_stage(L, fc.seed, fc.draw, fc.uid0 + (uint64_t)gj_first); __syncthreads(); } }
            fixed_walk<RS>(fc, a.q_prev, a.n, nb, last_shard, gj_first, n_out, s_found.loc, guess, PREFETCH, q_m1, q_0, q_p1, anc, L);
            if (SHARDED) {
                const int64_t l0 = s_found.l0, l1 = s_found.l1;
                const int64_t col0 = a.ld + a.annex_base[t - 1];
#pragma unroll
                for (int k = 0; k < kPPT; ++k) {
                    const int64_t j = j0 + k;
                    if (j < l0) anc[k] = (int32_t)(col0 + j);
                    else if (j >= l1 && j < a.n) anc[k] = (int32_t)(col0 + l0 + (j - l1));
                }
            }
#pragma unroll
            for (int k = 0; k < kPPT; ++k) anc[k] = max(anc[k], 0);   // padding outputs of the last tile
            // (equal weights after resampling: lw_carry stays 0)
        } else if (a.may_carry) {
            load4(a.logw_prev, j0, lw_carry);                           // weights carry over (a launch that resamples never reads them)
        }
    } else if (bid == 0 && tid == 0) {
        a.ctrl->ref_cur = ref;
    }

    CPH_STAMP(3);
    const S* prev_row = a.values + (int64_t)a.row_r * a.rs;
    if constexpr (!kDrawEarly) {
        if (park) rnd[0] = s_park[kPark ? tid : 0];                     // (behind the decision's barrier)
        else {
#pragma unroll
            for (int q = 0; q < kPPT / 4; ++q) Model::draw4(a.seed, a.pid0 + (uint64_t)j0 + 4 * q, t, rnd[q]);
        }
    }
    V prev[kPPT], x[kPPT];
    if (t > 0 && !resample) load4_as(prev_row, j0, prev);                                    // every slot extends itself: one vector load
    else {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) prev[k] = t > 0 ? static_cast<V>(prev_row[anc[k]]) : V(0);             // ancestor's state (sorted gather)
    }
#pragma unroll
    for (int q = 0; q < kPPT / 4; ++q)                                                        // sample #t
        Model::apply4(a.mp, t, rnd[q], reinterpret_cast<const V(&)[4]>(prev[4 * q]), reinterpret_cast<V(&)[4]>(x[4 * q]));
    store4_as(a.values + (int64_t)a.row_w * a.rs, j0, x);                                     // predict #t
    CPH_STAMP(4);
    // (a step that follows no resampling extends every slot by itself: nobody walks that row -- the read-out, the exchange packing
    //  and the skip rows test resampled[t-1] first, and cpprob_hip_copy_ancestors writes the identity on its way out)
    if (a.anc && (t == 0 || resample)) store4_write_through(a.anc + (int64_t)t * a.rs, j0, anc);

    // ---- observe #t: log-weights, integer weights, the tile's mass / squares / maximum ----
    double lw[kPPT];
    U4 q;
    uint64_t s_l = 0, q_l = 0;
    double m_l = -INFINITY;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        lw[k] = lw_carry[k] + Model::loglik(a.mp, x[k], t, a.obs);
        if (j0 + k >= a.n) lw[k] = -INFINITY;                                                  // padding slots
        const uint32_t w = fix_weight(lw[k], ref);
        q[k] = w;
        s_l += w; q_l += fix_square(w);
        m_l = fmax(m_l, lw[k]);
    }
    // (the tile's totals by DPP reductions: three same-address 64-bit LDS atomics per lane were measured at 1.7x the whole step);
    // published BEFORE this workgroup's weight / log-weight stores are issued: the hierarchy's atomics -- and, above 4096 tiles, the
    // wait in front of the arrival count -- travel under them instead of behind them
    CPH_STAMP(5);
    const uint64_t s_incl = RS == kFixMultinomialLiteral ? wave_incl_scan_u34(s_l) : 0ull;              // (literal multinomial: the lanes' prefixes)
    const uint64_t s_w = RS == kFixMultinomialLiteral ? read_lane_u64(s_incl, kWave - 1) : wave_sum_u34(s_l);
    const uint64_t q_w = wave_sum_u34(q_l), m_w = wave_max_key(dkey(m_l));                              // (sums of four 32-bit terms)
    if (lane_id() == 0) { s_red[wave_id()] = s_w; s_red[kWaves + wave_id()] = q_w; s_red[2 * kWaves + wave_id()] = m_w; }
    __syncthreads();
    if constexpr (RS == kFixMultinomialLiteral) {
        uint64_t off = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) if (w < wave_id()) off += s_red[w];
        a.lane_prefix_next[(int64_t)bid * kThreads + tid] = off + s_incl;
    }
    if (tid == 0) {
        uint64_t St = 0, Qt = 0, Mk = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { St += s_red[w]; Qt += s_red[kWaves + w]; Mk = umax64(Mk, s_red[2 * kWaves + w]); }
        fhier_publish(a.f, bid, nb, St, Qt, Mk);
    }
    CPH_STAMP(6);
    *reinterpret_cast<U4*>(a.q_next + j0) = q;
    if (a.may_carry || t + 1 == a.T) store4(a.logw_next, j0, lw);
#ifdef CPPROB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CPH_STAMP(7);
#endif
}

// The step that resamples nothing (t > 0, a schedule on which weights may carry): every slot extends itself.  Each wavefront takes the
// decision itself (no barrier in front of it), then streams: states and log-weights in, states, integer weights and log-weights out.
// (A form whose workgroups follow several tiles with the next tile's inputs in flight was built and measured: the loop keeps the
//  polynomial constants live -- 40 -> 94 registers -- and the launch took 87 us against this form's 70: profiles/r05_notes.md.)
template <class Model, bool SHARDED>
__device__ __forceinline__ void smc_step_fixed_carry_body(const StepFixedArgs<Model>& a)
{
    using V = typename Model::value_t;
    __shared__ uint64_t s_red[3 * kWaves];
    const int tid = threadIdx.x, lane = lane_id();
    const int nb = (int)gridDim.x;
    const int bid = xcd_contiguous_tile((int)blockIdx.x, nb);
    const int64_t j0 = (int64_t)bid * kTile + (int64_t)tid * kPPT;
    const int t = a.t;
    uint64_t St, Qt; double Mt;
    if (SHARDED) {
        uint64_t rs_ = 0, rq_ = 0, rm_ = 0;
        if (lane < a.world) { rs_ = a.all_totals[3 * lane]; rq_ = a.all_totals[3 * lane + 1]; rm_ = a.all_totals[3 * lane + 2]; }
        St = wave_sum_u64(rs_); Qt = wave_sum_u64(rq_); Mt = dkey_inv(wave_max_u64(rm_));
    } else { const FTot own = ftot(a.f); St = own.S; Qt = own.Q; Mt = own.M; }
    const FixedDecision d = fixed_decide(St, Qt, a.n_pop, a.ess_frac, true);
    if (d.resample) return;                                            // this step resamples: the other launch's
    const double ref = fixed_reference(false, Mt, a.bound);
    if (bid == 0 && tid == 0) {
        StepCtrl* c = a.ctrl;
        fixed_bookkeep(c, t - 1, d, c->ref_cur, a.n_pop, a.u0, a.ess_trace, a.resampled, false, Mt);
        c->ref_cur = ref;
    }
    double lw_carry[kPPT];
    load4(a.logw_prev, j0, lw_carry);
    V prev[kPPT], x[kPPT];
    load4_as(a.values + (int64_t)a.row_r * a.rs, j0, prev);
    typename Model::Rand rnd[kPPT / 4];
#pragma unroll
    for (int q = 0; q < kPPT / 4; ++q) Model::draw4(a.seed, a.pid0 + (uint64_t)j0 + 4 * q, t, rnd[q]);
#pragma unroll
    for (int q = 0; q < kPPT / 4; ++q)
        Model::apply4(a.mp, t, rnd[q], reinterpret_cast<const V(&)[4]>(prev[4 * q]), reinterpret_cast<V(&)[4]>(x[4 * q]));
    store4_as(a.values + (int64_t)a.row_w * a.rs, j0, x);
    double lw[kPPT];
    U4 q;
    uint64_t s_l = 0, q_l = 0;
    double m_l = -INFINITY;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        lw[k] = lw_carry[k] + Model::loglik(a.mp, x[k], t, a.obs);
        if (j0 + k >= a.n) lw[k] = -INFINITY;
        const uint32_t w = fix_weight(lw[k], ref);
        q[k] = w;
        s_l += w; q_l += fix_square(w);
        m_l = fmax(m_l, lw[k]);
    }
    const uint64_t s_w = wave_sum_u34(s_l), q_w = wave_sum_u34(q_l), m_w = wave_max_key(dkey(m_l));
    if (lane == 0) { s_red[wave_id()] = s_w; s_red[kWaves + wave_id()] = q_w; s_red[2 * kWaves + wave_id()] = m_w; }
    __syncthreads();
    if (tid == 0) {
        uint64_t S2 = 0, Q2 = 0, Mk = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { S2 += s_red[w]; Q2 += s_red[kWaves + w]; Mk = umax64(Mk, s_red[2 * kWaves + w]); }
        fhier_publish(a.f, bid, nb, S2, Q2, Mk);
    }
    *reinterpret_cast<U4*>(a.q_next + j0) = q;
    store4(a.logw_next, j0, lw);
}
template <class Model, bool SHARDED>
__global__ __launch_bounds__(kThreads) void smc_step_fixed_carry_kernel(StepFixedArgs<Model> a) { smc_step_fixed_carry_body<Model, SHARDED>(a); }
// ... and its partner: the step that does resample (systematic / stratified / strata-form multinomial), no prefetch of source tiles
template <class Model, bool SHARDED, int RS>
__global__ __launch_bounds__(kThreads) void smc_step_fixed_resampling_kernel(StepFixedArgs<Model> a)
{ smc_step_fixed_body<Model, SHARDED, false, RS, true>(a); }

// One population on this GPU.  PREFETCH: the three likely source tiles' weights are fetched at kernel entry (every-step schedules).
template <class Model, bool PREFETCH, int RS = kFixSystematic>
__global__ __launch_bounds__(kThreads) void smc_step_fixed_kernel(StepFixedArgs<Model> a) { smc_step_fixed_body<Model, false, PREFETCH, RS>(a); }
// The continuous models without the prefetch sit one register above five wavefronts a SIMD: told to fit (one spilled dword), 1221
// workgroups -- 1.25 10^6 particles -- run in one pass.
template <class Model, int RS = kFixSystematic>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5))) void smc_step_fixed_five_kernel(StepFixedArgs<Model> a) { smc_step_fixed_body<Model, false, false, RS>(a); }
// One shard of a joint population.  Five wavefronts a SIMD (96 registers; the continuous models' build wants 138 and spills ~25
// of them): configs[3]'s shard of 1.25 10^6 particles is 1221 workgroups, and 256 CUs hold 1280 of them at five a CU but 768 at
// three -- a second pass of workgroups behind the first costs more than the spills (profiles/r03_notes.md).
template <class Model, bool PREFETCH, int RS = kFixSystematic>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(5))) void smc_step_fixed_sharded_kernel(StepFixedArgs<Model> a) { smc_step_fixed_body<Model, true, PREFETCH, RS>(a); }

// ---- the run's last generation ------------------------------------------------------------------------------------------------
struct FixedFinal {
    FHier f;                     // the final generation's masses (the copy the last step wrote)
    double n_pop, ess_frac; int T;
    int bookkeep;
    StepCtrl* ctrl; double* ess_trace; int32_t* resampled;
};
__device__ __forceinline__ void fixed_final_bookkeep(const FixedFinal& ff, uint64_t S, uint64_t Q, double M)
{
    const FixedDecision d = fixed_decide(S, Q, ff.n_pop, ff.ess_frac, false);
    fixed_bookkeep(ff.ctrl, ff.T - 1, d, ff.ctrl->ref_cur, ff.n_pop, 0.0, ff.ess_trace, ff.resampled, true, M);
}
// one shard of a joint population: from the all-gathered totals (the sums a single GPU's hierarchy would hold)
__global__ __launch_bounds__(kWave) void fixed_final_ctrl_kernel(FixedFinal ff, const uint64_t* __restrict__ all_totals, int world)
{
    const FixedRanks r = fixed_ranks(all_totals, world, 0);
    if (threadIdx.x == 0) fixed_final_bookkeep(ff, r.S, r.Q, r.M);
}
// the exact maximum of a generation's recomputed log-weights as a rank's contribution to the repair's all-gather: {key(M), 0, 0}
__global__ __launch_bounds__(kWave) void fixed_repair_max_kernel(FHier f, uint64_t* __restrict__ out)
{
    const double m = bbf_top_max(f);
    if (threadIdx.x == 0) { out[0] = dkey(m); out[1] = 0; out[2] = 0; }
}
// {S, Q, key(M)} of this shard's generation: what a sharded run all-gathers between two steps (24 bytes)
__global__ __launch_bounds__(kWave) void fixed_totals_kernel(FHier f, uint64_t* __restrict__ out)
{
    const FTot t = ftot(f);
    if (threadIdx.x == 0) { out[0] = t.S; out[1] = t.Q; out[2] = dkey(t.M); }
}

// A filtering-only run has no lineages to walk: the final generation's bookkeeping is the whole read-out.
__global__ __launch_bounds__(kWave) void fixed_filter_final_kernel(FixedFinal ff)
{
    const FTot t = ftot(ff.f);
    if (threadIdx.x == 0) fixed_final_bookkeep(ff, t.S, t.Q, t.M);
}

// Filtering-only runs: predict hit t's sums under generation t's own integer weights, per workgroup {0, sum q, sum q f_0 .. f_{K-1}}
// in filter_partials_kernel's layout (kernels.hpp; every workgroup's reference is the same, so filter_finalize_kernel's rescaling is 1).
template <class Model>
__global__ __launch_bounds__(kThreads) void filter_partials_fixed_kernel(const typename Model::store_t* __restrict__ row, const uint32_t* __restrict__ q,
                                                                          int64_t n, double* __restrict__ fpart_t)
{
    using V = typename Model::value_t;
    constexpr int K = Model::kStats;
    __shared__ double s_sum[(K + 1) * kWaves];
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int64_t ntiles = (n + kTile - 1) / kTile;
    double acc[K + 1];
#pragma unroll
    for (int j = 0; j <= K; ++j) acc[j] = 0.0;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t j0 = tile * kTile + (int64_t)tid * kPPT;
        const U4 w = *reinterpret_cast<const U4*>(q + j0);                 // padding slots: 0
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const double e = (double)w[k] * kFixInv;
            acc[0] += e;
            Model::accumulate(static_cast<V>(row[j0 + k]), e, reinterpret_cast<double(&)[K]>(acc[1]));
        }
    }
#pragma unroll
    for (int j = 0; j <= K; ++j) { const double w = wave_sum(acc[j]); if (lane == 0) s_sum[j * kWaves + wv] = w; }
    __syncthreads();
    if (tid <= K) {
        double t = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kWaves; ++w2) t += s_sum[tid * kWaves + w2];
        fpart_t[(int64_t)(tid + 1) * gridDim.x + blockIdx.x] = t;
    }
    if (tid == 0) fpart_t[blockIdx.x] = 0.0;
}

// The lineage walk of a DISCRETE model in the fixed-point form, in integers: the final weights are the integers q_i and a predict
// hit's statistic is sum_i q_i [x_t(lineage i) = s] -- exact in 64 bits, so a row's sums are taken as integers: per lane the masses
// of the states 0 .. K-2 (the last state's follows from the lane's total, which no row changes), per wavefront two single-instruction
// DPP scans a state (18-bit halves of values below 2^36: eight lineages of 32-bit weights) instead of a six-stage fp64 reduction --
// the walk is bound by vector issue (r04: ~55 %), and the reductions were half of its instructions.  smooth_body's tile order.
// Tiles a workgroup of the INTEGER walk follows at a time: sixteen lineages a lane.  The kernel needs 56 registers (eight wavefronts a
// SIMD with room to spare) and lives on gathers in flight: hmm<128> at 1.25e7 particles, us per launch, 2 / 3 / 4 tiles: 1002 / 985 / 967.
#ifndef CPPROB_SMOOTH_TILES_INT
#define CPPROB_SMOOTH_TILES_INT 4
#endif
constexpr int kSmoothTilesInt = CPPROB_SMOOTH_TILES_INT;
// PATHS: the traces are materialised on the way (dumps / tests) -- a build of its own, so that the walk proper carries no test per lineage
// and row.  Slots are UNSIGNED 32-bit offsets from a row's base: a gather is the load alone (scalar base + vector offset), no 64-bit
// address arithmetic in vector registers.
template <class Model, bool PATHS>
__device__ __forceinline__ void smooth_body_fixed_int(const SmoothArgs<Model>& a, double* s_stat, const uint32_t* __restrict__ q_last)
{
    using V = typename Model::value_t;
    constexpr int K = Model::kStats;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int TK = a.T * K;
    // (the accumulators are INTEGERS -- the weights are: a wavefront's row sums go into them by one LDS add each, exact whatever the
    //  order, and become doubles once, when the workgroup's partial leaves: no conversion, no fp64 add per row and state)
    unsigned long long* s_int = reinterpret_cast<unsigned long long*>(s_stat);
    for (int i = tid; i < kWaves * TK; i += kThreads) s_int[i] = 0ull;
    __syncthreads();
    const int64_t ntiles = (a.n + kTile - 1) / kTile;
    auto wave_sum_u36 = [](uint64_t v) -> uint64_t {
        const uint32_t lo = wave_sum_u32((uint32_t)v & 0x3ffffu), hi = wave_sum_u32((uint32_t)(v >> 18));
        return ((uint64_t)hi << 18) + lo;
    };
    auto walk = [&](auto tiles_tag) {
    constexpr int NT = decltype(tiles_tag)::value;
    static_assert(NT * kPPT <= 16, "a lane's row sums stay below 2^36");
    for (int64_t tile0 = ntiles <= 2048 ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x; tile0 < ntiles; tile0 += (int64_t)NT * gridDim.x) {
        constexpr int L = NT * kPPT;
        uint32_t idx[L]; uint32_t w[L];
        uint64_t w_lane = 0;
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int64_t tile = tile0 + (int64_t)(k / kPPT) * gridDim.x;
            const int64_t i = tile * kTile + (int64_t)(k % kPPT) * kThreads + tid;
            const bool on = tile < ntiles;
            idx[k] = on ? (uint32_t)i : 0u;
            w[k] = on ? q_last[i] : 0u;                                    // padding slots: q = 0
            w_lane += w[k];
        }
        const uint64_t w_wave = wave_sum_u36(w_lane);
        for (int t = a.T - 1; t >= 0; --t) {
            uint64_t acc[K - 1];
#pragma unroll
            for (int j = 0; j < K - 1; ++j) acc[j] = 0;
            const typename Model::store_t* row = a.values + (int64_t)t * a.rs;
#pragma unroll
            for (int k = 0; k < L; ++k) {
                const V x = static_cast<V>(row[idx[k]]);
#pragma unroll
                for (int j = 0; j < K - 1; ++j) acc[j] += (int)x == j ? (uint64_t)w[k] : 0ull;
                const int64_t tile = tile0 + (int64_t)(k / kPPT) * gridDim.x;
                if (PATHS && tile < ntiles) a.paths[(int64_t)t * a.ld + tile * kTile + (int64_t)(k % kPPT) * kThreads + tid] = x;
            }
            uint64_t rest = w_wave;
#pragma unroll
            for (int j = 0; j < K - 1; ++j) { acc[j] = wave_sum_u36(acc[j]); rest -= acc[j]; }
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < K - 1; ++j) __hip_atomic_fetch_add(s_int + wv * TK + t * K + j, (unsigned long long)acc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(s_int + wv * TK + t * K + K - 1, (unsigned long long)rest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (t > 0 && !a.identity && a.resampled[t - 1]) {
                const int32_t* arow = a.anc + (int64_t)t * a.rs;
#pragma unroll
                for (int k = 0; k < L; ++k) idx[k] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(arow) + (idx[k] << 2));   // (fixed-point runs: n <= 2^28, the byte offset fits 32 bits)
            }
        }
    }
    };
    if (kSmoothTilesInt > 1 && ntiles > (int64_t)gridDim.x) walk(std::integral_constant<int, kSmoothTilesInt>{});
    else walk(std::integral_constant<int, 1>{});
    __syncthreads();
    for (int i = tid; i < TK; i += kThreads) {
        unsigned long long s2 = 0ull;
#pragma unroll
        for (int w2 = 0; w2 < kWaves; ++w2) s2 += s_int[w2 * TK + i];
        a.stats_part[(int64_t)i * gridDim.x + blockIdx.x] = u64_to_double((uint64_t)s2) * kFixInv;
    }
}

// Read-out: the final weight of particle i is q_i 2^-32 (relative to exp(R)); the normaliser is the final generation's mass.
template <class Model>
__global__ __launch_bounds__(kThreads) void smooth_fixed_kernel(SmoothArgs<Model> a, FixedFinal ff, const uint32_t* __restrict__ q_last)
{
    extern __shared__ __attribute__((aligned(16))) double s_stat[];   // [kWaves][T*K]
    if (ff.bookkeep && blockIdx.x == 0 && wave_id() == 0) {
        const FTot t = ftot(ff.f);
        if (threadIdx.x == 0) fixed_final_bookkeep(ff, t.S, t.Q, t.M);
    }
    smooth_body<Model>(a, s_stat, [q_last](int64_t, int64_t i) { return (double)q_last[i] * kFixInv; });        // padding slots: q = 0
}
// ... discrete models whose lineages stay on this device: the integer walk, a kernel of its own (the general body's registers -- remote
// stores, fp64 sums -- would cost this one two wavefronts a SIMD, and the walk lives on loads in flight); the host picks it (launch_smooth).
template <class Model, bool PATHS>
__global__ __launch_bounds__(kThreads) void smooth_fixed_int_kernel(SmoothArgs<Model> a, FixedFinal ff, const uint32_t* __restrict__ q_last)
{
    extern __shared__ __attribute__((aligned(16))) double s_stat[];   // [kWaves][T*K] (integers: smooth_body_fixed_int)
    if (ff.bookkeep && blockIdx.x == 0 && wave_id() == 0) {
        const FTot t = ftot(ff.f);
        if (threadIdx.x == 0) fixed_final_bookkeep(ff, t.S, t.Q, t.M);
    }
    smooth_body_fixed_int<Model, PATHS>(a, s_stat, q_last);
}

// ---- repair of a generation whose weights lost their bits (cpprob_hip.hip: settle_fixed) -------------------------------------------
// Generation g's log-weights again, from the particle store: lw_i = sum over the steps s = s0 .. g since the last resampling of the
// step's log-likelihood at values[s][i] (slots extend themselves between two resamplings) -- the very sums the step kernels carried,
// in their order.  Padding slots: -inf.
template <class Model>
__global__ __launch_bounds__(kThreads) void fixed_relogw_kernel(ModelParams mp, const double* __restrict__ obs, const typename Model::store_t* __restrict__ values, int64_t rs,
                                                                 int s0, int g, int64_t n, int64_t ld, double* __restrict__ logw)
{
    using V = typename Model::value_t;
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= ld) return;
    double lw = 0.0;
    for (int s = s0; s <= g; ++s) lw = lw + Model::loglik(mp, static_cast<V>(values[(int64_t)s * rs + i]), s, obs);
    logw[i] = i < n ? lw : -INFINITY;
}
// The books as they stood before generation g was weighed against its reference, with the exact maximum in that reference's place.
__global__ __launch_bounds__(kWave) void fixed_repair_ctrl_kernel(StepCtrl* c, FHier f, int g, const int32_t* __restrict__ resampled,
                                                                  const uint64_t* __restrict__ all_keys = nullptr, int world = 0)
{
    // (one shard of a joint population: the population's exact maximum, from the ranks' all-gathered keys)
    const double m = all_keys ? dkey_inv(wave_max_u64(lane_id() < world ? all_keys[3 * lane_id()] : 0ull)) : bbf_top_max(f);
    if (threadIdx.x == 0) {
        c->ref_cur = m;
        c->log_z = c->lz_trace[g];
        int nr = 0;
        for (int s = 0; s < g; ++s) nr += resampled[s];
        c->n_resampled = nr;
        c->fix_gap = 0.0; c->first_bad = -1;                        // (the generations before g kept their bits: that is what first_bad said)
    }
}

// Offspring bounds of the ranks (lane r: o_r, r = 0 .. world) and the decision, from the all-gathered totals.  One wave.
struct PlanFixedIn { const uint64_t* all_totals; double u0, n_pop, ess_frac; int rs; uint64_t seed, draw; };      // rs: kFixSystematic / kFixStratified (the outputs' uniforms: seed, draw)
__device__ __forceinline__ double plan_bounds_fixed(const PlanFixedIn& pf, int world, bool& resample)
{
    const int lane = lane_id();
    uint64_t s = 0, q = 0;
    if (lane < world) { s = pf.all_totals[3 * lane]; q = pf.all_totals[3 * lane + 1]; }
    const uint64_t incl = wave_incl_scan_u64(s);
    const uint64_t St = read_lane_u64(incl, kWave - 1), Qt = wave_sum_u64(q);
    const FixedDecision d = fixed_decide(St, Qt, pf.n_pop, pf.ess_frac, true);
    resample = d.resample;
    FixedCdf fc;
    fc.inv = d.inv; fc.u0 = pf.u0; fc.n_pop = pf.n_pop; fc.base = 0;
    fc.seed = pf.seed; fc.draw = pf.draw; fc.uid0 = 0;
    double o = pf.rs == kFixStratified ? fc.first_stratified(incl - s) : fc.g(incl - s);
    if (lane >= world) o = pf.n_pop;
    return o;
}

}  // namespace cph
