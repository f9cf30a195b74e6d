// SMC step for table-weight models on an every-step resampling schedule, built on INTEGER prefix counts.
//
// When every particle enters a step at the same log-weight (t = 0, or the previous step resampled) and the model's
// incremental weight takes one of K = 3 values (HMM: log N(y_t; mean[s], 1), reference include/models/models.hpp:130-139),
// the weight of particle k of generation t-1 is e[x_k], e[s] = exp(ll_s - max ll), and its inclusive CDF value is a function of
// the prefix counts c_s(k) = #{i <= k : x_i = s} alone:
//        C_k = fma(c_2, e_2, fma(c_1, e_1, c_0 * e_0)),   G_k = ceil(fma(C_k, N / W, -u0)),   ancestor of output j = min{k : G_k > j}
// (the CPU restatement the parity tests compare with states the same arithmetic).  No running floating-point sum exists, so
// tiles, wavefronts and shards may evaluate it in any order and still produce the same integers: ancestors are bit-exact against
// that restatement at every population size and over any number of GPUs.
//
// The prefix counts live in a 64-ary hierarchy written by the kernel that produced the generation:
//   level 0     one entry per 1024-particle tile   (plain store by the tile's workgroup)
//   level 1     one entry per 64 tiles              (one 64-bit integer atomic add per workgroup: exact, order-free)
//   level 2     one entry per 4096 tiles            (added to by the LAST workgroup to arrive at each level-1 entry, which the
//                                                    entry's own arrival field tells from the value the add returned)
// so no address ever takes more than 64 atomics per step (a flat "everybody adds to the top" form serialises at ~12 ns per add:
// 117 us per step at 10^7 particles -- measured).  Entries of levels >= 1 sit on cache lines of their own.
// entry = n_0 | n_1 << 28 | arrivals << 56 (n_2 follows from the number of valid particles).  A workgroup that needs the prefix at
// tile c sums, per level, the < 64 entries that precede c's block inside its parent block: one masked load per level per lane and
// ONE wavefront reduction -- instead of every workgroup re-reading every tile partial (the r01 prologue: 3.9 of 12.3 us, 23 MB of
// L2 reads per step).  No LDS table, no normalisation launch between steps, up to 64^3 tiles (2.7e8 particles) per GPU.
// Three copies rotate: a step reads one (generation t-1), adds into the next and clears the third.
//
// Shape of the step kernel (profiles/r02_notes.md has the measurements behind each choice): everything the prologue reads is
// addressed by the launch geometry, so it is fetched at kernel entry in ONE round trip under the Philox draws; the workgroup's
// first wavefront does the search (totals, N / W, probes of the hierarchy) and hands {first source tile, its prefix counts, last
// source tile} to the other three through LDS; all four then walk those source tiles.  The run's last step is a step like any
// other: the read-out (smooth_counts_kernel / counts_filter_final_kernel) takes weights and normaliser from the counts it leaves.
#pragma once
#include "kernels.hpp"
#include "cpprob/detail/fixed_mass.hpp"
#include "strata_cut.hpp"

namespace cph {

// (hierarchy layout, 32-bit wavefront scans, prefix / probe fetches: cpprob/detail/fixed_mass.hpp)
// Wave-uniform doubles the compiler would otherwise keep in scalar registers: the step kernel has more of them than the 102 SGPRs
// a wave owns (58 spilled, each reloaded ~10 times, in the first build).  Laundering a value through an empty asm with a vector
// constraint parks it in a VGPR and its arithmetic stays on the vector unit.
__device__ __forceinline__ double in_vgpr(double x) { asm volatile("" : "+v"(x)); return x; }

struct Cnt2 { uint32_t n0, n1; };
__device__ __forceinline__ uint32_t cnt_n0(uint64_t w) { return (uint32_t)(w & kCntMask); }
__device__ __forceinline__ uint32_t cnt_n1(uint64_t w) { return (uint32_t)((w >> 28) & kCntMask); }

// The tile-level CDF of generation t-1 in the canonical arithmetic.  Counts of the shards that precede this one enter as exact
// doubles (integers below 2^53), so a sharded run evaluates the very expression a single GPU would.
struct TableCdf {
    double e0, e1, e2, inv, u0, n_pop;
    double base0, base1, basev;                // states 0 / 1 and particles before this shard
    uint64_t seed, draw, uid0;                 // stratified resampling (cpprob/detail/fixed_mass.hpp: FixedCdf): Philox key, draw index, id of output 0
    __device__ __forceinline__ double cdf(double c0, double c1, double cv) const      // c*: GLOBAL inclusive counts
    {
        const double c2 = cv - c0 - c1;
        return fma(c2, e2, fma(c1, e1, __dmul_rn(c0, e0)));
    }
    // (never negative: C >= 0 and u0 < 1; may exceed n_pop by rounding only for the population's last sources, whose surplus
    //  outputs nobody consumes -- the clamp of the stated arithmetic changes no ancestor)
    __device__ __forceinline__ double g(double C) const { return ceil(fma(C, inv, -u0)); }
    // Stratified: output j sits at j + u_j (u_j = the 32-bit uniform of OUTPUT j), the sources up to CDF value C reach H = C * (N / W)
    // (one rounded product) and own the outputs with j + u_j < H: A = F + [u_F < H - F], F = floor(H) -- FixedCdf::first_stratified
    // with the table CDF in place of the integer mass; the CPU restatement: orc_resample_table_stratified.
    __device__ __forceinline__ double h(double C) const { return __dmul_rn(C, inv); }
    __device__ __forceinline__ double first_stratified(double C) const
    {
        const double H = h(C), F = floor(H);
        if (F >= n_pop) return n_pop;
        const double u = u01_32(draw_word(seed, uid0 + (uint64_t)F, draw));
        return u < H - F ? F + 1.0 : F;
    }
    template <int RS>
    __device__ __forceinline__ double first(double C) const
    {
        if constexpr (RS == kFixStratified) return first_stratified(C);
        else return g(C);
    }
    // The same for comparisons against the outputs ga and gb only (the search's probe): A is F or F + 1, so "A <= g" is "F <= g"
    // unless F == g -- the Philox block of output F is drawn only where some lane's F hits one of the two (wave-uniform branch).
    template <int RS>
    __device__ __forceinline__ double first_near(double C, double ga, double gb, bool wanted) const
    {
        if constexpr (RS == kFixStratified) {
            const double H = h(C), F = floor(H);
            double r = F >= n_pop ? n_pop : F;
            if (__any(wanted && F < n_pop && (F == ga || F == gb))) {
                const double u = u01_32(draw_word(seed, uid0 + (uint64_t)fmin(F, n_pop - 1.0), draw));
                if (F < n_pop && u < H - F) r = F + 1.0;
            }
            return r;
        } else return g(C);
    }
    // first output owned by the sources that follow `n0, n1` state-0/1 particles among `nv` local particles
    template <int RS = kFixSystematic>
    __device__ __forceinline__ double g_at(uint32_t n0, uint32_t n1, int64_t nv) const
    {
        return first<RS>(cdf(base0 + (double)n0, base1 + (double)n1, basev + (double)nv));
    }
};

__device__ __forceinline__ Cnt2 hier_prefix_sum(int c, const uint64_t (&w)[kHierMaxLevels])
{
    const int lane = lane_id();
    uint32_t s0 = 0, s1 = 0;
#pragma unroll
    for (int l = 0; l < kHierMaxLevels; ++l) {
        const bool in = lane < ((c >> (6 * l)) & 63);
        s0 += in ? cnt_n0(w[l]) : 0u; s1 += in ? cnt_n1(w[l]) : 0u;
    }
    return Cnt2{wave_sum_u32(s0), wave_sum_u32(s1)};
}
__device__ __forceinline__ Cnt2 hier_prefix(const Hier& h, int c)
{
    uint64_t w[kHierMaxLevels];
    hier_prefix_fetch(h, c, w);
    return hier_prefix_sum(c, w);
}

// Totals of the generation: the sum of the (<= 64) top-level entries.
__device__ __forceinline__ uint64_t hier_total_fetch(const Hier& h)
{
    const int lane = lane_id();
    return h.top[(int64_t)(lane < h.top_n ? lane : 0) * h.top_stride];
}
__device__ __forceinline__ Cnt2 hier_total_sum(const Hier& h, uint64_t w)
{
    if (lane_id() >= h.top_n) w = 0;
    return Cnt2{wave_sum_u32(cnt_n0(w)), wave_sum_u32(cnt_n1(w))};
}
__device__ __forceinline__ Cnt2 hier_total(const Hier& h) { return hier_total_sum(h, hier_total_fetch(h)); }

// Largest tile c in [0, nb) whose first owned output G(prefix(c)) is <= g (0 when there is none), with its exclusive prefix
// counts: top-down descent, one load + one scan per level.  Wave-uniform; the last resort of the ancestor search (tile masses so
// uneven that two local probes miss) and the exchange scope's packing, whose outputs sit at the ends of the shard.
template <int RS = kFixSystematic>
__device__ __forceinline__ int hier_locate(const HierTable* __restrict__ ht, int copy, const TableCdf& tc, int64_t n, double g, Cnt2& P)
{
    const int lane = lane_id();
    int blk = 0;
    uint32_t p0 = 0, p1 = 0;
    for (int l = ht->n_lev - 1; l >= 0; --l) {
        const int idx = (blk << 6) + lane;
        uint64_t w = 0;
        const bool in = idx < ht->n_ent[l];
        if (in) w = ht->lvl[copy][l][(int64_t)idx * (l == 0 ? 1 : kHierStride)];
        const uint32_t v0 = cnt_n0(w), v1 = cnt_n1(w);
        const uint32_t i0 = wave_incl_scan_u32(v0), i1 = wave_incl_scan_u32(v1);
        const uint32_t x0 = p0 + i0 - v0, x1 = p1 + i1 - v1;                                   // exclusive prefix at child `lane`
        const int64_t tile0 = (int64_t)idx << (6 * l);                                         // first tile of the child
        const int64_t nv = tile0 * kTile < n ? tile0 * kTile : n;
        const bool ok = in && tc.template g_at<RS>(x0, x1, nv) <= g;
        const unsigned long long m = __ballot(ok);
        const int child = m ? (63 - __builtin_clzll(m)) : 0;                                     // (G is monotone: the set is a prefix)
        p0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, child);
        p1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, child);
        blk = (blk << 6) + child;
    }
    P = Cnt2{p0, p1};
    return blk;
}

template <int RS>
struct CountsLdsT {
    int32_t slot[kTile];        // scatter slots of the output tile
    uint32_t scan[2][kWaves];   // packed per-wave totals of the in-tile scan, double-buffered across source tiles
    int iscr[kWaves];
    uint32_t ustrat[RS == kFixStratified ? kTile : 1];   // stratified: the 32-bit uniforms of the output tile's outputs
    double cd[RS == kFixMultinomial ? kStrataTiles * kTile : 1];   // multinomial: the source tiles' inclusive CDF values, side by side
    uint32_t wtot[RS == kFixMultinomial ? kStrataTiles : 1][kWaves];  // ... and their wavefronts' packed count totals
};
using CountsLds = CountsLdsT<kFixSystematic>;

// Ancestors of the kTile consecutive outputs starting at global output index gj_first (n_out of them), among THIS shard's sources,
// -1 where the ancestor belongs to a shard that precedes this one; outputs at or beyond o_hi = G(all local sources) belong to the
// shards that follow (the caller tests that).  Two parts: the SEARCH for the source tiles involved (counts_locate: the work of one
// wavefront, uniform over the workgroup) and the WALK over them (counts_walk: the whole workgroup).
//
// counts_locate.  `guess` = a tile expected to hold the first ancestor; `first` = the words of the probe at `guess`, fetched by the
// caller ahead of time (nullptr: fetched here).  Returns the first source tile c, its exclusive prefix counts, and the last source
// tile c_last (nb when the probe cannot tell).
struct Located { int c, c_last; uint32_t p0, p1; };
template <int RS = kFixSystematic>
__device__ __forceinline__ Located counts_locate(const Hier& h, const TableCdf& tc, int64_t n, int nb, double gj_first, int n_out, int guess,
                                                 const ProbeWords* first)
{
    const int lane = lane_id();
    const double gj_last = gj_first + (double)(n_out - 1);
    auto nvalid_before = [&](int c) -> int64_t { const int64_t v = (int64_t)c * kTile; return v < n ? v : n; };
    // Probe the tiles around `at`: one hierarchical sum gives the prefix counts of tile cs = max(at - 1, 0), four tile entries those
    // of cs+1 .. cs+4, and lane i evaluates the first output of tile cs + i.  d_out = how far (in outputs) the first output lies
    // from tile cs's: the next probe's aim when this one misses.  The same five values tell the LAST source tile this output tile
    // draws from.
    int c = 0, c_last = nb;
    Cnt2 P{0, 0};
    auto probe = [&](int at, const ProbeWords& pw, double& d_out) -> bool {
        const int cs = at > 0 ? at - 1 : 0;
        const Cnt2 Pc = hier_prefix_sum(cs, pw.lvl);
        const uint64_t we = (lane < 4 && cs + lane < nb) ? pw.we : 0ull;
        const uint32_t v0 = cnt_n0(we), v1 = cnt_n1(we);
        const uint32_t i0 = wave_incl_scan_u32(v0), i1 = wave_incl_scan_u32(v1);
        const uint32_t x0 = Pc.n0 + i0 - v0, x1 = Pc.n1 + i1 - v1;        // lanes 0..4: the prefix at cs + lane (lanes >= 4 hold zeros)
        const bool known = lane < 5 && cs + lane < nb;
        const double gt = tc.template first_near<RS>(tc.cdf(tc.base0 + (double)x0, tc.base1 + (double)x1, tc.basev + (double)nvalid_before(cs + lane)), gj_first, gj_last, known);
        const unsigned long long m = __ballot(known && gt <= gj_first);
        const int i_lo = m ? (63 - __builtin_clzll(m)) : -1;
        d_out = gj_first - read_lane(gt, 0);
        if ((i_lo >= 0 || cs == 0) && i_lo < 4) {
            const int i = i_lo < 0 ? 0 : i_lo;
            c = cs + i;
            P.n0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, i);
            P.n1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, i);
            // (G is monotone in the tile index: the tiles that start at or before the last output form a prefix of the five)
            const unsigned long long mh = __ballot(known && gt <= gj_last);
            const int i_hi = mh ? (63 - __builtin_clzll(mh)) : i;
            c_last = (i_hi >= 4 && cs + 5 < nb) ? nb : cs + (i_hi > i ? i_hi : i);
            return true;
        }
        return false;
    };
    double d;
    bool hit;
    if (first) hit = probe(guess, *first, d);
    else { ProbeWords pw; probe_fetch(h, guess, nb, pw); hit = probe(guess, pw, d); }
    if (!hit) {
        // tile masses are nearly even, so the miss distance in outputs is the miss distance in tiles (x 1024) up to a few tiles:
        // aim again (large populations: the CDF wanders sqrt(N) outputs off the diagonal), then descend from the top
        const double aim = (double)(guess > 0 ? guess - 1 : 0) + floor(d * (1.0 / kTile));
        const int at = (int)fmin(fmax(aim, 0.0), (double)(nb - 1));
        ProbeWords pw;
        probe_fetch(h, at, nb, pw);
        if (!probe(at, pw, d)) { c = hier_locate<RS>(h.table, h.copy, tc, n, gj_first, P); c_last = nb; }
    }
    return Located{c, c_last, P.n0, P.n1};
}

// counts_walk.  raw_m1 / raw_0 / raw_p1 = the states of tiles guess-1, guess, guess+1 fetched at kernel entry (an output tile
// overlaps two of them almost surely, so no load waits for the search).  Slots must hold -1 and be visible (the caller's barrier)
// on entry.
template <class S, bool sharded, int RS = kFixSystematic>
__device__ __forceinline__ void counts_walk(const TableCdf& tc, const S* __restrict__ states, int64_t n, int nb, bool last_shard, double gj_first,
                                            int n_out, const Located& loc, int guess, uint32_t raw_m1, uint32_t raw_0, uint32_t raw_p1,
                                            int32_t (&anc)[kPPT], CountsLdsT<RS>& L)
{
    static_assert(sizeof(S) == 1 && kPPT == 4, "states travel as one byte: 4 per lane = one dword");
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const double gj_last = gj_first + (double)(n_out - 1);
    auto nvalid_before = [&](int c) -> int64_t { const int64_t v = (int64_t)c * kTile; return v < n ? v : n; };
    int c = loc.c, c_last = loc.c_last;
    Cnt2 P{loc.p0, loc.p1};
    // ---- walk the source tiles that own outputs of this tile ----
    auto load_states = [&](int cc) -> uint32_t {
        if (cc == guess) return raw_0;
        if (cc == guess - 1) return raw_m1;
        if (cc == guess + 1) return raw_p1;
        return cc < nb ? *reinterpret_cast<const uint32_t*>(states + (int64_t)cc * kTile + (int64_t)tid * kPPT) : 0u;
    };
    c = __builtin_amdgcn_readfirstlane(c);
    c_last = __builtin_amdgcn_readfirstlane(c_last);
    uint32_t raw = load_states(c);
    int it = 0;
    const double base2 = tc.basev - tc.base0 - tc.base1;                 // (uniform; zero on a single shard)
    // One source tile: in-tile scan of the state counts, G at the lane's five particle boundaries, one slot per source that owns an
    // output of this tile.  EDGE = the shard's last tile (the only one that may be partly valid or hold the population's last
    // source): every other tile runs the form without those tests.
    auto tile = [&](auto edge_tag, uint32_t raw_c) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const int vb = tid * kPPT;                                       // particles of this tile before the lane's first
        const int nvt = EDGE ? (int)(n - (int64_t)c * kTile) : kTile;    // valid particles of this tile (>= 1)
        // per-lane inclusive counts of states 0 / 1, packed 16 + 16 bits
        uint32_t q[kPPT];
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const uint32_t s = (raw_c >> (8 * k)) & 0xffu;
            const bool valid = !EDGE || vb + k < nvt;
            run += (valid && s == 0) ? 1u : 0u;
            run += (valid && s == 1) ? 0x10000u : 0u;
            q[k] = run;
        }
        const uint32_t incl = wave_incl_scan_u32(run);
        if (lane == kWave - 1) L.scan[it & 1][wv] = incl;
        __syncthreads();
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            const uint32_t s = L.scan[it & 1][w];
            if (w < wv) off += s;
            tot += s;
        }
        const uint32_t excl = off + incl - run;                          // packed exclusive prefix of this lane
        // counts enter the CDF as exact integers: summed as integers, converted once (any association of exact integers below
        // 2^53 is the same double, so this IS the stated arithmetic).  What leaves is the output's place in THIS tile, clamped to
        // [0, kTile]: a source owns outputs here iff its clamped end exceeds its clamped start, which is then its first slot.
        const uint32_t nvb = (uint32_t)nvalid_before(c);                 // local particles before this tile (< 2^31)
        // exact: integers, |g - gj_first| <= the population (< 2^31): converted, then clamped as an integer -- one 32-bit median instead
        // of a maximum and a minimum at the fp64 rate, five times a tile
        auto place = [&](double g) -> int { const int x = (int)(g - gj_first); return x < 0 ? 0 : (x > kTile ? kTile : x); };
        auto gk = [&](uint32_t packed, int upto) -> int {                // after `upto` particles of the tile, `packed` of them in states 0 / 1
            const uint32_t n0 = P.n0 + (packed & 0xffffu), n1 = P.n1 + (packed >> 16);
            const uint32_t n2 = nvb + (uint32_t)(EDGE && upto > nvt ? nvt : upto) - n0 - n1;
            double c0 = (double)n0, c1 = (double)n1, c2 = (double)n2;
            if (sharded) { c0 += tc.base0; c1 += tc.base1; c2 += base2; }
            const double C = fma(c2, tc.e2, fma(c1, tc.e1, __dmul_rn(c0, tc.e0)));
            if constexpr (RS == kFixStratified) {
                // (L.ustrat: the outputs' uniforms, staged by the caller)
                const double H = tc.h(C), F = floor(H), d = F - gj_first;
                if (!(d >= 0.0)) return 0;
                if (d >= (double)kTile) return kTile;
                const int i = (int)d;
                return i + (u01_32(L.ustrat[i]) < H - F ? 1 : 0);
            } else return place(tc.g(C));
        };
        const int src0 = c * kTile + vb;
        const int p_all = EDGE && last_shard ? place(tc.n_pop) : 0;
        int p_prev = gk(excl, vb);
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            int p = gk(excl + q[k], vb + k + 1);
            if (EDGE && last_shard && vb + k + 1 == nvt) p = p_all;      // the population's last source owns the rest
            if (p > p_prev) { L.slot[p_prev] = src0 + k; p_prev = p; }
        }
        P.n0 += tot & 0xffffu; P.n1 += tot >> 16;
    };
    while (c < nb && c <= c_last) {
        // (wave-uniform values -- the branches are made scalar so that the barrier inside the loop sits in uniform control flow)
        if (c_last >= nb &&                                              // the last tile is not known from the probe: test where this one starts
            __builtin_amdgcn_readfirstlane(tc.template g_at<RS>(P.n0, P.n1, nvalid_before(c)) > gj_last ? 1 : 0)) break;
        const uint32_t raw_next = c < c_last ? load_states(c + 1) : 0u;  // (beyond the prefetched three: travels while this tile is processed)
        if (c == nb - 1) tile(std::true_type{}, raw); else tile(std::false_type{}, raw);
        ++it;
        raw = raw_next;
        ++c;
    }
    __syncthreads();
    // inclusive prefix-max over the slots
    int32_t v[kPPT];
    load4(L.slot, (int64_t)tid * kPPT, v);
    lane_prefix_max(v);
    int32_t incl = wave_incl_max_i32(v[kPPT - 1]);
    if (lane == kWave - 1) L.iscr[wv] = incl;
    int32_t excl = dpp_or_i32<0x138 /* wave_shr:1 */>(incl, -1);
    if (lane == 0) excl = -1;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < kWaves; ++w)
        if (w < wv) excl = max(excl, L.iscr[w]);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) anc[k] = max(v[k], excl);
}

// Both parts in every wavefront (the exchange scope's packing, whose output tiles sit anywhere in the shard).
template <class S, bool sharded, int RS = kFixSystematic>
__device__ __forceinline__ void ancestors_counts(const Hier& h, const TableCdf& tc, const S* __restrict__ states, int64_t n, int nb,
                                                 bool last_shard, double gj_first, int n_out, int guess, int32_t (&anc)[kPPT], CountsLdsT<RS>& L)
{
    if constexpr (RS == kFixStratified) {                        // (the outputs' uniforms, behind a barrier of their own)
        uint32_t w[kPPT];
        draw_words4(tc.seed, tc.uid0 + (uint64_t)gj_first + (uint64_t)threadIdx.x * kPPT, tc.draw, w);
        store4(L.ustrat, (int64_t)threadIdx.x * kPPT, w);
        __syncthreads();
    }
    const Located loc = counts_locate<RS>(h, tc, n, nb, gj_first, n_out, guess, nullptr);
    counts_walk<S, sharded, RS>(tc, states, n, nb, last_shard, gj_first, n_out, loc, /* no tile was prefetched */ -16, 0u, 0u, 0u, anc, L);
}

// ---- multinomial resampling, strata form, on the table CDF (csrc/step_fixed.hpp states the form; the CPU restatement:
//      orc_resample_table_multinomial) ----
// The strata's bounds are B_w = w (W 2^-k): exact scaling, one rounded product, B_K = W.  Output s of stratum w takes
// tau_s = fma(v_s, B_w+1 - B_w, B_w), v_s the 53-bit uniform of output s; ancestor = min{k : C_k > tau_s}, C_k the CDF of the
// inclusive prefix counts (TableCdf::cdf: the systematic comb's own values).  One population per context.
// The SEARCH (one wavefront): the strata of the outputs (strata_window) and the source tiles that hold their CDF range: the largest
// tile whose starting CDF value is <= x, by the systematic search's probe of five tiles, a second probe aimed by the value, or the descent.
// One SHARD of a population (exchange scope): its sources hold the CDF range [c_lo, c_hi) of W (the population's last shard also
// the thresholds that round up to W); only thresholds inside it are searched here.  One population: c_lo = 0, c_hi = W, last.
struct LocatedStrata { Located loc; int w0, w1; };
__device__ __forceinline__ LocatedStrata counts_strata_locate(const Hier& h, const TableCdf& tc, const uint32_t* __restrict__ offs, int k, int64_t n, int nb, int w_near, int guess,
                                                              uint32_t s_first, uint32_t s_last, double W, double c_lo, double c_hi, bool last_shard, const ProbeWords* first)
{
    const int lane = lane_id();
    LocatedStrata r;
    strata_window(offs, k, w_near, s_first, s_last, r.w0, r.w1);
    const double unit = ldexp(W, -k);
    const double x_lo = fmax((double)r.w0 * unit, c_lo), x_hi = last_shard ? (double)(r.w1 + 1) * unit : fmin((double)(r.w1 + 1) * unit, c_hi);       // (thresholds lie in [x_lo, x_hi]: the fma may round up to the bound)
    if (x_hi < x_lo) { r.loc = Located{1, 0, 0u, 0u}; return r; }      // no threshold of these outputs lies in this shard's range
    auto nvalid_before = [&](int c) -> int64_t { const int64_t v = (int64_t)c * kTile; return v < n ? v : n; };
    int c = 0, c_last = nb;
    Cnt2 P{0, 0};
    auto probe = [&](int at, const ProbeWords& pw) -> bool {
        const int cs = at > 0 ? at - 1 : 0;
        const Cnt2 Pc = hier_prefix_sum(cs, pw.lvl);
        const uint64_t we = (lane < 4 && cs + lane < nb) ? pw.we : 0ull;
        const uint32_t v0 = cnt_n0(we), v1 = cnt_n1(we);
        const uint32_t i0 = wave_incl_scan_u32(v0), i1 = wave_incl_scan_u32(v1);
        const uint32_t x0 = Pc.n0 + i0 - v0, x1 = Pc.n1 + i1 - v1;        // lanes 0..4: the prefix counts at tile cs + lane
        const double cv = tc.cdf(tc.base0 + (double)x0, tc.base1 + (double)x1, tc.basev + (double)nvalid_before(cs + lane));
        const bool known = lane < 5 && cs + lane < nb;
        const unsigned long long m = __ballot(known && cv <= x_lo);
        const int i_lo = m ? (63 - __builtin_clzll(m)) : -1;
        if ((i_lo >= 0 || cs == 0) && i_lo < 4) {
            const int i = i_lo < 0 ? 0 : i_lo;
            c = cs + i;
            P.n0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, i);
            P.n1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, i);
            const unsigned long long mh = __ballot(known && cv <= x_hi);
            const int i_hi = mh ? (63 - __builtin_clzll(mh)) : i;
            c_last = (i_hi >= 4 && cs + 5 < nb) ? nb : cs + (i_hi > i ? i_hi : i);
            return true;
        }
        return false;
    };
    if (!(first && probe(guess, *first))) {
        const double aim = (x_lo - c_lo) * ((double)nb / (c_hi - c_lo > 0.0 ? c_hi - c_lo : 1.0));
        const int at = (int)fmin(fmax(aim, 0.0), (double)(nb - 1));
        ProbeWords pw;
        probe_fetch(h, at, nb, pw);
        if (!probe(at, pw)) {
            // top-down descent: the largest tile whose starting CDF value is <= x_lo
            const HierTable* __restrict__ ht = h.table;
            int blk = 0;
            uint32_t p0 = 0, p1 = 0;
            for (int l = ht->n_lev - 1; l >= 0; --l) {
                const int idx = (blk << 6) + lane;
                uint64_t w = 0;
                const bool in = idx < ht->n_ent[l];
                if (in) w = ht->lvl[h.copy][l][(int64_t)idx * (l == 0 ? 1 : kHierStride)];
                const uint32_t v0 = cnt_n0(w), v1 = cnt_n1(w);
                const uint32_t i0 = wave_incl_scan_u32(v0), i1 = wave_incl_scan_u32(v1);
                const uint32_t x0 = p0 + i0 - v0, x1 = p1 + i1 - v1;
                const int64_t tile0 = (int64_t)idx << (6 * l);
                const int64_t nv = tile0 * kTile < n ? tile0 * kTile : n;
                const bool ok = in && tc.cdf(tc.base0 + (double)x0, tc.base1 + (double)x1, tc.basev + (double)nv) <= x_lo;
                const unsigned long long m = __ballot(ok);
                const int child = m ? (63 - __builtin_clzll(m)) : 0;
                p0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, child);
                p1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, child);
                blk = (blk << 6) + child;
            }
            c = blk; P = Cnt2{p0, p1}; c_last = nb;
        }
    }
    r.loc = Located{c, c_last, P.n0, P.n1};
    return r;
}

// The WALK (the whole workgroup): the lane's four outputs take their thresholds; the source tiles, up to kStrataTiles at a time, put
// their particles' CDF values side by side in LDS (one count scan a tile, two barriers a group), and every output whose threshold
// lies in the group searches them once.  uid = the id of the lane's first output (a multiple of four); j0 = its index.
template <class S>
__device__ __forceinline__ void counts_strata_walk(const TableCdf& tc, const uint32_t* __restrict__ offs, int k, const S* __restrict__ states, int64_t n, int nb,
                                                   const LocatedStrata& sl, double W, int64_t j0, uint64_t seed, uint64_t draw, uint64_t uid,
                                                   int32_t (&anc)[kPPT], CountsLdsT<kFixMultinomial>& L, double c_lo, double c_hi, bool last_shard, bool (&mine)[kPPT],
                                                   bool parked = false)
{
    static_assert(sizeof(S) == 1 && kPPT == 4, "states travel as one byte: 4 per lane = one dword");
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    uint64_t vb[kPPT];
    // (parked: the caller's wavefronts drew the outputs' bits while they waited for the search and left them where L.cd begins -- read here,
    //  before the first barrier of the walk, behind which the CDF values overwrite them)
    if (parked) {
#pragma unroll
        for (int i = 0; i < kPPT; ++i) vb[i] = reinterpret_cast<const uint64_t*>(L.cd)[i * kThreads + tid];
    } else strata_bits4(seed, draw, uid, vb);
    const double v[kPPT] = {(double)vb[0] * kTwoPowM53, (double)vb[1] * kTwoPowM53, (double)vb[2] * kTwoPowM53, (double)vb[3] * kTwoPowM53};
    double tau[kPPT];
    bool live[kPPT];
#pragma unroll
    for (int i = 0; i < kPPT; ++i) { live[i] = false; tau[i] = 0.0; }
    const int w0 = __builtin_amdgcn_readfirstlane(sl.w0), w1 = __builtin_amdgcn_readfirstlane(sl.w1);
    const double unit = ldexp(W, -k);
    {
        int ws[kPPT];
        lane_strata4(offs, k, w0, w1, (uint32_t)j0, ws, live);
#pragma unroll
        for (int i = 0; i < kPPT; ++i) {
            const double b_lo = __dmul_rn((double)ws[i], unit), b_hi = __dmul_rn((double)(ws[i] + 1), unit);
            tau[i] = live[i] ? fma(v[i], __dsub_rn(b_hi, b_lo), b_lo) : 0.0;
        }
    }
    // (j0 = the lane's first output in the POPULATION; of a shard's outputs only those whose threshold lies in its sources' range)
#pragma unroll
    for (int i = 0; i < kPPT; ++i) { live[i] = live[i] && tau[i] >= c_lo && (last_shard || tau[i] < c_hi); mine[i] = live[i]; }
    CPH_STAMP(8);
    const double x_hi = (double)(w1 + 1) * unit;
    auto nvalid_before = [&](int c) -> int64_t { const int64_t v2 = (int64_t)c * kTile; return v2 < n ? v2 : n; };
    int c = __builtin_amdgcn_readfirstlane(sl.loc.c);
    const int c_last = __builtin_amdgcn_readfirstlane(sl.loc.c_last);
    uint32_t P0 = sl.loc.p0, P1 = sl.loc.p1;                              // state-0 / state-1 particles before tile c
    const double base2 = tc.basev - tc.base0 - tc.base1;
    bool first_group = true;
    while (c < nb && c <= c_last) {
        // (the last tile may be unknown: stop where a tile starts beyond the largest threshold)
        if (c_last >= nb && __builtin_amdgcn_readfirstlane(tc.cdf(tc.base0 + (double)P0, tc.base1 + (double)P1, tc.basev + (double)nvalid_before(c)) > x_hi ? 1 : 0)) break;
        int nt = (c_last < nb ? c_last : nb - 1) - c + 1;
        if (nt > kStrataTiles) nt = kStrataTiles;
        if (!first_group) __syncthreads();                              // (the previous group's searches have read the LDS arrays)
        first_group = false;
        uint32_t raw[kStrataTiles], qk[kStrataTiles][kPPT], runs[kStrataTiles], incls[kStrataTiles];
#pragma unroll
        for (int j = 0; j < kStrataTiles; ++j)
            raw[j] = j < nt ? *reinterpret_cast<const uint32_t*>(states + (int64_t)(c + j) * kTile + (int64_t)tid * kPPT) : 0u;
#pragma unroll
        for (int j = 0; j < kStrataTiles; ++j) {
            const int64_t nvt = n - (int64_t)(c + j) * kTile;             // valid particles of the tile (the shard's last one may be ragged)
            uint32_t run = 0;
#pragma unroll
            for (int i = 0; i < kPPT; ++i) {
                const uint32_t st = (raw[j] >> (8 * i)) & 0xffu;
                const bool valid = j < nt && (int64_t)(tid * kPPT + i) < nvt;
                run += (valid && st == 0) ? 1u : 0u;
                run += (valid && st == 1) ? 0x10000u : 0u;
                qk[j][i] = run;
            }
            runs[j] = run;
            incls[j] = wave_incl_scan_u32(run);
            if (lane == kWave - 1) L.wtot[j][wv] = incls[j];
        }
        __syncthreads();
        CPH_STAMP(9);
        double t_end = 0.0;                                               // the CDF value at the end of the group
#pragma unroll
        for (int j = 0; j < kStrataTiles; ++j) {
            if (j < nt) {
                uint32_t off = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) { const uint32_t x = L.wtot[j][w]; if (w < wv) off += x; tot += x; }
                const uint32_t excl = off + incls[j] - runs[j];
                const int64_t nvt = n - (int64_t)(c + j) * kTile;
                const uint32_t nvb = (uint32_t)nvalid_before(c + j);
                double cdv[kPPT];
#pragma unroll
                for (int i = 0; i < kPPT; ++i) {
                    const uint32_t packed = excl + qk[j][i];
                    const int upto = tid * kPPT + i + 1;
                    const uint32_t n0 = P0 + (packed & 0xffffu), n1 = P1 + (packed >> 16);
                    const uint32_t n2 = nvb + (uint32_t)((int64_t)upto > nvt ? (nvt > 0 ? nvt : 0) : upto) - n0 - n1;
                    cdv[i] = fma((double)n2 + base2, tc.e2, fma((double)n1 + tc.base1, tc.e1, __dmul_rn((double)n0 + tc.base0, tc.e0)));
                }
                using D2 = double __attribute__((ext_vector_type(2)));
                D2 a0, a1;
                a0[0] = cdv[0]; a0[1] = cdv[1]; a1[0] = cdv[2]; a1[1] = cdv[3];
                *reinterpret_cast<D2*>(L.cd + (size_t)j * kTile + (size_t)tid * kPPT) = a0;
                *reinterpret_cast<D2*>(L.cd + (size_t)j * kTile + (size_t)tid * kPPT + 2) = a1;
                P0 += tot & 0xffffu; P1 += tot >> 16;
                t_end = tc.cdf(tc.base0 + (double)P0, tc.base1 + (double)P1, tc.basev + (double)nvalid_before(c + j + 1));
            }
        }
        __syncthreads();
        CPH_STAMP(10);
        const int len = nt * kTile;
#pragma unroll
        for (int i = 0; i < kPPT; ++i) {
            if (live[i] && tau[i] < t_end) {
                int a = 0, b = len - 1;                                    // the first idx with cd[idx] > tau (cd[len - 1] = t_end > tau)
                for (int hstep = 0; hstep < 12; ++hstep) { const int mid = (a + b) >> 1; if (L.cd[mid] > tau[i]) b = mid; else a = mid + 1; }
                static_assert(kStrataTiles * kTile <= 4096, "twelve halvings");
                anc[i] = c * kTile + (a < len ? a : len - 1);
                live[i] = false;
            }
        }
        c += nt;
        CPH_STAMP(11);
    }
    // a threshold that rounded up to the population's whole mass: its last particle
#pragma unroll
    for (int i = 0; i < kPPT; ++i) if (live[i]) anc[i] = (int32_t)(n - 1);
}

struct StepFound { Located loc; double inv, base0, base1, basev; int64_t l0, l1; int w0, w1; double W, c_lo, c_hi; };      // what the searching wavefront hands the other three

#ifndef CPPROB_HAND_OVER
#define CPPROB_HAND_OVER 1
#endif
template <class Model>
struct StepCountsArgs {
    ModelParams mp; int t, T; int64_t n, ld, rs;
    uint64_t seed, pid0;
    typename Model::store_t* values; int32_t* anc;
    Hier h;                                                     // generation t-1's counts (read); generation t's are written one copy further
    // host-evaluated per-step constants (kernel arguments: no memory round trip in front of the prologue)
    double e_prev[4];                                           // exp(ll_s - max ll) of step t-1, s = 0..2, then max ll
    double u0;                                                  // systematic offset of the resampling before step t (Philox, evaluated on the host)
    StepCtrl* ctrl; double n_pop; double* ess_trace; int32_t* resampled;
    // one shard of a joint population (exchange scope): the all-gathered {n_0, n_1, particles} of every rank's generation t-1,
    // exact doubles; nullptr on a single shard
    const double* all_totals; int world, rank;
    const int64_t* annex_base;                                  // [T + 1]: annex columns in use before the immigrants of step t arrive
    const int64_t* src_shift;                                   // exchange scope: tiles by which this shard's outputs sit off its sources (the previous exchange's plan)
    int row_w, row_r;                                           // rows of values[] this step writes / reads (t, t - 1; a filtering-only run: its two rows in turn)
    double* filter_stats;                                       // filtering-only run: [T][3], generation t-1's P(x = s) from its totals (nullptr otherwise)
    const uint32_t* trace_prev; uint32_t* trace_next;           // trace words (trace_words.hpp) of generations t-1 / t, or nullptr: short discrete traces (shards: [rs], annex included)
    const uint32_t* strata_offs; int strata_k;                  // multinomial, strata form: first output of every stratum at this step (step_fixed.hpp: multinomial_strata_kernel)
    CutView cut;                                                // ... of one shard of a joint population: what the ranks' boundaries cut (strata_cut.hpp)
};

// This tile's entry of generation t's hierarchy, added into the levels above (see the header of this file), and the entries of
// the third copy this tile is responsible for clearing.  One thread.
__device__ __forceinline__ void hier_publish(const Hier& h, int bid, int nb, uint32_t n0, uint32_t n1, bool publish)
{
    uint64_t* l0 = const_cast<uint64_t*>(h.lvl[0]);
    uint64_t* l1 = const_cast<uint64_t*>(h.lvl[1]);
    uint64_t* l2 = const_cast<uint64_t*>(h.lvl[2]);
    const uint64_t ent = (uint64_t)n0 | ((uint64_t)n1 << 28);
    const int b1 = bid >> 6, b2 = bid >> 12;
    if (publish) {
        l0[h.to_next + bid] = ent;
        if (h.n_lev == 2) {
            atomicAdd(reinterpret_cast<unsigned long long*>(l1 + h.to_next + (int64_t)b1 * kHierStride), (unsigned long long)ent);
        } else if (h.n_lev == 3) {
            // the add returns what the entry held: the arrival field tells the last tile of the block, which forwards the block's total
            const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(l1 + h.to_next + (int64_t)b1 * kHierStride),
                                                     (unsigned long long)(ent + (1ull << 56)));
            const int tiles_in_block = nb - (b1 << 6) < 64 ? nb - (b1 << 6) : 64;
            if ((int)(old >> 56) == tiles_in_block - 1) {
                const uint64_t tot = (old + ent) & ((1ull << 56) - 1);
                atomicAdd(reinterpret_cast<unsigned long long*>(l2 + h.to_next + (int64_t)b2 * kHierStride), (unsigned long long)tot);
            }
        }
    }
    if (h.n_lev >= 2 && (b1 << 6) == bid) l1[h.to_clear + (int64_t)b1 * kHierStride] = 0;
    if (h.n_lev >= 3 && (b2 << 12) == bid) l2[h.to_clear + (int64_t)b2 * kHierStride] = 0;
}

// SHARDED: one shard of a joint population (exchange scope).  Compile-time forms: each keeps only the arguments it uses in scalar
// registers.  The run's last step is a step like any other: the read-out works from the counts it leaves (smooth_counts_kernel).
#ifdef CPPROB_COUNTS_WAVES
#define CPPROB_COUNTS_OCC __attribute__((amdgpu_waves_per_eu(CPPROB_COUNTS_WAVES)))
#else
#define CPPROB_COUNTS_OCC
#endif
template <class Model, bool SHARDED, int RS = kFixSystematic>
__global__ __launch_bounds__(kThreads) CPPROB_COUNTS_OCC void smc_step_counts_kernel(StepCountsArgs<Model> a)
{
    using V = typename Model::value_t;
    using S = typename Model::store_t;
    static_assert(Model::kWeightTable == 3, "prefix-count form: three table values (two stored counts)");
    static_assert(RS == kFixSystematic || RS == kFixStratified || RS == kFixMultinomial, "prefix-count form: systematic, stratified or (strata-form) multinomial resampling");
    __shared__ CountsLdsT<RS> L;
    __shared__ int s_cnt[kWaves * 4];
    __shared__ __attribute__((aligned(16))) uint64_t s_model[Model::kStagedWords];
    __shared__ __attribute__((aligned(16))) StepFound s_found;
    constexpr bool kHandOver = CPPROB_HAND_OVER;
    __shared__ __attribute__((aligned(16))) typename Model::Rand s_rnd[kHandOver ? kWave : 1];
    const int tid = threadIdx.x;
    const int nb = (int)gridDim.x;
    const int bid = xcd_contiguous_tile((int)blockIdx.x, nb);
    const int64_t j0 = (int64_t)bid * kTile + (int64_t)tid * kPPT;
    const int t = a.t;

    // Everything the prologue reads from memory is addressed by the launch geometry alone -- the states of the source tiles this
    // output tile almost surely descends from (its own index and both neighbours) and, in the workgroup's first wavefront (which
    // searches for all four), the generation's totals and the words of the search's first probe: fetched here, in one round trip
    // that the random draws below cover.
    CPH_STAMP(0);
    const S* prev_row = a.values + (int64_t)a.row_r * a.rs;
    const bool searcher = wave_id() == 0;
    uint32_t raw_0 = 0, raw_m1 = 0, raw_p1 = 0;
    uint64_t w_tot = 0;
    ProbeWords pw0{};
    double r0 = 0.0, r1 = 0.0, rv = 0.0;
    // the source tile this output tile is expected to start in: its own index, moved by the shard's offset in a sharded run
    int guess = bid;
    if (SHARDED && t > 0 && a.src_shift) { const int64_t g2 = (int64_t)bid + *a.src_shift; guess = (int)(g2 < 0 ? 0 : (g2 >= nb ? nb - 1 : g2)); }
    const bool traced = a.trace_next != nullptr;                                              // (kernel-uniform; trace_words.hpp)
    if (t > 0) {
        const int64_t g0 = (int64_t)guess * kTile + (int64_t)tid * kPPT;
        raw_0 = *reinterpret_cast<const uint32_t*>(prev_row + g0);
        raw_m1 = *reinterpret_cast<const uint32_t*>(prev_row + (guess > 0 ? g0 - kTile : g0));
        raw_p1 = *reinterpret_cast<const uint32_t*>(prev_row + (guess + 1 < nb ? g0 + kTile : g0));
        if (searcher) {
            w_tot = hier_total_fetch(a.h);
            probe_fetch(a.h, guess, nb, pw0);
            if (SHARDED) {
                const int r = tid < a.world ? tid : 0;
                r0 = a.all_totals[3 * r]; r1 = a.all_totals[3 * r + 1]; rv = a.all_totals[3 * r + 2];
            }
        }
    }
    // The variates.  Where a search follows (t > 0) the searching wavefront does not draw its own: the search is the workgroup's serial
    // chain and starts as soon as its loads are back; wavefront 2 draws that share too, under the search, and hands it over through LDS
    // behind the search's barrier.
    typename Model::Rand rnd[kPPT / 4];
    static_assert(kPPT == 4, "one draw4 a lane");
    const bool hand_over = kHandOver && t > 0;
    if (!(hand_over && searcher)) {
#pragma unroll
        for (int q = 0; q < kPPT / 4; ++q) Model::draw4(a.seed, a.pid0 + (uint64_t)j0 + 4 * q, t, rnd[q]);
    }
    if (hand_over && wave_id() == 2) {
        typename Model::Rand r0;
        Model::draw4(a.seed, a.pid0 + (uint64_t)((int64_t)bid * kTile + (int64_t)lane_id() * kPPT), t, r0);
        s_rnd[lane_id()] = r0;
    }
    CPH_STAMP(1);

    int32_t anc[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) anc[k] = (int32_t)(j0 + k);
    if (t > 0) {
        if (guess == 0) raw_m1 = 0u;
        if (guess + 1 >= nb) raw_p1 = 0u;
        if constexpr (RS != kFixMultinomial) {
            int32_t neg[kPPT];
            lane_fill(neg, (int32_t)-1);
            store4(L.slot, (int64_t)tid * kPPT, neg);
        }
        const int64_t rem = a.n - (int64_t)bid * kTile;
        const int n_out = rem < kTile ? (int)rem : kTile;
        // (a shard of a joint population draws the population's outputs; a population of its own -- islands included, whose pid0
        //  only selects RNG streams -- draws its own)
        const double gj_first = SHARDED ? (double)(a.pid0 + (uint64_t)bid * kTile) : (double)((uint64_t)bid * kTile);
        const bool last_shard = SHARDED ? a.rank + 1 == a.world : true;
        TableCdf tc;
        tc.e0 = in_vgpr(a.e_prev[0]); tc.e1 = in_vgpr(a.e_prev[1]); tc.e2 = in_vgpr(a.e_prev[2]); tc.n_pop = a.n_pop;
        tc.u0 = in_vgpr(a.u0);
        tc.base0 = 0.0; tc.base1 = 0.0; tc.basev = 0.0;
        tc.seed = a.seed; tc.draw = kResampleDrawBase + (uint64_t)t; tc.uid0 = SHARDED ? 0 : a.pid0;
        if constexpr (RS == kFixStratified) {
            // (the outputs' uniforms, staged by the three wavefronts that do not search: the second one also draws the searching
            //  wavefront's share, so no Philox block sits on the search's chain)
            if (!searcher) {
                uint32_t w[kPPT];
                draw_words4(tc.seed, tc.uid0 + (uint64_t)gj_first + (uint64_t)tid * kPPT, tc.draw, w);
                store4(L.ustrat, (int64_t)tid * kPPT, w);
                if (wave_id() == 1) {
                    draw_words4(tc.seed, tc.uid0 + (uint64_t)gj_first + (uint64_t)(tid - kWave) * kPPT, tc.draw, w);
                    store4(L.ustrat, (int64_t)(tid - kWave) * kPPT, w);
                }
            }
        }
        if constexpr (RS == kFixMultinomial) {
            // (the 53-bit uniforms of this tile's outputs inside their strata: two Philox blocks a lane that depend on nothing but ids --
            //  drawn by the wavefronts that wait for the search, wavefront 1 also the searching one's, and parked where L.cd begins)
            if (!searcher) {
                uint64_t vb[kPPT];
                uint64_t* park = reinterpret_cast<uint64_t*>(L.cd);
                strata_bits4(a.seed, kResampleDrawBase2 + (uint64_t)t, a.pid0 + (uint64_t)j0, vb);
#pragma unroll
                for (int i = 0; i < kPPT; ++i) park[i * kThreads + tid] = vb[i];
                if (wave_id() == 1) {
                    strata_bits4(a.seed, kResampleDrawBase2 + (uint64_t)t, a.pid0 + (uint64_t)((int64_t)bid * kTile + (int64_t)lane_id() * kPPT), vb);
#pragma unroll
                    for (int i = 0; i < kPPT; ++i) park[i * kThreads + lane_id()] = vb[i];
                }
            }
        }
        if (searcher) {
            // ---- one wavefront: the generation's totals, this shard's place in the joint population, the source tiles this output
            //      tile draws from; the other three pick the results up behind the barrier ----
            const int lane = tid;
            double tot0, tot1;
            if (SHARDED) {
                if (lane >= a.world) { r0 = 0.0; r1 = 0.0; rv = 0.0; }
                const bool before = lane < a.rank;
                tc.base0 = wave_sum(before ? r0 : 0.0); tc.base1 = wave_sum(before ? r1 : 0.0); tc.basev = wave_sum(before ? rv : 0.0);
                tot0 = wave_sum(r0); tot1 = wave_sum(r1);              // (sums of integers below 2^53: exact in any order)
            } else {
                const Cnt2 tl = hier_total_sum(a.h, w_tot);
                tot0 = (double)tl.n0; tot1 = (double)tl.n1;
            }
            const double W = tc.cdf(tot0, tot1, tc.n_pop);
            tc.inv = in_vgpr(a.n_pop / W);
            if (bid == 0 && tid == 0) {                            // bookkeeping of step t-1 for the host: ESS (thesis p.37), evidence
                const double mref = a.e_prev[3];
                const double tot2 = a.n_pop - tot0 - tot1;
                const double Q = fma(tot2, __dmul_rn(tc.e2, tc.e2), fma(tot1, __dmul_rn(tc.e1, tc.e1), __dmul_rn(tot0, __dmul_rn(tc.e0, tc.e0))));
                const double ess = W * W / Q;
                StepCtrl* c = a.ctrl;
                c->M = mref; c->W = W; c->Q = Q; c->ess = ess; c->do_resample = 1;
                c->cdf_lo = 0.0; c->w_local = W; c->scale = 1.0; c->u0 = tc.u0; c->inv_stepw = tc.inv; c->lw_after = 0.0; c->inv_global = tc.inv;
                double lz = (t == 1) ? 0.0 : c->log_z;
                int nr = (t == 1) ? 0 : c->n_resampled;
                lz += mref + log(W / a.n_pop); nr += 1;
                c->log_z = lz; c->n_resampled = nr;
                if (a.ess_trace) a.ess_trace[t - 1] = ess;
                if (a.resampled) a.resampled[t - 1] = 1;
                if (a.filter_stats) {                              // predict hit t-1 under generation t-1's own weights
                    double* fs = a.filter_stats + 3 * (t - 1);
                    fs[0] = __dmul_rn(tot0, tc.e0) / W; fs[1] = __dmul_rn(tot1, tc.e1) / W; fs[2] = __dmul_rn(tot2, tc.e2) / W;
                }
            }
            Located loc{0, 0, 0, 0};
            int sw0 = 0, sw1 = 0;
            double c_lo = 0.0, c_hi = W;
            if constexpr (RS == kFixMultinomial) {
                // (a shard: strata, outputs and thresholds are the POPULATION's; only those inside this shard's CDF range are searched here)
                const uint64_t gfirst = SHARDED ? a.pid0 + (uint64_t)bid * kTile : (uint64_t)bid * kTile;
                const int64_t nb_pop = SHARDED ? ((int64_t)a.n_pop + kTile - 1) / kTile : (int64_t)nb;
                if (SHARDED) {
                    const Cnt2 tl = hier_total_sum(a.h, w_tot);
                    c_lo = tc.cdf(tc.base0, tc.base1, tc.basev);
                    c_hi = tc.cdf(tc.base0 + (double)tl.n0, tc.base1 + (double)tl.n1, tc.basev + (double)a.n);
                }
                const LocatedStrata ls = counts_strata_locate(a.h, tc, a.strata_offs, a.strata_k, a.n, nb, strata_near(a.strata_k, nb_pop, (int64_t)(gfirst / kTile)), guess, (uint32_t)gfirst,
                                                              (uint32_t)(gfirst + (uint64_t)n_out - 1), W, c_lo, c_hi, last_shard, &pw0);
                loc = ls.loc; sw0 = ls.w0; sw1 = ls.w1;
            } else loc = counts_locate<RS>(a.h, tc, a.n, nb, gj_first, n_out, guess, &pw0);
            int64_t l0 = 0, l1 = 0;
            if (SHARDED && RS != kFixMultinomial) {
                // outputs below o_lo / at or beyond o_hi descend from other shards' sources
                const Cnt2 tl = hier_total_sum(a.h, w_tot);
                const double o_lo = tc.template g_at<RS>(0, 0, 0), o_hi = last_shard ? a.n_pop : tc.template g_at<RS>(tl.n0, tl.n1, a.n);
                const double sb = (double)a.pid0;
                l0 = (int64_t)fmin(fmax(o_lo - sb, 0.0), (double)a.n); l1 = (int64_t)fmin(fmax(o_hi - sb, 0.0), (double)a.n);
            }
            if (tid == 0) {
                s_found.loc = loc; s_found.inv = tc.inv; s_found.base0 = tc.base0; s_found.base1 = tc.base1; s_found.basev = tc.basev;
                s_found.l0 = l0; s_found.l1 = l1; s_found.w0 = sw0; s_found.w1 = sw1; s_found.W = W; s_found.c_lo = c_lo; s_found.c_hi = c_hi;
                Model::stage(a.mp, s_model);
            }
        }
        __syncthreads();                                           // slots reset, search results and the model's table in place
        if (hand_over && searcher) rnd[0] = s_rnd[lane_id()];
        CPH_STAMP(2);
        const Located loc = s_found.loc;
        tc.inv = s_found.inv;
        if (SHARDED) { tc.base0 = s_found.base0; tc.base1 = s_found.base1; tc.basev = s_found.basev; }
        if constexpr (RS == kFixMultinomial) {
            LocatedStrata ls;
            ls.loc = loc; ls.w0 = s_found.w0; ls.w1 = s_found.w1;
            bool mine[kPPT];
            const uint64_t gj = SHARDED ? a.pid0 + (uint64_t)j0 : (uint64_t)j0;            // the lane's first output in the population
            counts_strata_walk<S>(tc, a.strata_offs, a.strata_k, prev_row, a.n, nb, ls, s_found.W, (int64_t)gj, tc.seed, kResampleDrawBase2 + (uint64_t)t, a.pid0 + (uint64_t)j0, anc, L,
                                  s_found.c_lo, s_found.c_hi, last_shard, mine, true);
            if constexpr (SHARDED) {
                // an output whose threshold lies in another rank's range: its ancestor arrived as an annex column, in output order
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
        } else counts_walk<S, SHARDED, RS>(tc, prev_row, a.n, nb, last_shard, gj_first, n_out, loc, guess, raw_m1, raw_0, raw_p1, anc, L);
        if (SHARDED && RS != kFixMultinomial) {
            // the lineages of the outputs other shards' sources own arrived as annex columns, in output order (cpprob_hip exchange
            // commit)
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
    }

    CPH_STAMP(3);
    V prev[kPPT], x[kPPT];
    uint32_t tw[kPPT];
    if (traced) {
        // the ancestor's trace word where its state byte would be gathered: the same dependent load, and the state is its top field
#pragma unroll
        for (int k = 0; k < kPPT; ++k) tw[k] = t > 0 ? a.trace_prev[anc[k]] : 0u;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) prev[k] = t > 0 ? static_cast<V>((tw[k] >> (Model::kTraceBits * (t - 1))) & ((1u << Model::kTraceBits) - 1u)) : V(0);
    } else {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) prev[k] = t > 0 ? static_cast<V>(prev_row[anc[k]]) : V(0);             // ancestor's state (sorted gather)
    }
#pragma unroll
    for (int q = 0; q < kPPT / 4; ++q)                                                        // sample #t
        Model::apply4_staged(s_model, t, rnd[q], reinterpret_cast<const V(&)[4]>(prev[4 * q]), reinterpret_cast<V(&)[4]>(x[4 * q]));
    bool valid[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) valid[k] = j0 + k < a.n;
    store4_as(a.values + (int64_t)a.row_w * a.rs, j0, x);                                     // predict #t
    if (traced) {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) tw[k] |= (uint32_t)x[k] << (Model::kTraceBits * t);
        store4(a.trace_next, j0, tw);
    }
    if (a.anc) store4_write_through(a.anc + (int64_t)t * a.rs, j0, anc);                      // (a filtering-only run keeps no ancestors)

    CPH_STAMP(4);
    // ---- observe #t as counts ----
    uint32_t c0 = 0, c1 = 0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        const int s = Model::weight_index(x[k]);
        c0 += (uint32_t)__popcll(__ballot(valid[k] && s == 0));
        c1 += (uint32_t)__popcll(__ballot(valid[k] && s == 1));
    }
    if (lane_id() == 0) { s_cnt[2 * wave_id()] = (int)c0; s_cnt[2 * wave_id() + 1] = (int)c1; }
    __syncthreads();
    if (tid == 0) {
        uint32_t n0 = 0, n1 = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { n0 += (uint32_t)s_cnt[2 * w]; n1 += (uint32_t)s_cnt[2 * w + 1]; }
        hier_publish(a.h, bid, nb, n0, n1, true);
    }
    CPH_STAMP(5);
}

// Read-out of a single-shard run in the prefix-count form.  The last step is an ordinary step: it leaves its generation's counts
// like any other, and what the read-out needs follows from them and the particles' states -- the final weight of a particle is
// e[x] (three values), the normaliser W the same fma chain over the generation's totals that every step's prologue evaluates.  So
// no log-weight / linear-weight arrays are written, no tile partials, and no normalisation launch sits between the last step and
// the lineage walk: workgroup 0's first wavefront does the bookkeeping of the final generation on its way in.
struct CountsFinal {
    Hier h;                      // the final generation's counts (the copy the last step wrote)
    double e[4];                 // exp(ll_s - max ll) of the last step, then max ll
    double n_pop; int T;
    int bookkeep;                // 0: the run's bookkeeping is done (a joint population's, by counts_final_ctrl_kernel; a repeated read-out)
    StepCtrl* ctrl; double* ess_trace; int32_t* resampled;
    double* filter_stats;        // filtering-only run: [T][3]; the final generation's row is written with its bookkeeping
};

// Bookkeeping of the final generation (scan_tail's, for a generation no resampling follows) from its totals.  One thread.
__device__ __forceinline__ void counts_final_bookkeep(const CountsFinal& f, double tot0, double tot1)
{
    const double tot2 = f.n_pop - tot0 - tot1;
    const double W = fma(tot2, f.e[2], fma(tot1, f.e[1], __dmul_rn(tot0, f.e[0])));
    const double Q = fma(tot2, __dmul_rn(f.e[2], f.e[2]), fma(tot1, __dmul_rn(f.e[1], f.e[1]), __dmul_rn(tot0, __dmul_rn(f.e[0], f.e[0]))));
    const double ess = W * W / Q;
    StepCtrl* c = f.ctrl;
    c->M = f.e[3]; c->W = W; c->Q = Q; c->ess = ess; c->do_resample = 0;
    c->cdf_lo = 0.0; c->w_local = W; c->scale = 1.0; c->lw_after = 0.0; c->inv_stepw = f.n_pop / W; c->inv_global = f.n_pop / W;
    const double lz = (f.T == 1) ? 0.0 : c->log_z;
    if (f.T == 1) c->n_resampled = 0;
    c->log_z = lz + (f.e[3] + log(W / f.n_pop));
    if (f.ess_trace) f.ess_trace[f.T - 1] = ess;
    if (f.resampled) f.resampled[f.T - 1] = 0;
    if (f.filter_stats) {
        double* fs = f.filter_stats + 3 * (f.T - 1);
        fs[0] = __dmul_rn(tot0, f.e[0]) / W; fs[1] = __dmul_rn(tot1, f.e[1]) / W; fs[2] = __dmul_rn(tot2, f.e[2]) / W;
    }
}

// The same for one shard of a joint population, from the all-gathered {n_0, n_1, particles} of every rank (exact doubles: the sums
// are those a single GPU's hierarchy would hold, so the evidence does not depend on how the population is sharded).  One wavefront.
__global__ __launch_bounds__(kWave) void counts_final_ctrl_kernel(CountsFinal f, const double* __restrict__ all_totals, int world)
{
    const int lane = threadIdx.x;
    double r0 = 0.0, r1 = 0.0;
    if (lane < world) { r0 = all_totals[3 * lane]; r1 = all_totals[3 * lane + 1]; }
    const double tot0 = wave_sum(r0), tot1 = wave_sum(r1);
    if (lane == 0) counts_final_bookkeep(f, tot0, tot1);
}

// A filtering-only run has no lineages to walk: the final generation's bookkeeping (and its row of statistics) is the whole read-out.
__global__ __launch_bounds__(kWave) void counts_filter_final_kernel(CountsFinal f)
{
    const Cnt2 tl = hier_total(f.h);
    if (threadIdx.x == 0) counts_final_bookkeep(f, (double)tl.n0, (double)tl.n1);
}

template <class Model>
__global__ __launch_bounds__(kThreads) void smooth_counts_kernel(SmoothArgs<Model> a, CountsFinal f)
{
    extern __shared__ __attribute__((aligned(16))) double s_stat[];   // [kWaves][T*K]
    if (f.bookkeep && blockIdx.x == 0 && wave_id() == 0) {
        const Cnt2 tl = hier_total(f.h);
        if (threadIdx.x == 0) counts_final_bookkeep(f, (double)tl.n0, (double)tl.n1);
    }
    const double e0 = f.e[0], e1 = f.e[1], e2 = f.e[2];
    const typename Model::store_t* last = a.values + (int64_t)(a.T - 1) * a.rs;
    const int64_t n = a.n;
    smooth_body<Model>(a, s_stat, [last, n, e0, e1, e2](int64_t, int64_t i) {
        const int s = Model::weight_index(static_cast<typename Model::value_t>(last[i]));
        return i < n ? (s == 0 ? e0 : (s == 1 ? e1 : e2)) : 0.0;
    });
}

// Log-weights of the final generation, for the callers that ask for them (cpprob_hip_copy_logw): ll[x], -inf in padding slots.
template <class Model>
__global__ __launch_bounds__(kThreads) void logw_from_states_kernel(const typename Model::store_t* __restrict__ last, int64_t n, int64_t ld,
                                                                     double l0, double l1, double l2, double* __restrict__ logw)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= ld) return;
    const int s = Model::weight_index(static_cast<typename Model::value_t>(last[i]));
    logw[i] = i < n ? (s == 0 ? l0 : (s == 1 ? l1 : l2)) : -INFINITY;
}

// {n_0, n_1, particles} of this shard's generation as exact doubles: what a sharded run all-gathers between two steps.
__global__ __launch_bounds__(kWave) void counts_totals_kernel(Hier h, double n_local, double* __restrict__ out)
{
    const Cnt2 t = hier_total(h);
    if (threadIdx.x == 0) { out[0] = (double)t.n0; out[1] = (double)t.n1; out[2] = n_local; }
}

}  // namespace cph
