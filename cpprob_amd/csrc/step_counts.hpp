// SMC step for table-weight models on an every-step resampling schedule, built on INTEGER prefix counts.
//
// When every particle enters a step at the same log-weight (t = 0, or the previous step resampled) and the model's
// incremental weight takes one of K = 3 values (HMM: log N(y_t; mean[s], 1), reference include/models/models.hpp:130-139),
// the weight of particle k of generation t-1 is e[x_k], e[s] = exp(ll_s - max ll), and its inclusive CDF value is a function of
// the prefix counts c_s(k) = #{i <= k : x_i = s} alone:
//        C_k = fma(c_2, e_2, fma(c_1, e_1, c_0 * e_0)),   G_k = ceil(fma(C_k, N / W, -u0)),   ancestor of output j = min{k : G_k > j}
// (oracle/cpprob_oracle.c::orc_resample_table_systematic states the same arithmetic).  No running floating-point sum exists, so
// tiles, wavefronts and shards may evaluate it in any order and still produce the same integers: ancestors are bit-exact against
// the oracle at every population size and over any number of GPUs.
//
// The prefix counts live in a 64-ary hierarchy written by the kernel that produced the generation:
//   level 0     one entry per 1024-particle tile   (plain store by the tile's workgroup)
//   level l     one entry per 64^l tiles            (64-bit integer atomic add by every workgroup below it; exact, order-free)
// entry = n_0 | n_1 << 32 (n_2 follows from the number of valid particles).  A workgroup that needs the prefix at tile c sums, per
// level, the < 64 entries that precede c's block inside its parent block: one masked load per level per lane and ONE wavefront
// reduction -- instead of every workgroup re-reading every tile partial (the r01 prologue: 3.9 of 12.3 us, 23 MB of L2 reads per step).
// No LDS table, no workgroup barrier, any population size, no normalisation launch between steps.
// Three copies rotate: step t reads copy t % 3 (generation t-1), adds into copy (t + 1) % 3 and clears copy (t + 2) % 3.
#pragma once
#include "kernels.hpp"

namespace cph {

constexpr int kHierMaxLevels = 4;              // 64^4 tiles of 1024 particles exceed the int32 particle index range

struct Hier {
    uint64_t* lvl[3][kHierMaxLevels];          // [copy][level]
    int n_ent[kHierMaxLevels];                 // entries per level; n_ent[0] = tiles
    int n_lev;                                 // levels in use: the last one has <= 64 entries
};

// ---- 32-bit wavefront sums / scans (one instruction per DPP step) -------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += dpp_u32<kDppRowShr1>(v);
    v += dpp_u32<kDppRowShr2>(v);
    v += dpp_u32<kDppRowShr4>(v);
    v += dpp_u32<kDppRowShr8>(v);
    v += dpp_u32<kDppRowBcast15, 0xA>(v);
    v += dpp_u32<kDppRowBcast31, 0xC>(v);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan_u32(v), kWave - 1); }

struct Cnt2 { uint32_t n0, n1; };
__device__ __forceinline__ Cnt2 unpack2(uint64_t w) { return Cnt2{(uint32_t)w, (uint32_t)(w >> 32)}; }

// The tile-level CDF of generation t-1 in the canonical arithmetic.  Counts of the shards that precede this one enter as exact
// doubles (integers below 2^53), so a sharded run evaluates the very expression a single GPU would.
struct TableCdf {
    double e0, e1, e2, inv, u0, n_pop;
    double base0, base1, basev;                // states 0 / 1 and particles before this shard
    __device__ __forceinline__ double cdf(double c0, double c1, double cv) const      // c*: GLOBAL inclusive counts
    {
        const double c2 = cv - c0 - c1;
        return fma(c2, e2, fma(c1, e1, __dmul_rn(c0, e0)));
    }
    __device__ __forceinline__ double g(double C) const { return fmin(fmax(ceil(fma(C, inv, -u0)), 0.0), n_pop); }
    // first output owned by the sources that follow `n0, n1` state-0/1 particles among `nv` local particles
    __device__ __forceinline__ double g_at(uint32_t n0, uint32_t n1, int64_t nv) const
    {
        return g(cdf(base0 + (double)n0, base1 + (double)n1, basev + (double)nv));
    }
};

// Exclusive prefix counts at tile c (wave-uniform result; every lane of the calling wave takes part): per level, the entries that
// precede c's block inside its parent block.
__device__ __forceinline__ Cnt2 hier_prefix(const Hier& h, int copy, int c)
{
    const int lane = lane_id();
    uint32_t s0 = 0, s1 = 0;
#pragma unroll
    for (int l = 0; l < kHierMaxLevels; ++l) {
        if (l < h.n_lev) {
            const int blk = c >> (6 * l);                       // c's block at this level
            const int first = (blk >> 6) << 6;                  // first block of the parent
            if (lane < (blk & 63)) {
                const uint64_t w = h.lvl[copy][l][first + lane];
                s0 += (uint32_t)w; s1 += (uint32_t)(w >> 32);
            }
        }
    }
    return Cnt2{wave_sum_u32(s0), wave_sum_u32(s1)};
}

// Totals of the generation: the sum of the (<= 64) top-level entries.
__device__ __forceinline__ Cnt2 hier_total(const Hier& h, int copy)
{
    const int lane = lane_id();
    const int top = h.n_lev - 1;
    uint64_t w = 0;
    if (lane < h.n_ent[top]) w = h.lvl[copy][top][lane];
    return Cnt2{wave_sum_u32((uint32_t)w), wave_sum_u32((uint32_t)(w >> 32))};
}

// Largest tile c in [0, nb) whose first owned output G(prefix(c)) is <= g (0 when there is none), with its exclusive prefix
// counts: top-down descent, one load + one scan per level.  Wave-uniform; used when the answer is not next to the caller's guess
// (very uneven tile masses) and by the exchange scope's packing, whose outputs sit at the ends of the shard.
__device__ __forceinline__ int hier_locate(const Hier& h, int copy, const TableCdf& tc, int64_t n, double g, Cnt2& P)
{
    const int lane = lane_id();
    int blk = 0;
    uint32_t p0 = 0, p1 = 0;
    for (int l = h.n_lev - 1; l >= 0; --l) {
        const int idx = (blk << 6) + lane;
        uint64_t w = 0;
        const bool in = idx < h.n_ent[l];
        if (in) w = h.lvl[copy][l][idx];
        const uint32_t i0 = wave_incl_scan_u32((uint32_t)w), i1 = wave_incl_scan_u32((uint32_t)(w >> 32));
        const uint32_t x0 = p0 + i0 - (uint32_t)w, x1 = p1 + i1 - (uint32_t)(w >> 32);     // exclusive prefix at child `lane`
        const int64_t tile0 = (int64_t)idx << (6 * l);                                         // first tile of the child
        const int64_t nv = tile0 * kTile < n ? tile0 * kTile : n;
        const bool ok = in && tc.g_at(x0, x1, nv) <= g;
        const unsigned long long m = __ballot(ok);
        const int child = m ? (63 - __builtin_clzll(m)) : 0;                                     // (G is monotone: the set is a prefix)
        p0 = (uint32_t)__builtin_amdgcn_readlane((int)x0, child);
        p1 = (uint32_t)__builtin_amdgcn_readlane((int)x1, child);
        blk = (blk << 6) + child;
    }
    P = Cnt2{p0, p1};
    return blk;
}

struct CountsLds {
    int32_t slot[kTile];        // scatter slots of the output tile
    uint32_t scan[2][kWaves];   // packed per-wave totals of the in-tile scan, double-buffered across source tiles
    int iscr[kWaves];
};

// Ancestors of the kTile consecutive outputs starting at global output index gj_first (n_out of them), among THIS shard's sources,
// -1 where the ancestor belongs to a shard that precedes this one; outputs at or beyond o_hi = G(all local sources) belong to the
// shards that follow (the caller tests that).  `guess` = a tile expected to hold the first ancestor (window[] = prefix counts of
// tiles guess-1 .. guess+3 when the caller has them: have_window), `own` = the states of tile `guess` fetched at kernel entry.
// Slots must hold -1 and be visible (the caller's barrier) on entry.
template <class S>
__device__ __forceinline__ void ancestors_counts(const Hier& h, int copy, const TableCdf& tc, const S* __restrict__ states, int64_t n, int nb,
                                                 bool last_shard, double gj_first, int n_out, int guess, uint32_t own_raw,
                                                 int32_t (&anc)[kPPT], CountsLds& L)
{
    static_assert(sizeof(S) == 1 && kPPT == 4, "states travel as one byte: 4 per lane = one dword");
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const double gj_last = gj_first + (double)(n_out - 1);
    // ---- prefix counts of tiles cs .. cs+4, cs = max(guess - 1, 0): one hierarchical sum + four tile entries ----
    const int cs = guess > 0 ? guess - 1 : 0;
    Cnt2 P = hier_prefix(h, copy, cs);
    uint64_t we = 0;
    if (lane < 4 && cs + lane < nb) we = h.lvl[copy][0][cs + lane];
    uint32_t w0[5], w1[5];
    w0[0] = P.n0; w1[0] = P.n1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        w0[i + 1] = w0[i] + (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)we, i);
        w1[i + 1] = w1[i] + (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(we >> 32), i);
    }
    auto nvalid_before = [&](int c) -> int64_t { const int64_t v = (int64_t)c * kTile; return v < n ? v : n; };
    int i_lo = -1;
#pragma unroll
    for (int i = 0; i < 5; ++i)
        if (cs + i < nb && tc.g_at(w0[i], w1[i], nvalid_before(cs + i)) <= gj_first) i_lo = i;
    int c;
    if ((i_lo >= 0 || cs == 0) && i_lo < 4) {
        const int i = i_lo < 0 ? 0 : i_lo;
        c = cs + i;
        P.n0 = i == 0 ? w0[0] : (i == 1 ? w0[1] : (i == 2 ? w0[2] : w0[3]));
        P.n1 = i == 0 ? w1[0] : (i == 1 ? w1[1] : (i == 2 ? w1[2] : w1[3]));
    } else {
        c = hier_locate(h, copy, tc, n, gj_first, P);                 // rare: tile masses far from even
    }
    // ---- walk the source tiles that own outputs of this tile ----
    auto load_states = [&](int cc) -> uint32_t {
        return cc < nb ? *reinterpret_cast<const uint32_t*>(states + (int64_t)cc * kTile + (int64_t)tid * kPPT) : 0u;
    };
    uint32_t raw = (c == guess) ? own_raw : load_states(c);
    int it = 0;
    c = __builtin_amdgcn_readfirstlane(c);
    while (c < nb) {
        // (wave-uniform values -- the branch is made scalar so that the barrier inside the loop sits in uniform control flow)
        if (__builtin_amdgcn_readfirstlane(tc.g_at(P.n0, P.n1, nvalid_before(c)) > gj_last ? 1 : 0)) break;   // the tile's sources start beyond this output tile
        const uint32_t raw_next = (c + 1 == guess) ? own_raw : load_states(c + 1);      // travels while this tile is processed
        const int64_t i0 = (int64_t)c * kTile + (int64_t)tid * kPPT;
        // per-lane inclusive counts of states 0 / 1, packed 16 + 16 bits
        uint32_t q[kPPT];
        uint32_t run = 0;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const uint32_t s = (raw >> (8 * k)) & 0xffu;
            const bool valid = i0 + k < n;
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
        ++it;
        const uint32_t excl = off + incl - run;                           // packed exclusive prefix of this lane
        const double b0 = tc.base0 + (double)P.n0, b1 = tc.base1 + (double)P.n1;     // (uniform)
        const double bv = tc.basev + (double)nvalid_before(c);
        const int64_t nv_tile = n - (int64_t)c * kTile;                  // valid particles from this tile on (>= 1)
        const int vb = tid * kPPT;                                       // particles of this tile before the lane's first
        auto gk = [&](uint32_t packed, int upto) -> double {             // G after `upto` particles of the tile, `packed` of them in states 0 / 1
            const double c0 = b0 + (double)(packed & 0xffffu), c1 = b1 + (double)(packed >> 16);
            const double cv = bv + (double)((int64_t)upto < nv_tile ? (int64_t)upto : nv_tile);
            return tc.g(tc.cdf(c0, c1, cv));
        };
        double g_prev = gk(excl, vb);
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            double g = gk(excl + q[k], vb + k + 1);
            if (last_shard && i0 + k + 1 == n) g = tc.n_pop;             // the population's last source owns the rest
            if (g > g_prev) {
                const double s = g_prev - gj_first, e = g - gj_first;    // exact: integers
                if (e > 0.0 && s < (double)kTile) L.slot[s > 0.0 ? (int)s : 0] = (int32_t)(i0 + k);
                g_prev = g;
            }
        }
        P.n0 += tot & 0xffffu; P.n1 += tot >> 16;
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

template <class Model>
struct StepCountsArgs {
    ModelParams mp; int t, T; int64_t n, ld, rs;
    uint64_t seed, pid0;
    typename Model::store_t* values; int32_t* anc;
    double* logw_next; double* wrel_next; Partial* part;       // written by the last step only (the read-out's inputs)
    Hier h;
    StepCtrl* ctrl; double n_pop; double* ess_trace; int32_t* resampled;
    // one shard of a joint population (exchange scope): the all-gathered {n_0, n_1, particles} of every rank's generation t-1,
    // exact doubles; nullptr on a single shard
    const double* all_totals; int world, rank;
    const int64_t* annex_base;                                  // [T + 1]: annex columns in use before the immigrants of step t arrive
};

template <class Model>
__global__ __launch_bounds__(kThreads) void smc_step_counts_kernel(StepCountsArgs<Model> a)
{
    using V = typename Model::value_t;
    using S = typename Model::store_t;
    static_assert(Model::kWeightTable == 3, "prefix-count form: three table values (two stored counts)");
    __shared__ CountsLds L;
    __shared__ double s_scr[3 * kWaves];
    __shared__ int s_cnt[kWaves * 4];
    const int tid = threadIdx.x;
    const int nb = (int)gridDim.x;
    const int bid = xcd_contiguous_tile((int)blockIdx.x, nb);
    const int64_t j0 = (int64_t)bid * kTile + (int64_t)tid * kPPT;
    const int t = a.t;
    const int copy_prev = t % 3, copy_next = (t + 1) % 3, copy_clear = (t + 2) % 3;

    typename Model::Rand rnd[kPPT / 4];
#pragma unroll
    for (int q = 0; q < kPPT / 4; ++q) Model::draw4(a.seed, a.pid0 + (uint64_t)j0 + 4 * q, t, rnd[q]);
    const S* prev_row = a.values + (int64_t)(t > 0 ? t - 1 : 0) * a.rs;

    int32_t anc[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) anc[k] = (int32_t)(j0 + k);
    if (t > 0) {
        const uint32_t own_raw = *reinterpret_cast<const uint32_t*>(prev_row + j0);
        {
            int32_t neg[kPPT];
            lane_fill(neg, (int32_t)-1);
            store4(L.slot, (int64_t)tid * kPPT, neg);
        }
        // ---- the generation's totals and this shard's place in the joint population (every wave, identically) ----
        double ll[3], et[3], mref;
        Model::weight_table(a.mp, t - 1, ll, et, mref);
        TableCdf tc;
        tc.e0 = et[0]; tc.e1 = et[1]; tc.e2 = et[2]; tc.n_pop = a.n_pop;
        double tot0, tot1;
        bool last_shard = true;
        if (a.all_totals) {
            const int lane = lane_id();
            double r0 = 0.0, r1 = 0.0, rv = 0.0;
            if (lane < a.world) { r0 = a.all_totals[3 * lane]; r1 = a.all_totals[3 * lane + 1]; rv = a.all_totals[3 * lane + 2]; }
            const bool before = lane < a.rank;
            tc.base0 = wave_sum(before ? r0 : 0.0); tc.base1 = wave_sum(before ? r1 : 0.0); tc.basev = wave_sum(before ? rv : 0.0);
            tot0 = wave_sum(r0); tot1 = wave_sum(r1);              // (sums of integers below 2^53: exact in any order)
            last_shard = a.rank + 1 == a.world;
        } else {
            const Cnt2 tl = hier_total(a.h, copy_prev);
            tc.base0 = 0.0; tc.base1 = 0.0; tc.basev = 0.0;
            tot0 = (double)tl.n0; tot1 = (double)tl.n1;
        }
        tc.inv = 1.0; tc.u0 = 0.0;
        const double W = tc.cdf(tot0, tot1, a.n_pop);
        tc.inv = a.n_pop / W;
        tc.u0 = a.ctrl->u0_pp[t & 1];                              // left there by workgroup 0 of step t-1
        if (bid == 0 && tid == 0) {                                // bookkeeping of step t-1 for the host: ESS (thesis p.37), evidence
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
        }
        __syncthreads();                                           // slots reset
        const int64_t rem = a.n - (int64_t)bid * kTile;
        const int n_out = rem < kTile ? (int)rem : kTile;
        const double gj_first = (double)(a.pid0 + (uint64_t)bid * kTile);
        ancestors_counts<S>(a.h, copy_prev, tc, prev_row, a.n, nb, last_shard, gj_first, n_out, bid, own_raw, anc, L);
        if (a.all_totals) {
            // outputs below o_lo / at or beyond o_hi descend from other shards' sources: their lineages arrived as annex columns,
            // in output order (cpprob_hip exchange commit)
            const Cnt2 tl = hier_total(a.h, copy_prev);
            const double o_lo = tc.g_at(0, 0, 0), o_hi = last_shard ? a.n_pop : tc.g_at(tl.n0, tl.n1, a.n);
            const double sb = (double)a.pid0;
            const int64_t l0 = (int64_t)fmin(fmax(o_lo - sb, 0.0), (double)a.n), l1 = (int64_t)fmin(fmax(o_hi - sb, 0.0), (double)a.n);
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

    V prev[kPPT], x[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) prev[k] = t > 0 ? static_cast<V>(prev_row[anc[k]]) : V(0);                 // ancestor's state (sorted gather)
#pragma unroll
    for (int q = 0; q < kPPT / 4; ++q)                                                        // sample #t
        Model::apply4(a.mp, t, rnd[q], reinterpret_cast<const V(&)[4]>(prev[4 * q]), reinterpret_cast<V(&)[4]>(x[4 * q]));
    bool valid[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) valid[k] = j0 + k < a.n;
    store4_as(a.values + (int64_t)t * a.rs, j0, x);                                           // predict #t
    store4_write_through(a.anc + (int64_t)t * a.rs, j0, anc);

    if (t + 1 == a.T) {
        // last step: the read-out wants log-weights, linear weights and an fp64 tile partial (observe #t: table look-ups only)
        double ll[3], et[3], mref;
        Model::weight_table(a.mp, t, ll, et, mref);
        int idx[kPPT]; double lw[kPPT], e[kPPT];
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            idx[k] = Model::weight_index(x[k]);
            const double l = idx[k] == 0 ? ll[0] : (idx[k] == 1 ? ll[1] : ll[2]);
            lw[k] = valid[k] ? l : -INFINITY;
        }
        tile_partial_table<3>(idx, valid, et, mref, e, a.part, s_cnt, false, bid);
        store4(a.logw_next, j0, lw);
        store4(a.wrel_next, j0, e);
        (void)s_scr;
        return;
    }
    // ---- observe #t as counts: this tile's entry of the hierarchy, added into every level above ----
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
        const uint64_t ent = (uint64_t)n0 | ((uint64_t)n1 << 32);
        a.h.lvl[copy_next][0][bid] = ent;
        for (int l = 1; l < a.h.n_lev; ++l)
            atomicAdd(reinterpret_cast<unsigned long long*>(a.h.lvl[copy_next][l] + (bid >> (6 * l))), (unsigned long long)ent);
        for (int l = 1; l < a.h.n_lev; ++l)
            if (bid < a.h.n_ent[l]) a.h.lvl[copy_clear][l][bid] = 0;
        if (bid == 0) {                                            // systematic offset of the resampling before step t+1
            const u32x4 r = draw_block(a.seed, 0, kResampleDrawBase + (uint64_t)(t + 1));
            a.ctrl->u0_pp[(t + 1) & 1] = u01_53(r.x, r.y);
        }
    }
}

// {n_0, n_1, particles} of this shard's generation as exact doubles: what a sharded run all-gathers between two steps.
__global__ __launch_bounds__(kWave) void counts_totals_kernel(Hier h, int copy, double n_local, double* __restrict__ out)
{
    const Cnt2 t = hier_total(h, copy);
    if (threadIdx.x == 0) { out[0] = (double)t.n0; out[1] = (double)t.n1; out[2] = n_local; }
}

}  // namespace cph
