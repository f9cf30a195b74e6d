// Exchange scope (exact global resampling over shards): the device-resident plan and the kernels that move lineages.
//
// One joint population is sharded contiguously over `world` ranks.  With systematic resampling -- one shared offset -- the sources of
// rank r own the outputs [o_r, o_{r+1}), o_r = G(CDF at the start of shard r): a function of the all-gathered rank totals alone, so
// every rank derives the SAME plan on its own device, with no host in the loop:
//     output j lives on the rank whose shard contains j; the part of [o_rank, o_rank+1) outside the own shard is sent (the
//     lineage x_0..x_t of each such output's ancestor), the part of the own shard outside it arrives, in output order.
// exchange_plan_kernel   o_r -> per-peer send / receive intervals, offsets into the transport buffers, annex bookkeeping, overflow flag
// exchange_pack_kernel   ancestors of the outputs this shard's sources own elsewhere (the step kernels' own search) + their lineages
// exchange_commit_kernel received lineages -> annex columns of the particle store (identity ancestors): later kernels see ordinary particles
// With remote lineages (the default where every rank can address every rank's store: ExchangeGeom::remote) a migrant is its current
// state, an (origin rank, slot) word and -- short discrete traces -- its trace word, stored by the packing kernel straight into the
// RECEIVING rank's annex: no transport buffer, no commit launch, and the lineage is followed across ranks only when somebody reads it.
// Otherwise: transport buffers hold one fixed-capacity segment per peer slot (sizes are host constants: RCCL send/recv counts cannot depend on
// device data), or -- for callers that synchronise and size their buffers exactly -- one compact block per rank (cpprob_hip_exchange_*).
// A step that does not resample, or whose shards' masses happen to match, plans zero records; its transfers carry stale bytes nobody reads.
#pragma once
#include "kernels.hpp"
#include "step_counts.hpp"
#include "step_fixed.hpp"

namespace cph {

// The plan lives on ONE wavefront whose lane r holds o_r for r = 0 .. world (world + 1 bounds): world <= 63.
constexpr int kMaxWorld = 63;
constexpr int kWorldSlots = 64;                 // per-rank arrays (a power of two)

struct ExchangePlan {
    int32_t resample;                 // the step's decision (device-side)
    int32_t overflow;                 // sticky per run, bit set: 1 = a peer segment was too small, 2 = a rank outside the peer list was needed, 4 = the annex was too small
    int64_t l0, l1;                   // local outputs [l0, l1) descend from local sources
    int64_t n_send, n_recv;
    int64_t send_lo[kWorldSlots];       // per RANK: first global output of mine that lives there, how many, where its records start (in records)
    int64_t send_cnt[kWorldSlots];
    int64_t send_base[kWorldSlots];
    int64_t recv_cnt[kWorldSlots];      // per RANK: records that arrive from it, where they sit in the receive buffer, their first annex column offset
    int64_t recv_base[kWorldSlots];
    int64_t recv_off[kWorldSlots];
    int64_t run_records, run_bytes;   // what this rank's sources sent since step 0 of the run: lineage records, and their bytes (records x (t + 1) x value size)
    int64_t src_shift;                // tiles by which this shard's outputs sit off its sources: output tile b draws from source tile ~ b + src_shift
                                      // (the next step kernel aims its first probe and its prefetch there: a sharded CDF's offset is O(sqrt N) outputs)
    int64_t dst_col[kWorldSlots];     // remote lineages, per RANK: the annex column of THAT rank my first record for it goes to
};

struct ExchangeGeom {
    int world, rank;
    int64_t n;                        // local particles
    const int64_t* shard_begin;       // [world + 1] device
    const int32_t* slot_of_rank;      // [world] device: transport slot of each rank (-1: not a peer in this mode); nullptr: compact layout
    int64_t cap;                      // records per slot (fixed layout)
    int64_t annex_cap;
    int bytes_per_value;              // of the transported records (traffic accounting)
    int no_history;                   // filtering-only shards (keep_history = 0): a migrant is its current state alone -- records of ONE
                                      // value, and the annex is reused by every step's immigrants
    int remote;                       // remote lineages: a migrant is its current state and the slot it sits in on the rank it leaves --
                                      // its history STAYS there, and whoever walks the lineage later (read-out, trace dump) continues in
                                      // that rank's particle store through the peer mapping (cpprob_hip_exchange_remote).  The packing
                                      // kernel stores both straight into the receiving rank's annex column and origin table: every rank
                                      // keeps every rank's annex fill (a function of the all-gathered totals), so no receive buffer, no
                                      // segment capacity, no peer list and no commit launch
    int trace_words;                  // (remote) a migrant also takes its 4-byte trace word along (trace_words.hpp)
    const RemoteStores* rem;          // (remote) every rank's store as this device addresses it
    int64_t* annex_all;               // (remote) [T + 1][kWorldSlots] device: annex columns in use on every rank before each step's exchange
    int64_t* sent_per_step;           // [T] device, may be nullptr: records this rank sent after each step
};

struct PlanCountsIn {                 // prefix-count form: o_r from the all-gathered {n_0, n_1, particles}
    const double* all_totals; double e0, e1, e2, u0, n_pop;
    int rs; uint64_t seed, draw;      // kFixSystematic / kFixStratified (the outputs' uniforms: Philox key, draw index)
};

// The plan as one wavefront holds it: lane r = what concerns rank r, plus the wave-uniform part.
struct PlanLane { int64_t send_lo, send_cnt, send_base, recv_cnt, recv_base, recv_off, dst_col, fill_next; };
struct PlanWave { int64_t l0, l1, n_send, n_recv, src_shift; int32_t flags; bool resample; };

// o_r (lane r, r = 0 .. world) from the all-gathered {n_0, n_1, particles} of every rank: canonical integer arithmetic, the step
// kernel's own expressions.  One wave, every lane.
__device__ __forceinline__ double plan_bounds_counts(const PlanCountsIn& pc, int world)
{
    const int lane = lane_id();
    double r0 = 0.0, r1 = 0.0, rv = 0.0;
    if (lane < world) { r0 = pc.all_totals[3 * lane]; r1 = pc.all_totals[3 * lane + 1]; rv = pc.all_totals[3 * lane + 2]; }
    // exclusive prefix over ranks: sums of integers below 2^53 -- exact, so every rank (and the step kernel's masked sums) agree
    const double i0 = wave_incl_scan(r0), i1 = wave_incl_scan(r1), iv = wave_incl_scan(rv);
    TableCdf tc;
    tc.e0 = pc.e0; tc.e1 = pc.e1; tc.e2 = pc.e2; tc.u0 = pc.u0; tc.n_pop = pc.n_pop;
    tc.base0 = 0.0; tc.base1 = 0.0; tc.basev = 0.0;
    const double W = tc.cdf(read_lane(i0, kWave - 1), read_lane(i1, kWave - 1), pc.n_pop);
    tc.inv = pc.n_pop / W;
    tc.seed = pc.seed; tc.draw = pc.draw; tc.uid0 = 0;
    const double Cb = tc.cdf(i0 - r0, i1 - r1, iv - rv);
    double o = pc.rs == kFixStratified ? tc.first_stratified(Cb) : tc.g(Cb);
    if (lane >= world) o = pc.n_pop;
    return o;
}

// From the bounds to the plan: my sources' interval and what each rank's shard takes of it / gives to me.  One wave, every lane.
__device__ __forceinline__ void plan_wave(const ExchangeGeom& g, int t, double o, bool resample, PlanLane& pl, PlanWave& pw)
{
    const int lane = lane_id();
    const int world = g.world, rank = g.rank;
    const double my_lo = read_lane(o, rank), my_hi = read_lane(o, rank + 1);
    const double o_next = dpp_or<0x130 /* wave_shl:1 */>(o, 0.0);        // lane r: o_{r+1} (world <= 63: lane world exists)
    auto clampd = [](double v, double lo, double hi) { return fmin(fmax(v, lo), hi); };
    int64_t send_lo = 0, send_cnt = 0, recv_cnt = 0, l0 = 0, l1 = g.n;
    int32_t flag = 0;
    const double mb = (double)g.shard_begin[rank], me = (double)g.shard_begin[rank + 1];
    if (lane < world && resample) {
        const double sb = (double)g.shard_begin[lane], se = (double)g.shard_begin[lane + 1];
        if (lane != rank) {
            const double lo = clampd(my_lo, sb, se), hi = clampd(my_hi, sb, se);          // my sources' outputs that live on rank `lane`
            send_lo = (int64_t)lo; send_cnt = (int64_t)(hi - lo);
            const double rl = clampd(o, mb, me), rh = clampd(o_next, mb, me);             // rank `lane`'s sources' outputs that live here
            recv_cnt = (int64_t)(rh - rl);
        }
    }
    if (resample) { l0 = (int64_t)(clampd(my_lo, mb, me) - mb); l1 = (int64_t)(clampd(my_hi, mb, me) - mb); }
    pl.dst_col = 0; pl.fill_next = 0;
    if (g.remote && lane < world) {
        // rank `lane`'s annex: what it holds (every rank keeps the same table), what this step adds -- its shard less the outputs its
        // own sources keep there -- and where my records land in it: behind those of the ranks before me (receive order = rank order)
        const double sb = (double)g.shard_begin[lane], se = (double)g.shard_begin[lane + 1];
        const int64_t fill = t == 0 ? 0 : g.annex_all[(int64_t)t * kWorldSlots + lane];
        const int64_t kept = resample ? (int64_t)(clampd(o_next, sb, se) - clampd(o, sb, se)) : 0;
        const int64_t arrive = resample ? (int64_t)(se - sb) - kept : 0;
        const int64_t room = g.rem->rs[lane] - g.rem->ld[lane];
        pl.fill_next = fill + (fill + arrive > room ? 0 : arrive);
        if (resample && lane != rank) pl.dst_col = fill + (int64_t)(clampd(my_lo, sb, se) - sb) - (lane < rank ? kept : 0);
    }
    pw.src_shift = resample ? (int64_t)floor((mb - my_lo) * (1.0 / kTile)) : 0;        // output mb is my sources' output number mb - my_lo
    // layout: fixed slots (capacity-checked) or compact blocks in rank order
    int64_t send_base = 0, recv_base = 0;
    const uint32_t rs_incl = wave_incl_scan_u32((uint32_t)recv_cnt), ss_incl = wave_incl_scan_u32((uint32_t)send_cnt);
    if (g.slot_of_rank) {
        const int slot = lane < world ? g.slot_of_rank[lane] : -1;
        if (lane < world && lane != rank) {
            if (slot < 0) { if (send_cnt | recv_cnt) flag = 2; send_cnt = 0; recv_cnt = 0; }
            else {
                if (send_cnt > g.cap || recv_cnt > g.cap) { flag = 1; send_cnt = send_cnt < g.cap ? send_cnt : g.cap; recv_cnt = recv_cnt < g.cap ? recv_cnt : g.cap; }
                send_base = (int64_t)slot * g.cap; recv_base = (int64_t)slot * g.cap;
            }
        }
    } else {
        send_base = (int64_t)(ss_incl - (uint32_t)send_cnt); recv_base = (int64_t)(rs_incl - (uint32_t)recv_cnt);
    }
    // (annex columns follow the receive order = rank order = output order; recomputed after any clamping)
    const uint32_t ro_incl = wave_incl_scan_u32((uint32_t)recv_cnt), so_incl = wave_incl_scan_u32((uint32_t)send_cnt);
    pl.send_lo = send_lo; pl.send_cnt = send_cnt; pl.send_base = send_base;
    pl.recv_cnt = recv_cnt; pl.recv_base = recv_base; pl.recv_off = (int64_t)(ro_incl - (uint32_t)recv_cnt);
    pw.n_recv = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)ro_incl, kWave - 1);
    pw.n_send = (int64_t)(uint32_t)__builtin_amdgcn_readlane((int)so_incl, kWave - 1);
    pw.l0 = l0; pw.l1 = l1; pw.resample = resample;
    pw.flags = (__ballot(flag == 1) ? 1 : 0) | (__ballot(flag == 2) ? 2 : 0);
}

// The plan into memory (what the commit kernel and the next step read), the annex bookkeeping, the traffic counters.  One wave.
__device__ __forceinline__ void plan_store(const ExchangeGeom& g, int t, const PlanLane& pl, const PlanWave& pw, int64_t* __restrict__ annex_base,
                                           ExchangePlan* __restrict__ plan)
{
    const int lane = lane_id();
    const int64_t base = (t == 0 || g.no_history) ? 0 : annex_base[t];
    if (lane < g.world) {
        plan->send_lo[lane] = pl.send_lo; plan->send_cnt[lane] = pl.send_cnt; plan->send_base[lane] = pl.send_base;
        plan->recv_cnt[lane] = pl.recv_cnt; plan->recv_base[lane] = pl.recv_base; plan->recv_off[lane] = pl.recv_off;
        plan->dst_col[lane] = pl.dst_col;
        if (g.remote) g.annex_all[(int64_t)(t + 1) * kWorldSlots + lane] = pl.fill_next;
    }
    if (lane == 0) {
        const bool over_annex = base + pw.n_recv > g.annex_cap;
        plan->resample = pw.resample ? 1 : 0;
        int32_t ov = t == 0 ? 0 : plan->overflow;
        ov |= pw.flags;
        if (over_annex) ov |= 4;
        plan->overflow = ov;
        plan->l0 = pw.l0; plan->l1 = pw.l1; plan->n_send = pw.n_send; plan->n_recv = over_annex ? 0 : pw.n_recv; plan->src_shift = pw.src_shift;
        if (t == 0) annex_base[0] = 0;
        annex_base[t + 1] = g.no_history ? 0 : base + (over_annex ? 0 : pw.n_recv);
        const int64_t rec0 = t == 0 ? 0 : plan->run_records, byt0 = t == 0 ? 0 : plan->run_bytes;
        plan->run_records = rec0 + pw.n_send;
        plan->run_bytes = byt0 + pw.n_send * (g.remote ? (int64_t)g.bytes_per_value + 8 + (g.trace_words ? 4 : 0) : (int64_t)(g.no_history ? 1 : t + 1) * (int64_t)g.bytes_per_value);
        if (g.sent_per_step) g.sent_per_step[t] = pw.n_send;
    }
}

// One wave.  COUNTS: bounds from integer counts (plan_bounds_counts); otherwise from obound[] (scan_exchange_bounds) and its decision word.
// SCAN2: the launch also combines the all-gathered {max, sum, sum of squares} of the ranks into ctrl and the ranks' output bounds
// first (what scan_partials_kernel's phase 2 does on one thread): one launch less per step of a floating-point-form exchange run.
template <bool COUNTS, bool SCAN2 = false>
__global__ __launch_bounds__(kWave) void exchange_plan_kernel(ExchangeGeom g, PlanCountsIn pc, const double* __restrict__ obound, int t,
                                                              int64_t* __restrict__ annex_base, ExchangePlan* __restrict__ plan, ScanArgs sa)
{
    const int lane = lane_id();
    const int world = g.world;
    __shared__ double s_ob[kWorldSlots + 2];
    if (SCAN2) {
        if (lane == 0) {
            scan_combine_ranks(sa);
            scan_tail(sa);
            scan_exchange_bounds(sa);
            for (int r = 0; r <= world + 1; ++r) s_ob[r] = sa.obound[r];      // (this thread's own stores)
        }
        __syncthreads();
    }
    double o = 0.0;                                            // lane r: o_r, r = 0..world
    bool resample;
    if (COUNTS) {
        o = plan_bounds_counts(pc, world);
        resample = true;
    } else {
        if (lane <= world) o = SCAN2 ? s_ob[lane] : obound[lane];
        resample = (SCAN2 ? s_ob[world + 1] : obound[world + 1]) != 0.0;
    }
    PlanLane pl; PlanWave pw;
    plan_wave(g, t, o, resample, pl, pw);
    plan_store(g, t, pl, pw, annex_base, plan);
}

// The same from the ranks' integer masses (fixed-point form): bounds and decision are pure functions of the all-gathered totals.
__global__ __launch_bounds__(kWave) void exchange_plan_fixed_kernel(ExchangeGeom g, PlanFixedIn pf, int t, int64_t* __restrict__ annex_base, ExchangePlan* __restrict__ plan)
{
    bool resample;
    const double o = plan_bounds_fixed(pf, g.world, resample);
    PlanLane pl; PlanWave pw;
    plan_wave(g, t, o, resample, pl, pw);
    plan_store(g, t, pl, pw, annex_base, plan);
}

enum { kPackFloat = 0, kPackCounts = 1, kPackFixed = 2 };

// ---- multinomial resampling (strata form) in the exchange scope: strata_cut.hpp states the plan ---------------------------------
// The ranks' mass bounds, as one wavefront holds them (lane r: P_r, r = 0 .. world; P_world = the population's mass / W).
struct CutArgs {
    int world; const int64_t* shard_begin;
    const uint32_t* offs; int k;                       // this resampling's strata (population-wide): first output of every stratum
    uint64_t seed, draw2;                              // Philox key, kResampleDrawBase2 + step: the outputs' uniforms inside their strata
    const uint64_t* totals_u; double n_pop, ess_frac;  // fixed-point form: all-gathered {S, Q, key(M)}
    const double* totals_d; double e0, e1, e2;         // prefix-count form: all-gathered {n_0, n_1, particles}
    uint32_t* tab; CutHead* head; uint32_t* srccnt;
};
// lane r: the exclusive prefix of the ranks' masses; total: every lane
__device__ __forceinline__ uint64_t cut_bounds_fixed(const uint64_t* __restrict__ all, int world, uint64_t& total, uint64_t& squares)
{
    const int lane = lane_id();
    uint64_t s = 0, q = 0;
    if (lane < world) { s = all[3 * lane]; q = all[3 * lane + 1]; }
    const uint64_t incl = wave_incl_scan_u64(s);
    total = read_lane_u64(incl, kWave - 1); squares = wave_sum_u64(q);
    return incl - s;                                   // (lanes >= world: the total)
}
__device__ __forceinline__ double cut_bounds_counts(const double* __restrict__ all, int world, double e0, double e1, double e2, double n_pop, double& W)
{
    const int lane = lane_id();
    double r0 = 0.0, r1 = 0.0, rv = 0.0;
    if (lane < world) { r0 = all[3 * lane]; r1 = all[3 * lane + 1]; rv = all[3 * lane + 2]; }
    const double i0 = wave_incl_scan(r0), i1 = wave_incl_scan(r1), iv = wave_incl_scan(rv);
    TableCdf tc;
    tc.e0 = e0; tc.e1 = e1; tc.e2 = e2; tc.inv = 1.0; tc.u0 = 0.0; tc.n_pop = n_pop; tc.base0 = 0.0; tc.base1 = 0.0; tc.basev = 0.0; tc.seed = 0; tc.draw = 0; tc.uid0 = 0;
    W = tc.cdf(read_lane(i0, kWave - 1), read_lane(i1, kWave - 1), n_pop);
    return lane < world ? tc.cdf(i0 - r0, i1 - r1, iv - rv) : W;
}

// grid = world - 1: workgroup i looks at boundary b = i + 1 (between the ranks b - 1 and b); workgroup 0 also writes the two ends.
template <bool COUNTS>
__global__ __launch_bounds__(kThreads) void exchange_cut_kernel(CutArgs a)
{
    using T = typename std::conditional<COUNTS, double, uint64_t>::type;
    __shared__ T s_P[kWorldSlots + 1];
    __shared__ uint32_t s_begin[kWorldSlots + 1];
    __shared__ uint32_t s_cnt[kWorldSlots];
    __shared__ uint32_t s_wave[kWaves];
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int world = a.world, b = (int)blockIdx.x + 1;
    const int k = a.k;
    const uint32_t K = 1u << k;
    T Pl, total;
    double unit = 0.0;
    if constexpr (COUNTS) {
        double W;
        Pl = cut_bounds_counts(a.totals_d, world, a.e0, a.e1, a.e2, a.n_pop, W);
        total = W; unit = ldexp(W, -k);
    } else {
        uint64_t S, Q;
        Pl = cut_bounds_fixed(a.totals_u, world, S, Q);
        total = S;
        if (!fixed_decide(S, Q, a.n_pop, a.ess_frac, true).resample) return;     // this generation is not resampled: nothing is cut
    }
    if (wv == 0) { if (lane <= world) s_P[lane] = lane < world ? Pl : total; if (lane <= world) s_begin[lane] = (uint32_t)a.shard_begin[lane]; if (lane < kWorldSlots) s_cnt[lane] = 0u; }
    __syncthreads();
    auto bound = [&](uint32_t w) -> T {
        if constexpr (COUNTS) return (double)w * unit;
        else return strata_bound(total, (uint64_t)w, k);
    };
    const T Pb = s_P[b];
    // lo = min{w in [0, K] : B_w >= P_b}   (B_K = the total >= P_b)
    uint32_t lo = 0, hi_s = K;
    while (lo < hi_s) { const uint32_t mid = (lo + hi_s) >> 1; if (bound(mid) >= Pb) hi_s = mid; else lo = mid + 1; }
    // hi = max{w : B_w <= P_b} (integers) / max{w : B_w < P_b} (table CDF: a threshold below may round up to the bound itself)
    uint32_t hi;
    if constexpr (COUNTS) hi = lo > 0 ? lo - 1 : 0;
    else hi = (bound(lo) == Pb || lo == 0) ? lo : lo - 1;
    const uint32_t A = a.offs[lo], Z = a.offs[hi];
    const uint32_t len = A - Z;
    if (tid == 0) {
        a.head[b] = CutHead{A, Z, len > (uint32_t)kCutCap ? 1u : 0u, 0u};
        if (b == 1) { a.head[0] = CutHead{0u, 0u, 0u, 0u}; a.head[world] = CutHead{s_begin[world], s_begin[world], 0u, 0u}; }
    }
    if (len == 0) return;
    const T b_lo = bound(hi), b_hi = bound(hi + 1);
    uint32_t* tab = a.tab + (size_t)b * kCutRow;
    uint32_t running = 0;
    for (uint32_t base = 0; base < len; base += kThreads) {
        const uint32_t i = base + (uint32_t)tid;
        const bool valid = i < len;
        const uint64_t s = (uint64_t)Z + (valid ? i : 0u);
        const u32x4 blk = draw_block(a.seed, s >> 1, a.draw2);
        const uint64_t bits = (s & 1) ? bits53(blk.z, blk.w) : bits53(blk.x, blk.y);
        T tau;
        if constexpr (COUNTS) tau = fma((double)bits * kTwoPowM53, b_hi - b_lo, b_lo);
        else tau = b_lo + __umul64hi(bits << 11, b_hi - b_lo);
        uint32_t src = 0, dst = 0;
        for (int r = 1; r < world; ++r) { src += s_P[r] <= tau ? 1u : 0u; dst += s_begin[r] <= (uint32_t)s ? 1u : 0u; }       // (P and the shards' begins do not decrease)
        const bool home = valid && src == dst;
        const unsigned long long m = __ballot(home);
        if (lane == 0) s_wave[wv] = (uint32_t)__popcll(m);
        if (valid) atomicAdd(&s_cnt[src], 1u);
        __syncthreads();
        uint32_t off = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) { const uint32_t x = s_wave[w]; if (w < wv) off += x; tot += x; }
        const uint32_t cum = running + off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (valid && i < (uint32_t)kCutCap) tab[i] = (src << 24) | cum;
        running += tot;
        __syncthreads();
    }
    if (tid == 0) tab[len < (uint32_t)kCutCap ? len : (uint32_t)kCutCap] = running;
    if (tid < kWorldSlots) a.srccnt[(size_t)b * kCutSlots + tid] = s_cnt[tid];
}

// The plan of a multinomial exchange (one wave, lane = rank): who keeps how many of its own outputs, hence every rank's arrivals and
// annex fill; this rank's regular sends; its totals.  heads / tables: the cut launch that precedes.
__device__ __forceinline__ void plan_wave_strata(const ExchangeGeom& g, int t, const CutView& cv, bool resample, PlanLane& pl, PlanWave& pw)
{
    const int lane = lane_id();
    const int world = g.world, rank = g.rank;
    pl = PlanLane{0, 0, 0, 0, 0, 0, 0, 0};
    pw = PlanWave{0, g.n, 0, 0, 0, 0, resample};
    int64_t arrive = 0, sourced = 0, kept = 0;
    uint32_t over = 0;
    if (resample && lane < world) {
        const uint32_t sb = (uint32_t)g.shard_begin[lane], se = (uint32_t)g.shard_begin[lane + 1];
        const KeptCtx kc = kept_ctx(cv, lane, sb);
        kept = (int64_t)kept_before(kc, se);
        arrive = (int64_t)(se - sb) - kept;
        const CutHead h0 = cv.head[lane], h1 = cv.head[lane + 1];
        over = h0.over | h1.over;
        // outputs this rank's sources own: its regular interval + its share of the cut strata at its two boundaries
        sourced = h1.Z > h0.A ? (int64_t)(h1.Z - h0.A) : 0;
        if (kc.en[0] > kc.st[0]) sourced += (int64_t)cv.srccnt[(size_t)lane * kCutSlots + lane];
        if (kc.en[1] > kc.st[1]) sourced += (int64_t)cv.srccnt[(size_t)(lane + 1) * kCutSlots + lane];
        if (lane != rank) {
            // this rank's regular outputs that live in rank `lane`'s shard
            const uint32_t A = cv.head[rank].A, Zn = cv.head[rank + 1].Z;
            const uint32_t lo = min(max(A, sb), se), hi = min(max(Zn, sb), se);
            pl.send_lo = (int64_t)lo; pl.send_cnt = hi > lo ? (int64_t)(hi - lo) : 0;
        }
    }
    if (g.remote && lane < world) {
        const int64_t fill = t == 0 ? 0 : g.annex_all[(int64_t)t * kWorldSlots + lane];
        const int64_t room = g.rem->rs[lane] - g.rem->ld[lane];
        pl.fill_next = fill + (fill + arrive > room ? 0 : arrive);
        pl.dst_col = fill;                                               // (the step's first column on that rank: col(s) is added per output)
    }
    const int64_t my_arrive = read_lane((double)arrive, rank) , my_send = read_lane((double)(sourced - kept), rank);     // (below 2^32: exact)
    pw.n_recv = my_arrive; pw.n_send = my_send > 0 ? my_send : 0;
    pw.l0 = 0; pw.l1 = 0;
    pw.flags = __ballot(over != 0) ? 8 : 0;
}

template <class Model>
struct PackStrataArgs {
    const typename Model::store_t* values; int64_t rs, n; int nb; int t;
    int world, rank; uint64_t pid0; double n_pop;
    ExchangeGeom geom; int64_t* annex_base; ExchangePlan* plan_out;
    CutView cut; const uint32_t* offs; int k; uint64_t seed, draw2;
    Hier h; const double* totals_d; double e0, e1, e2;                   // prefix-count form
    FHier f; const uint64_t* totals_u; double ess_frac; const uint32_t* q_prev;      // fixed-point form
    const uint32_t* trace_cur; int trace_par;
};

// The packing launch of a multinomial exchange (remote lineages): every workgroup derives the plan on its first wavefront (workgroup 0
// stores it), then the work items -- per destination rank d: this rank's regular outputs inside d's shard, and its entries of the cut
// strata at its two boundaries inside d's shard -- are walked tile by tile (tiles of the POPULATION's output index): thresholds, the
// strata search among this rank's sources (step_fixed.hpp / step_counts.hpp), and each migrant stored into column col(s) of d's annex.
template <class Model, int MODE>
__global__ __launch_bounds__(kThreads) void exchange_pack_strata_kernel(PackStrataArgs<Model> a)
{
    using S = typename Model::store_t;
    constexpr bool COUNTS = MODE == kPackCounts;
    static_assert(MODE == kPackCounts || MODE == kPackFixed, "the integer forms");
    __shared__ CountsLdsT<COUNTS ? kFixMultinomial : kFixSystematic> Lc;
    __shared__ FixedLdsT<COUNTS ? kFixSystematic : kFixMultinomial> Lx;
    __shared__ int64_t s_send_lo[kWorldSlots], s_send_cnt[kWorldSlots], s_fill[kWorldSlots];
    __shared__ int64_t s_nsend;
    __shared__ int s_resample;
    __shared__ uint32_t s_sb[kWorldSlots + 1];                           // the shards' first outputs (read once: the work items below are many and mostly empty)
    __shared__ __attribute__((aligned(16))) FixedFound s_ff;
    __shared__ __attribute__((aligned(16))) StepFound s_cf;
    const int tid = threadIdx.x, lane = lane_id();
    const int world = a.world, rank = a.rank;
    if (tid <= world) s_sb[tid] = (uint32_t)a.geom.shard_begin[tid];
    // ---- this rank's place in the population, the decision ----
    uint64_t S_tot = 0, before = 0, own = 0;
    double W = 0.0, c_lo = 0.0, c_hi = 0.0;
    TableCdf tc;
    bool resample = true;
    if constexpr (COUNTS) {
        double r0 = 0.0, r1 = 0.0, rv = 0.0;
        if (lane < world) { r0 = a.totals_d[3 * lane]; r1 = a.totals_d[3 * lane + 1]; rv = a.totals_d[3 * lane + 2]; }
        const bool bef = lane < rank, upto = lane <= rank;
        tc.e0 = a.e0; tc.e1 = a.e1; tc.e2 = a.e2; tc.u0 = 0.0; tc.n_pop = a.n_pop; tc.inv = 1.0; tc.seed = a.seed; tc.draw = 0; tc.uid0 = 0;
        tc.base0 = wave_sum(bef ? r0 : 0.0); tc.base1 = wave_sum(bef ? r1 : 0.0); tc.basev = wave_sum(bef ? rv : 0.0);
        W = tc.cdf(wave_sum(r0), wave_sum(r1), a.n_pop);
        c_lo = tc.cdf(tc.base0, tc.base1, tc.basev);
        c_hi = tc.cdf(wave_sum(upto ? r0 : 0.0), wave_sum(upto ? r1 : 0.0), wave_sum(upto ? rv : 0.0));
    } else {
        uint64_t s = 0, q = 0;
        if (lane < world) { s = a.totals_u[3 * lane]; q = a.totals_u[3 * lane + 1]; }
        S_tot = wave_sum_u64(s); before = wave_sum_u64(lane < rank ? s : 0ull); own = wave_sum_u64(lane == rank ? s : 0ull);
        resample = fixed_decide(S_tot, wave_sum_u64(q), a.n_pop, a.ess_frac, true).resample;
    }
    if (wave_id() == 0) {
        PlanLane pl; PlanWave pw;
        plan_wave_strata(a.geom, a.t, a.cut, resample, pl, pw);
        s_send_lo[tid] = pl.send_lo; s_send_cnt[tid] = pl.send_cnt; s_fill[tid] = pl.dst_col;
        if (tid == 0) { s_nsend = pw.n_send; s_resample = resample ? 1 : 0; }
        if (blockIdx.x == 0) plan_store(a.geom, a.t, pl, pw, a.annex_base, a.plan_out);
    }
    __syncthreads();
    if (!s_resample || s_nsend == 0) return;                            // workgroup-uniform
    const bool last_shard = rank + 1 == world;
    const int64_t nb_pop = ((int64_t)a.n_pop + kTile - 1) / kTile;
    const RemoteStores* __restrict__ rem = a.geom.rem;
    const CutHead hme0 = a.cut.head[rank], hme1 = a.cut.head[rank + 1];
    const bool same = hme0.A > hme0.Z && hme0.A == hme1.A && hme0.Z == hme1.Z;
    // work items: (kind 0: regular, 1: the cut stratum of boundary `rank`, 2: of boundary `rank + 1`) x destination rank
    uint32_t unit = 0;
    for (int item = 0; item < 3 * world; ++item) {
        const int kind = item / world, d = item - kind * world;
        if (d == rank) continue;
        const uint32_t sb = s_sb[d], se = s_sb[d + 1];
        uint32_t lo, hi;
        const uint32_t* tab = nullptr; uint32_t tab0 = 0;
        if (kind == 0) { lo = (uint32_t)s_send_lo[d]; hi = lo + (uint32_t)s_send_cnt[d]; }
        else {
            const CutHead hb = kind == 1 ? hme0 : hme1;
            if ((kind == 2 && same) || hb.A <= hb.Z) continue;
            lo = min(max(hb.Z, sb), se); hi = min(max(hb.A, sb), se);
            tab = a.cut.tab + (size_t)(kind == 1 ? rank : rank + 1) * kCutRow; tab0 = hb.Z;
        }
        if (hi <= lo) continue;                                          // uniform
        // (the work units -- the items' tiles, counted through -- go round the workgroups: a workgroup that has none of this item's
        //  tiles reads nothing of it; with fewer units than workgroups nobody walks two)
        const uint32_t tile0 = lo / kTile, n_tiles = (hi - 1) / kTile - tile0 + 1;
        const uint32_t bx = (blockIdx.x + gridDim.x - unit % gridDim.x) % gridDim.x;
        unit += n_tiles;
        if (bx >= n_tiles) continue;
        const KeptCtx kc = kept_ctx(a.cut, d, sb);
        S* const annex_row = static_cast<S*>(const_cast<void*>(rem->values[d])) + (int64_t)a.t * rem->rs[d] + rem->ld[d];
        int64_t* const origin = const_cast<int64_t*>(rem->origin[d]);
        uint32_t* const annex_trace = a.trace_cur ? rem->trace[a.trace_par][d] + rem->ld[d] : nullptr;
        const int64_t col0 = s_fill[d], room = rem->rs[d] - rem->ld[d];
        for (uint32_t tile = tile0 + bx; tile < tile0 + n_tiles; tile += gridDim.x) {
            const uint32_t g0 = tile * kTile;
            const uint32_t s_first = max(lo, g0), s_last = min(hi, g0 + kTile) - 1;
            const uint64_t gj = (uint64_t)g0 + (uint64_t)tid * kPPT;
            int32_t anc[kPPT];
            bool mine[kPPT];
#pragma unroll
            for (int i = 0; i < kPPT; ++i) { anc[i] = 0; mine[i] = false; }
            if constexpr (COUNTS) {
                if (wave_id() == 0) {
                    const LocatedStrata ls = counts_strata_locate(a.h, tc, a.offs, a.k, a.n, a.nb, strata_near(a.k, nb_pop, (int64_t)tile), 0, s_first, s_last, W, c_lo, c_hi, last_shard, nullptr);
                    if (tid == 0) { s_cf.loc = ls.loc; s_cf.w0 = ls.w0; s_cf.w1 = ls.w1; }
                }
                __syncthreads();
                LocatedStrata ls;
                ls.loc = s_cf.loc; ls.w0 = s_cf.w0; ls.w1 = s_cf.w1;
                counts_strata_walk<S>(tc, a.offs, a.k, a.values + (int64_t)a.t * a.rs, a.n, a.nb, ls, W, (int64_t)gj, a.seed, a.draw2, gj, anc, Lc, c_lo, c_hi, last_shard, mine);
            } else {
                if (wave_id() == 0) {
                    const StrataLocated sl = strata_locate(a.f, a.offs, a.k, a.nb, strata_near(a.k, nb_pop, (int64_t)tile), 0, s_first, s_last, S_tot, before, own, nullptr);
                    if (tid == 0) { s_ff.loc = sl.loc; s_ff.w0 = sl.w0; s_ff.w1 = sl.w1; }
                }
                __syncthreads();
                StrataLocated sl;
                sl.loc = s_ff.loc; sl.w0 = s_ff.w0; sl.w1 = s_ff.w1;
                strata_walk(a.offs, a.k, a.q_prev, a.nb, sl, S_tot, (int64_t)gj, a.seed, a.draw2, gj, anc, Lx, before, own, mine);
            }
            const S* __restrict__ vrow = a.values + (int64_t)a.t * a.rs;
#pragma unroll
            for (int i = 0; i < kPPT; ++i) {
                const uint32_t s = (uint32_t)gj + (uint32_t)i;
                bool on = s >= lo && s < hi && mine[i];
                if (on && tab) on = (tab[min(s - tab0, (uint32_t)kCutCap)] >> 24) == (uint32_t)rank;
                if (!on) continue;
                const int64_t col = col0 + (int64_t)(s - sb) - (int64_t)kept_before(kc, s);
                if (col < room) {                                       // (an annex too small is flagged by its owner and the run repeated)
                    const int32_t idx = anc[i];
                    __builtin_nontemporal_store(vrow[idx], annex_row + col);
                    __builtin_nontemporal_store(((int64_t)rank << 32) | (int64_t)idx, origin + col);
                    if (annex_trace) annex_trace[col] = a.trace_cur[idx];
                }
            }
            __syncthreads();                                            // LDS is reused by the next tile
        }
    }
}

// Skip rows.  A migrating particle takes its lineage x_0 .. x_t along, and extracting a lineage is a chain of t dependent gathers
// (~1.7 us a hop: each is a row further away in memory) that no parallelism across particles shortens -- half of a long-trace run
// in the exchange scope (profiles/r02_notes.md).  Row m of `skip` (written after step 8m, one launch every eighth step) holds, for
// every local slot of generation 8m, the slot of its lineage at generation 8(m-1); annex columns get the identity when they are
// committed.  A lineage is then extracted in blocks of eight generations by different workgroups (gridDim.y of the packing launch):
// each reaches its block by <= 7 single hops, <= t/8 skip hops and one more, and walks eight generations -- chains of ~t/8 + 16.
constexpr int kSkipEvery = 8;
__global__ __launch_bounds__(kThreads) void skip_rows_kernel(const int32_t* __restrict__ anc, int64_t rs, int64_t n, const int32_t* __restrict__ resampled,
                                                             int t, int32_t* __restrict__ skip_row)
{
    const int64_t j = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (j >= n) return;
    bool hop[kSkipEvery];
#pragma unroll
    for (int d = 0; d < kSkipEvery; ++d) hop[d] = resampled[t - 1 - d] != 0;       // (off the chain: all eight flags first)
    int32_t idx = (int32_t)j;
#pragma unroll
    for (int d = 0; d < kSkipEvery; ++d) { if (hop[d]) idx = anc[(int64_t)(t - d) * rs + idx]; }
    skip_row[j] = idx;
}

// Lineage of the generation-t particle in slot `idx`: x_0 .. x_t into rec[0 .. t].
template <class S, class R>
__device__ __forceinline__ void extract_lineage(const S* __restrict__ values, const int32_t* __restrict__ anc, int64_t rs, const int32_t* __restrict__ resampled,
                                                int t, int32_t idx, R* __restrict__ rec)
{
    for (int tt = t; tt >= 0; --tt) {
        rec[tt] = static_cast<R>(values[(int64_t)tt * rs + idx]);
        if (tt > 0 && resampled[tt - 1]) idx = anc[(int64_t)tt * rs + idx];
    }
}

template <class Model, class R>
struct PackArgs {
    const typename Model::store_t* values; const int32_t* anc; int64_t rs, n; int nb; const int32_t* resampled; int t;   // generation t is the one resampled
    const int32_t* skip;                                         // skip rows (see skip_rows_kernel) or nullptr: gridDim.y = t / 8 + 1 blocks of generations then
    const ExchangePlan* plan; int world, rank;
    R* send;                                                     // records of t + 1 values: this rank's own send buffer ...
    // ... or DIRECT stores into the receivers' buffers (peer access / IPC mappings; loopback: plain device pointers): peer_recv[r] =
    // rank r's receive buffer as THIS device addresses it, peer_slot[r] = the slot rank r keeps for this rank, cap = records per slot
    void* const* peer_recv; const int32_t* peer_slot; int64_t cap;
    // PLAN_INSIDE: every workgroup derives the plan itself (one wavefront, from the all-gathered totals); workgroup (0, 0) also stores it
    ExchangeGeom geom; int64_t* annex_base; ExchangePlan* plan_out;
    // prefix-count form
    Hier h; PlanCountsIn pc;
    // fixed-point form
    FHier f; PlanFixedIn pf; const uint32_t* q_prev;
    // floating-point form
    const double* wrel; const double* bc; const double* bf; const StepCtrl* ctrl; uint64_t seed;
    uint64_t pid0;
    const uint32_t* trace_cur; int trace_par;                    // remote lineages with trace words: this rank's words of generation t, and t's parity
};


template <class Model, class R, int MODE, bool PLAN_INSIDE = false>
__global__ __launch_bounds__(kThreads) void exchange_pack_kernel(PackArgs<Model, R> a)
{
    using S = typename Model::store_t;
    constexpr bool COUNTS = MODE == kPackCounts, FIXED = MODE == kPackFixed;
    __shared__ CountsLdsT<kFixStratified> Lc;                    // (the stratified forms' layouts: the systematic ones' plus the outputs' uniforms)
    __shared__ AncestorLds Lf;
    __shared__ FixedLdsT<kFixStratified> Lx;
    __shared__ uint32_t s_hop[32];                              // bit tt - 1 of word (tt - 1) / 32: did step tt - 1 resample? (<= 1024 steps)
    __shared__ int64_t s_send_lo[kWorldSlots], s_send_cnt[kWorldSlots], s_send_base[kWorldSlots], s_dst_col[kWorldSlots];
    __shared__ int64_t s_nsend;
    const int tid = threadIdx.x;
    static_assert(!PLAN_INSIDE || COUNTS || FIXED, "the plan is a pure function of the all-gathered totals in the integer forms only");
    if constexpr (PLAN_INSIDE) {
        if (wave_id() == 0) {
            PlanLane pl; PlanWave pw;
            bool rs_ = true;
            double o;
            if constexpr (FIXED) o = plan_bounds_fixed(a.pf, a.world, rs_);
            else o = plan_bounds_counts(a.pc, a.world);
            plan_wave(a.geom, a.t, o, rs_, pl, pw);
            s_send_lo[tid] = pl.send_lo; s_send_cnt[tid] = pl.send_cnt; s_send_base[tid] = pl.send_base; s_dst_col[tid] = pl.dst_col;
            if (tid == 0) s_nsend = pw.n_send;
            if (blockIdx.x == 0 && blockIdx.y == 0) plan_store(a.geom, a.t, pl, pw, a.annex_base, a.plan_out);
        }
        __syncthreads();
        if (s_nsend == 0) return;                               // workgroup-uniform
    } else {
        if (!a.plan->resample || a.plan->n_send == 0) return;   // workgroup-uniform
    }
    // (the walk below is a chain of dependent gathers: nothing else may sit on it -- the per-step flags come out of LDS, not memory;
    //  a migrant that travels as its current state alone walks nothing)
    if (!(a.geom.no_history != 0 || a.geom.remote != 0)) {
        if (tid < 32) {
            uint32_t w = 0;
            for (int b = 0; b < 32; ++b) { const int s2 = tid * 32 + b; if (s2 < a.t && a.resampled[s2] != 0) w |= 1u << b; }
            s_hop[tid] = w;
        }
        __syncthreads();
    }
    TableCdf tc;
    FixedCdf fc;
    bool last_shard = a.rank + 1 == a.world;
    if constexpr (FIXED) {
        const FixedRanks rk = fixed_ranks(a.pf.all_totals, a.world, a.rank);
        const FixedDecision d = fixed_decide(rk.S, rk.Q, a.pf.n_pop, a.pf.ess_frac, true);
        fc.inv = d.inv; fc.u0 = a.pf.u0; fc.n_pop = a.pf.n_pop; fc.base = rk.before;
        fc.seed = a.pf.seed; fc.draw = a.pf.draw; fc.uid0 = 0;
    }
    if constexpr (COUNTS) {
        const int lane = lane_id();
        double r0 = 0.0, r1 = 0.0, rv = 0.0;
        if (lane < a.world) { r0 = a.pc.all_totals[3 * lane]; r1 = a.pc.all_totals[3 * lane + 1]; rv = a.pc.all_totals[3 * lane + 2]; }
        const bool before = lane < a.rank;
        tc.e0 = a.pc.e0; tc.e1 = a.pc.e1; tc.e2 = a.pc.e2; tc.u0 = a.pc.u0; tc.n_pop = a.pc.n_pop;
        tc.seed = a.pc.seed; tc.draw = a.pc.draw; tc.uid0 = 0;
        tc.base0 = wave_sum(before ? r0 : 0.0); tc.base1 = wave_sum(before ? r1 : 0.0); tc.basev = wave_sum(before ? rv : 0.0);
        const double t0 = wave_sum(r0), t1 = wave_sum(r1);
        tc.inv = 1.0;
        const double W = tc.cdf(t0, t1, a.pc.n_pop);
        tc.inv = a.pc.n_pop / W;
    }
    const bool no_history = a.geom.no_history != 0, remote = a.geom.remote != 0;
    const int len = (no_history || remote) ? 1 : a.t + 1;
    const int row_t = no_history ? (a.t & 1) : a.t;              // where generation t's values sit
    for (int r = 0; r < a.world; ++r) {
        const int64_t cnt = PLAN_INSIDE ? s_send_cnt[r] : a.plan->send_cnt[r];
        if (r == a.rank || cnt == 0) continue;                  // uniform
        const int64_t lo = PLAN_INSIDE ? s_send_lo[r] : a.plan->send_lo[r];
        // where rank r's records go: its own receive slot for this rank (direct stores), or this rank's send buffer; remote
        // lineages: row t of rank r's annex columns and its origin table, from the column the plan worked out
        R* const dst = remote ? nullptr
                     : a.peer_recv ? static_cast<R*>(a.peer_recv[r]) + (int64_t)a.peer_slot[r] * a.cap * (int64_t)len
                                   : a.send + (PLAN_INSIDE ? s_send_base[r] : a.plan->send_base[r]) * (int64_t)len;
        S* annex_row = nullptr; int64_t* origin = nullptr; int64_t col0 = 0, room = 0;
        uint32_t* annex_trace = nullptr;
        if (remote) {
            const RemoteStores* __restrict__ rem = a.geom.rem;
            annex_row = static_cast<S*>(const_cast<void*>(rem->values[r])) + (int64_t)a.t * rem->rs[r] + rem->ld[r];
            origin = const_cast<int64_t*>(rem->origin[r]);
            if (a.trace_cur) annex_trace = rem->trace[a.trace_par][r] + rem->ld[r];
            col0 = PLAN_INSIDE ? s_dst_col[r] : a.plan->dst_col[r]; room = rem->rs[r] - rem->ld[r];
        }
        for (int64_t tl = blockIdx.x; tl * kTile < cnt; tl += gridDim.x) {
            const int64_t rem = cnt - tl * kTile;
            const int n_out = rem < kTile ? (int)rem : kTile;
            const uint64_t gj0 = (uint64_t)(lo + tl * kTile);
            int32_t anc[kPPT];
            if constexpr (FIXED) {
                {
                    int32_t neg[kPPT];
                    lane_fill(neg, (int32_t)-1);
                    store4(Lx.slot, (int64_t)tid * kPPT, neg);
                }
                __syncthreads();
                if (a.pf.rs == kFixStratified) ancestors_fixed<kFixStratified>(a.f, fc, a.q_prev, a.n, a.nb, last_shard, (double)gj0, n_out, r < a.rank ? 0 : a.nb - 1, anc, Lx);
                else ancestors_fixed<kFixSystematic>(a.f, fc, a.q_prev, a.n, a.nb, last_shard, (double)gj0, n_out, r < a.rank ? 0 : a.nb - 1, anc, reinterpret_cast<FixedLds&>(Lx));
            } else if constexpr (COUNTS) {
                {
                    int32_t neg[kPPT];
                    lane_fill(neg, (int32_t)-1);
                    store4(Lc.slot, (int64_t)tid * kPPT, neg);
                }
                __syncthreads();
                if (a.pc.rs == kFixStratified) ancestors_counts<S, true, kFixStratified>(a.h, tc, a.values + (int64_t)row_t * a.rs, a.n, a.nb, last_shard, (double)gj0, n_out, r < a.rank ? 0 : a.nb - 1, anc, Lc);
                else ancestors_counts<S, true, kFixSystematic>(a.h, tc, a.values + (int64_t)row_t * a.rs, a.n, a.nb, last_shard, (double)gj0, n_out, r < a.rank ? 0 : a.nb - 1, anc, reinterpret_cast<CountsLds&>(Lc));
            } else {
                AncestorIn in;
                in.wrel = a.wrel; in.bc = a.bc; in.bf = a.bf; in.nb = a.nb; in.n_in = a.n;
                in.W = a.ctrl->W; in.scale = a.ctrl->scale; in.cdf_lo = a.ctrl->cdf_lo;
                in.u0 = a.ctrl->u0; in.inv_stepw = a.ctrl->inv_global; in.g_end = a.ctrl->g_end;
                in.seed = a.seed; in.step = (uint64_t)a.t + 1; in.gj_tile0 = gj0; in.n_total_out = (uint64_t)a.pc.n_pop;
                in.n_valid_tile = n_out; in.id0 = 0; in.bc_in_lds = 0; in.guess = -1;
                ancestors_systematic(in, anc, Lf, WrelSource{a.wrel});
            }
            {
                // the lane's four lineages side by side: each is a chain of dependent gathers (~1.7 us apiece: every hop is a row
                // further away in memory), so nothing but the chain's links may sit on it
                int32_t idx[kPPT]; R* rec[kPPT]; bool on[kPPT];
#pragma unroll
                for (int k = 0; k < kPPT; ++k) {
                    const int q = tid * kPPT + k;
                    on[k] = q < n_out; idx[k] = max(anc[k], 0); rec[k] = remote ? nullptr : dst + (tl * kTile + (on[k] ? q : 0)) * len;
                }
                auto hops = [&](int tt) { return tt > 0 && ((s_hop[(tt - 1) >> 5] >> ((tt - 1) & 31)) & 1u) != 0; };
                auto single = [&](int tt) {                                 // generation tt -> tt - 1
                    if (!hops(tt)) return;
                    const int32_t* __restrict__ arow = a.anc + (int64_t)tt * a.rs;
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) { if (on[k]) idx[k] = arow[idx[k]]; }
                };
                if (no_history || remote) {
                    // the migrant is its current state (and, with remote lineages, the slot its history stays in)
                    const typename Model::store_t* __restrict__ vrow = a.values + (int64_t)row_t * a.rs;
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) {
                        if (!on[k]) continue;
                        if (remote) {
                            const int64_t col = col0 + tl * kTile + tid * kPPT + k;
                            if (col < room) {                    // (an annex too small is flagged by its owner and the run repeated)
                                __builtin_nontemporal_store(vrow[idx[k]], annex_row + col);
                                __builtin_nontemporal_store(((int64_t)a.rank << 32) | (int64_t)idx[k], origin + col);
                                if (annex_trace) annex_trace[col] = a.trace_cur[idx[k]];          // the migrant's whole trace so far
                            }
                        } else __builtin_nontemporal_store(static_cast<R>(vrow[idx[k]]), rec[k]);
                    }
                    __syncthreads();
                    continue;
                }
                // this workgroup's block of generations [lo, hi] (the whole lineage without skip rows)
                int hi = a.t, lo = 0;
                if (a.skip) {
                    lo = (int)blockIdx.y * kSkipEvery; hi = lo + kSkipEvery - 1 < a.t ? lo + kSkipEvery - 1 : a.t;
                    int g = a.t;
                    while (g > hi && (g % kSkipEvery) != 0) { single(g); --g; }
                    while (g - kSkipEvery >= hi) {
                        const int32_t* __restrict__ srow = a.skip + (int64_t)(g / kSkipEvery) * a.rs;
#pragma unroll
                        for (int k = 0; k < kPPT; ++k) { if (on[k]) idx[k] = srow[idx[k]]; }
                        g -= kSkipEvery;
                    }
                    while (g > hi) { single(g); --g; }
                }
                for (int tt = hi; tt >= lo; --tt) {
                    const bool hop = tt > lo && hops(tt);
                    const typename Model::store_t* __restrict__ vrow = a.values + (int64_t)tt * a.rs;
                    const int32_t* __restrict__ arow = a.anc + (int64_t)tt * a.rs;
                    int32_t nxt[kPPT];
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) nxt[k] = (on[k] && hop) ? arow[idx[k]] : idx[k];      // the chain's link first
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) { if (on[k]) __builtin_nontemporal_store(static_cast<R>(vrow[idx[k]]), rec[k] + tt); }
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) idx[k] = nxt[k];
                }
            }
            __syncthreads();                                    // LDS is reused by the next tile
        }
    }
}

// Received records -> annex columns annex_base[t] + recv_off[r] + k of rows 0 .. t, identity ancestors.
template <class S, class R>
__global__ __launch_bounds__(kThreads) void exchange_commit_kernel(const ExchangePlan* __restrict__ plan, int world, const R* __restrict__ recv, int t,
                                                                    const int64_t* __restrict__ annex_base, S* __restrict__ values, int32_t* __restrict__ anc,
                                                                    int64_t rs, int64_t ld, int32_t* __restrict__ skip)
{
    if (!plan->resample || plan->n_recv == 0) return;
    if (!anc) {
        // filtering-only shards: records of one value into the row of generation t; the annex starts over every step
        for (int r = 0; r < world; ++r) {
            const int64_t cnt = plan->recv_cnt[r];
            if (cnt == 0) continue;
            const int64_t base = plan->recv_base[r], off = plan->recv_off[r];
            for (int64_t k = (int64_t)blockIdx.x * kThreads + threadIdx.x; k < cnt; k += (int64_t)gridDim.x * kThreads)
                values[(int64_t)(t & 1) * rs + ld + off + k] = static_cast<S>(recv[base + k]);
        }
        return;
    }
    const int len = t + 1;
    const int64_t col0 = ld + annex_base[t];
    for (int r = 0; r < world; ++r) {
        const int64_t cnt = plan->recv_cnt[r];
        if (cnt == 0) continue;
        const int64_t base = plan->recv_base[r], off = plan->recv_off[r];
        for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < cnt * len; i += (int64_t)gridDim.x * kThreads) {
            const int64_t k = i / len; const int tt = (int)(i - k * len);
            const int64_t col = col0 + off + k;
            values[(int64_t)tt * rs + col] = static_cast<S>(recv[(base + k) * len + tt]);
            anc[(int64_t)tt * rs + col] = (int32_t)col;
            if (skip && tt >= kSkipEvery && (tt % kSkipEvery) == 0) skip[(int64_t)(tt / kSkipEvery) * rs + col] = (int32_t)col;
        }
    }
}

}  // namespace cph
