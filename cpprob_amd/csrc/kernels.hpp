// Particle kernels of the SIS / SMC engine (gfx950, wave64, fp64).
//
// Data layout in HBM (one context = one shard of the population):
//   values   [T][rs]   value of predict hit t in slot i of generation t, stored as Model::store_t (fp64; one byte for the
//                      HMM's states); row t-1 doubles as the state the step-t kernel reads (no separate state).
//                      rs = ld + the immigrant annex of the exchange scope (0 otherwise)
//   anc      [T][rs]   int32 slot of generation t-1 extended by slot i of generation t (stored write-through)
//   logw     [2][ld]   fp64 log-weights, ping-pong between steps
//   wrel     [2][ld]   fp64 exp(logw - tile max): the linear weights the resampler scans, written
//                      by the kernel that produced the weights (it has them in registers for the
//                      tile partial anyway), so no exp() is ever recomputed downstream.  Table-weight models on an
//                      every-step schedule skip it between steps: state + emission table give the weight (WeightSource)
//   part     [nb]      per-tile {max, sum exp(lw-max), sum exp(2(lw-max))} from the same kernel -- or, in that same
//                      table-weight mode, one word of packed per-value counts per tile
//   bc       [nb+1]    exclusive prefix of the tile sums rescaled to the global max: the
//   bf       [nb]      resampling CDF at tile granularity, and the per-tile rescale factors
//   ctrl               device-resident control block: max, W, Q, ESS, log Z, resample decision,
//                      the systematic offset u0 of the next resampling step
// ld = N rounded up to the tile (1024): every lane's 4 consecutive particles are one aligned
// 32-B (fp64) or 16-B (int32) access and no kernel has a ragged tail; padding slots carry
// logw = -inf / wrel = 0, so they never become ancestors and add nothing to any sum.
// One particle per lane-slot, structure of arrays, every access of a step coalesced; the only
// gather (ancestor state) reads sorted indices.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include "cpprob/detail/dist.hpp"
#include "models.hpp"
#include "cpprob/detail/rng.hpp"
#include "cpprob/detail/wave.hpp"

namespace cph {

// Tile partials {max, sum e, sum e^2} are kept as three arrays of nb doubles (structure of arrays:
// the consumers read them coalesced): part[0..nb) = max, part[nb..2nb) = sum, part[2nb..3nb) = sum of squares.
using Partial = double;
// (stride = number of tiles rounded up to kPartPad so that consumers may read whole vectors past the last tile)
constexpr int kPartPad = 2048;
__host__ __device__ inline int part_stride(int nb) { return (nb + kPartPad - 1) / kPartPad * kPartPad; }
__device__ __forceinline__ void put_partial(Partial* __restrict__ part, int nb, int b, double m, double s, double q)
{
    const int st = part_stride(nb);
    part[b] = m; part[st + b] = s; part[2 * st + b] = q;
}

struct StepCtrl {
    double M;         // max log-weight of the current generation (global over shards)
    double W;         // sum exp(logw - M)
    double Q;         // sum exp(2 (logw - M))
    double ess;       // W^2 / Q
    double log_z;     // accumulated log evidence
    double cdf_lo;    // sharded runs: global CDF offset of this shard (0 on one GPU)
    double w_local;   // this shard's sum rescaled to M
    double scale;     // exp(M_local - M): rescales bc[] / bf[] (built against the local max) to M
    double u0;        // systematic offset in [0,1) of the resampling that precedes the NEXT step
    double u0_pp[2];  // fused step kernels: offset of the resampling before step t lives in u0_pp[t & 1] (written by
                      // workgroup 0 of step t-1 while step t-1's other workgroups may still read u0_pp[(t-1) & 1])
    double inv_stepw; // n_local / (local weight sum, local units): CDF positions -> local output indices
    double lw_after;  // log-weight every particle of this shard carries right after resampling: log of the
                      // shard's mean weight over the population's mean weight (0 on a single shard)
    double inv_global;// n_pop / W: global CDF positions -> global output indices (exchange scope)
    double g_end;     // first output index owned by the NEXT shard's sources (+inf on the last shard and on a single shard)
    int32_t do_resample;  // decision taken after the last weighted step
    int32_t n_resampled;
    double ref_cur;       // fixed-point form (step_fixed.hpp): the reference R_t of the generation just produced
    double fix_gap;       // ... and the largest R_t - max_i lw_i of the run: how far below its reference the heaviest particle of
                          // some generation sat (each 0.69 of it costs the integer weights one of their 32 bits)
    double* lz_trace;     // [T] (or nullptr): the evidence accumulated BEFORE generation t's books were kept -- what a repair of that generation rewinds to
    int32_t first_bad;    // the first generation whose gap exceeded kFixGapLimit (-1: none): where the host's repair starts (cpprob_hip.hip: settle_fixed)
    int32_t pad_;
};

enum { RS_SYSTEMATIC = 0, RS_STRATIFIED = 1, RS_PRECOMPUTED = 2 };

// (4-wide accesses load4 / store4: cpprob/detail/wave.hpp)

// Rows of the particle store may be narrower than the type the model computes in (Model::store_t vs value_t: the HMM's
// states 0..2 travel as one byte): converting forms of the 4-wide accesses.
template <class S, class V>
__device__ __forceinline__ void load4_as(const S* __restrict__ p, int64_t i, V (&v)[kPPT])
{
    S raw[kPPT];
    load4(p, i, raw);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) v[k] = static_cast<V>(raw[k]);
}
template <class S, class V>
__device__ __forceinline__ void store4_as(S* __restrict__ p, int64_t i, const V (&v)[kPPT])
{
    S raw[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) raw[k] = static_cast<S>(v[k]);
    store4(p, i, raw);
}

// Write-through store (sc1) for rows no step kernel reads again -- the ancestor rows, which only the read-out walks: they
// reach memory while the kernel still runs instead of at its boundary, where dirty bytes cost time (-2 % per run at 10^6;
// the same policy on re-read rows costs up to +33 %: profiles/r01_ab_notes.md).
__device__ __forceinline__ void store4_write_through(int32_t* __restrict__ p, int64_t i, const int32_t (&v)[kPPT])
{
    using I4 = int __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int q = 0; q < kPPT; q += 4) {
        I4 w;
        w[0] = v[q]; w[1] = v[q + 1]; w[2] = v[q + 2]; w[3] = v[q + 3];
        int32_t* addr = p + i + q;
        // (s_nop 1 inside the string: hipcc does not pad an asm statement's hazards, and its next instruction may otherwise
        //  overwrite the data registers before a 16-byte store has read them -- lanes 12..15 of every row lost anc[0] that way)
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(addr), "v"(w) : "memory");
    }
}

// ---------------------------------------------------------------------------------------------
// Tile partial of the weights this workgroup just produced: {max, sum e, sum e^2}; the linear
// weights e = exp(lw - max) are returned for the wrel store.  lw of padding slots must be -inf.
// ---------------------------------------------------------------------------------------------
// Grid references: continuous-weight models put a tile's reference on the grid {k ln 2} (the smallest grid point >= the tile max),
// so that every later rescaling between two references is an exact power of two -- ldexp instead of exp in the step kernels'
// all-to-all (8 per lane at 1.25e6 particles).  The linear weights stay in (0.5, 1] at the tile max.
constexpr double kLn2 = 0.693147180559945309417232121458, kInvLn2 = 1.44269504088896340735992468100;
__device__ __forceinline__ double grid_reference(double m) { return ceil(m * kInvLn2) * kLn2; }          // -inf stays -inf
// factor that takes sums relative to reference mc to reference M >= mc
__device__ __forceinline__ double rescale_factor(double mc, double M, bool grid)
{
    if (mc == M) return 1.0;
    if (mc == -INFINITY) return 0.0;
    return grid ? ldexp(1.0, (int)rint((mc - M) * kInvLn2)) : exp(mc - M);
}

__device__ __forceinline__ void tile_partial(const double (&lw)[kPPT], double (&e)[kPPT], Partial* __restrict__ part,
                                             double* s_scr /* >= 3*kWaves doubles, unused by any in-flight combine */, int tile, bool grid = false)
{
    double m = lw[0];
#pragma unroll
    for (int k = 1; k < kPPT; ++k) m = fmax(m, lw[k]);
    m = block_max(m, s_scr);
    if (grid) m = grid_reference(m);
    double s = 0.0, q = 0.0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        e[k] = exp_nonpos(fmax(lw[k] - m, -1000.0));            // lw <= m; -inf (padding) and the empty tile's NaN clamp to an exact 0
        s += e[k]; q += e[k] * e[k];
    }
    block_sum2(s, q, s_scr + kWaves);
    if (threadIdx.x == 0) put_partial(part, (int)gridDim.x, tile, m, s, q);
}

// Standalone: partials + linear weights of an arbitrary log-weight array of n entries
// (building block / tests).  wrel has room for gridDim.x * kTile entries.
__global__ __launch_bounds__(kThreads) void weights_partials_kernel(const double* __restrict__ logw, int64_t n, Partial* __restrict__ part,
                                                                     double* __restrict__ wrel)
{
    __shared__ double s_scr[3 * kWaves];
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    double lw[kPPT], e[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) lw[k] = (j0 + k < n) ? logw[j0 + k] : -INFINITY;
    tile_partial(lw, e, part, s_scr, (int)blockIdx.x);
    store4(wrel, j0, e);
}

// ---------------------------------------------------------------------------------------------
// scan_partials: one workgroup of 1024 threads turns the tile partials of the generation
// just weighted into: global max M, normaliser W, ESS, the tile-level CDF bc[] and rescale
// factors bf[], the evidence increment, the resampling decision (ESS < ess_frac * N; thesis
// p.37) and the systematic offset of the resampling that may follow.  Launched once per step;
// everything stays on the device, so the host never waits inside a run.
// ---------------------------------------------------------------------------------------------
constexpr int kScanThreads = 1024;

struct ScanArgs {
    const Partial* part; int nb;
    double* bc; double* bf; StepCtrl* ctrl;
    int t, T;
    double n_pop, ess_frac;    // joint population size (ESS test, evidence)
    double n_local;            // particles of this shard
    uint64_t seed;
    double* ess_trace; int32_t* resampled;
    int force_no_resample;     // SIS: never resample
    const double* all_totals; int world, rank;   // phase 2: all-gathered per-rank {M, W, Q}
    double* local_totals;      // phase 1 out: {M_local, W_local, Q_local}
    int phase;                 // 0: single shard (everything); 1: local part; 2: combine ranks
    double* log_z_out;         // optional: the running log evidence after this step's bookkeeping (building block)
    int grid_refs;             // the tile references sit on the grid {k ln 2} (tile_partial(..., grid)): rescale with ldexp
    int exchange;              // phase 2: resampling is global and exact (offspring of remote sources migrate in)
    double* obound;            // phase 2, exchange: [world + 1] first output index owned by each rank's sources
};

// Bookkeeping of a weighted generation once (M, W, Q) sit in ctrl: ESS, resampling decision, evidence,
// the local resampling scale, the next systematic offset.  One thread.
__device__ __forceinline__ void scan_tail(const ScanArgs& a)
{
    StepCtrl* ctrl = a.ctrl;
    const double W = ctrl->W, Q = ctrl->Q, M = ctrl->M;
    const double ess = W * W / Q;
    ctrl->ess = ess;
    const bool last = a.t + 1 == a.T;
    const bool rs = !a.force_no_resample && !last && (ess < a.ess_frac * a.n_pop);
    ctrl->do_resample = rs ? 1 : 0;
    if (a.t == 0 || a.force_no_resample) { ctrl->log_z = 0.0; ctrl->n_resampled = 0; }      // (SIS keeps books once, at its last observe: a repeated run starts over too)
    if (rs || last) ctrl->log_z += M + log(W / a.n_pop);
    if (rs) ctrl->n_resampled += 1;
    // Resampling is local to the shard (particles never migrate): outputs 0..n_local-1 are drawn over the
    // shard's own CDF, and the shard's share of the total mass is carried by lw_after
    // (distributed resampling with non-proportional allocation; exact for one shard, where lw_after = 0).
    const double w_loc_units = a.bc[a.nb];                       // local sum in local-max units
    ctrl->inv_stepw = a.n_local / w_loc_units;
    ctrl->lw_after = (a.phase == 2 && !a.exchange) ? log((ctrl->w_local / a.n_local) / (W / a.n_pop)) : 0.0;
    ctrl->inv_global = a.n_pop / W;
    if (a.phase != 2) ctrl->g_end = INFINITY;
    const u32x4 r = draw_block(a.seed, 0, kResampleDrawBase + (uint64_t)(a.t + 1));
    ctrl->u0 = u01_53(r.x, r.y);
    if (a.ess_trace) a.ess_trace[a.t] = ess;
    if (a.resampled) a.resampled[a.t] = rs ? 1 : 0;
    if (a.log_z_out) *a.log_z_out = ctrl->log_z;
}

// Phase 2 of a sharded step: the all-gathered per-rank {max, sum, sum of squares} into ctrl (one thread).
__device__ __forceinline__ void scan_combine_ranks(const ScanArgs& a)
{
    StepCtrl* ctrl = a.ctrl;
    double M = -INFINITY;
    for (int r = 0; r < a.world; ++r) M = fmax(M, a.all_totals[3 * r]);
    double W = 0.0, Q = 0.0, lo = 0.0, wl = 0.0, sc = 1.0;
    for (int r = 0; r < a.world; ++r) {
        const double mr = a.all_totals[3 * r];
        const double e = (mr == -INFINITY) ? 0.0 : exp(mr - M);
        if (r == a.rank) { lo = W; wl = a.all_totals[3 * r + 1] * e; sc = e; }
        W += a.all_totals[3 * r + 1] * e;
        Q += a.all_totals[3 * r + 2] * (e * e);
    }
    ctrl->M = M; ctrl->W = W; ctrl->Q = Q; ctrl->cdf_lo = lo; ctrl->w_local = wl; ctrl->scale = sc;
}

// Exchange scope: the sources of rank s own the outputs [o_s, o_{s+1}), o_s = G(B_s) with B_s the global CDF at the
// start of the shard -- evaluated with exactly the expression the ancestor search uses for its first tile, so the
// tile-level clamps of every shard meet without gap or overlap.  One thread; after scan_tail.
__device__ __forceinline__ double g_of(double c, double inv, double u0);
__device__ __forceinline__ void scan_exchange_bounds(const ScanArgs& a)
{
    StepCtrl* ctrl = a.ctrl;
    ctrl->g_end = INFINITY;
    if (!a.exchange) return;
    const double M = ctrl->M, inv = ctrl->inv_global, u0 = ctrl->u0;
    double B = 0.0;
    for (int r = 0; r <= a.world; ++r) {
        double o = (r == a.world) ? a.n_pop : g_of(B, inv, u0);
        o = fmin(fmax(o, 0.0), a.n_pop);
        if (a.obound) a.obound[r] = o;
        if (r == a.rank + 1 && r < a.world) ctrl->g_end = o;
        if (r < a.world) {
            const double mr = a.all_totals[3 * r];
            B += a.all_totals[3 * r + 1] * ((mr == -INFINITY) ? 0.0 : exp(mr - M));
        }
    }
    if (a.obound) a.obound[a.world + 1] = (double)ctrl->do_resample;
}

__global__ __launch_bounds__(kScanThreads) void scan_partials_kernel(ScanArgs a)
{
    constexpr int NW = kScanThreads / kWave;
    __shared__ double s_scr[3 * NW];
    const int tid = threadIdx.x;
    StepCtrl* ctrl = a.ctrl;
    // NOTE: every thread of the workgroup takes every barrier below (uniform trip counts)

    if (a.phase != 2) {
        const int pst = part_stride(a.nb);
        const double* pm = a.part; const double* psum = a.part + pst; const double* pq = a.part + 2 * pst;
        double m = -INFINITY;
        for (int c = tid; c < a.nb; c += kScanThreads) m = fmax(m, pm[c]);          // coalesced
        const double M = block_max<NW>(m, s_scr);
        // tiles in index order, 4096 at a time (4 consecutive tiles per lane: 32-B coalesced loads): one
        // workgroup scan per slab plus a running carry
        double W = 0.0, Qacc = 0.0;
        int it = 0;
        for (int base = 0; base < a.nb; base += 4 * kScanThreads, ++it) {
            const int c0 = base + 4 * tid;
            double e[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                e[k] = 0.0; v[k] = 0.0;
                if (c < a.nb) {
                    const double mc = pm[c];
                    e[k] = rescale_factor(mc, M, a.grid_refs != 0);
                    v[k] = psum[c] * e[k];
                    Qacc += pq[c] * (e[k] * e[k]);
                }
            }
            double tot;
            const double excl = block_excl_scan<NW>((v[0] + v[1]) + (v[2] + v[3]), s_scr + NW + (it & 1) * NW, &tot);
            double run = W + excl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = c0 + k;
                if (c < a.nb) { a.bc[c] = run; a.bf[c] = e[k]; }
                run += v[k];
            }
            W += tot;
        }
        __syncthreads();
        const double Qt = block_sum<NW>(Qacc, s_scr);
        if (tid == 0) {
            a.bc[a.nb] = W;
            if (a.phase == 1) {
                a.local_totals[0] = M; a.local_totals[1] = W; a.local_totals[2] = Qt;
            } else {
                ctrl->M = M; ctrl->W = W; ctrl->Q = Qt;
                ctrl->cdf_lo = 0.0; ctrl->w_local = W; ctrl->scale = 1.0;
            }
        }
        if (a.phase == 1) return;
    } else {
        // combine the all-gathered per-rank totals (tiny: world <= 64), thread 0 only
        if (tid == 0) scan_combine_ranks(a);
    }
    if (tid == 0) scan_tail(a);
    if (tid == 0 && a.phase == 2) scan_exchange_bounds(a);
}

// ---------------------------------------------------------------------------------------------
// Large populations (more than kSlabThreshold tiles): one workgroup cannot stream the partials fast
// enough (a single CU moves ~25 GB/s), so the same normalisation runs as two multi-workgroup launches
// over slabs of 1024 tiles:
//   scan_slab_partials_kernel : slab g -> {max, sum, sum of squares} relative to the slab max
//   scan_slab_finish_kernel   : every workgroup combines the <= 1024 slab partials (redundantly,
//                               identically), then writes bc[] / bf[] of its own slab against the global
//                               max; workgroup 0 does the bookkeeping.
// Consumers see exactly the arrays scan_partials_kernel would have produced.
// ---------------------------------------------------------------------------------------------
constexpr int kSlabTiles = kThreads * 4;    // 1024 tiles per workgroup: 4 consecutive tiles per lane
constexpr int kSlabThreshold = 4096;
constexpr int kMaxSlabs = 1024;

// Optional SIS read-out columns (kMaxReadoutCols at most): stile[j][nb] holds, per tile, sum e * f_j(x) relative to the tile's own
// reference (sis_kernel<Model, true>); they are carried through both launches like the tile sums, so that the weighted moments come
// out of the normalisation itself and the particle store is never read back.
constexpr int kMaxReadoutCols = 4;

__global__ __launch_bounds__(kThreads) void scan_slab_partials_kernel(const double* __restrict__ part, int nb, double* __restrict__ gpart, int grid_refs,
                                                                       const double* __restrict__ stile = nullptr, int n_col = 0, double* __restrict__ gstat = nullptr)
{
    __shared__ double s_scr[3 * kWaves];
    __shared__ double s_col[2 * kWaves];
    const int G = (int)gridDim.x, g = (int)blockIdx.x;
    const int pst = part_stride(nb);
    const double* pm = part; const double* psum = part + pst; const double* pq = part + 2 * pst;
    const int c0 = g * kSlabTiles + (int)threadIdx.x * 4;
    double m[4], sv[4], qv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + k;
        const bool in = c < nb;
        m[k] = in ? pm[c] : -INFINITY; sv[k] = in ? psum[c] : 0.0; qv[k] = in ? pq[c] : 0.0;
    }
    const double mg = block_max(fmax(fmax(m[0], m[1]), fmax(m[2], m[3])), s_scr);
    double S = 0.0, Q = 0.0, ef[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double e = rescale_factor(m[k], mg, grid_refs != 0);
        ef[k] = e;
        S += sv[k] * e; Q += qv[k] * (e * e);
    }
    block_sum2(S, Q, s_scr + kWaves);
    if (threadIdx.x == 0) { gpart[g] = mg; gpart[G + g] = S; gpart[2 * G + g] = Q; }
    for (int j = 0; j < n_col; j += 2) {                          // workgroup-uniform trip count
        double ca = 0.0, cb = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + k;
            if (c < nb) {
                ca += stile[(size_t)j * nb + c] * ef[k];
                if (j + 1 < n_col) cb += stile[(size_t)(j + 1) * nb + c] * ef[k];
            }
        }
        __syncthreads();                                          // s_col is reused by the next pair
        block_sum2(ca, cb, s_col);
        if (threadIdx.x == 0) { gstat[(size_t)j * G + g] = ca; if (j + 1 < n_col) gstat[(size_t)(j + 1) * G + g] = cb; }
    }
}

__global__ __launch_bounds__(kThreads) void scan_slab_finish_kernel(ScanArgs a, const double* __restrict__ gpart, int G,
                                                                     const double* __restrict__ gstat = nullptr, int n_col = 0, int K = 0, int is_int = 0,
                                                                     double* __restrict__ stats = nullptr)
{
    __shared__ double s_scr[4 * kWaves];
    __shared__ double s_col[kMaxReadoutCols][kWaves];
    __shared__ double s_ctot[kMaxReadoutCols];
    const int tid = threadIdx.x, g = (int)blockIdx.x;
    // combine the slab partials: global max, total mass, total squares, mass of the slabs before mine
    constexpr int kPerG = kMaxSlabs / kThreads;
    double gm[kPerG], gs[kPerG], gq[kPerG];
    double m = -INFINITY;
#pragma unroll
    for (int i = 0; i < kPerG; ++i) {
        const int j = tid + i * kThreads;
        const bool in = j < G;
        gm[i] = in ? gpart[j] : -INFINITY; gs[i] = in ? gpart[G + j] : 0.0; gq[i] = in ? gpart[2 * G + j] : 0.0;
        m = fmax(m, gm[i]);
    }
    const double M = block_max(m, s_scr);
    double Wt = 0.0, Qt = 0.0, before = 0.0, ct[kMaxReadoutCols] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < kPerG; ++i) {
        const int j = tid + i * kThreads;
        const double e = rescale_factor(gm[i], M, a.grid_refs != 0);
        const double v = gs[i] * e;
        Wt += v; Qt += gq[i] * (e * e);
        if (j < g) before += v;
        if (g == 0 && j < G) {
            for (int col = 0; col < n_col; ++col) ct[col] += gstat[(size_t)col * G + j] * e;
        }
    }
    block_sum2(Wt, Qt, s_scr + kWaves);
    before = block_sum(before, s_scr + 3 * kWaves);
    if (g == 0 && n_col > 0) {                                   // workgroup-uniform
        for (int col = 0; col < n_col; ++col) {
            const double tot = block_sum(ct[col], s_col[col]);
            if (tid == 0) s_ctot[col] = tot;
        }
        __syncthreads();
        if (tid == 0) {
            // finalize_kernel's normalisation: real -> {mean, raw2 - mean^2} per row, int -> probabilities
            for (int r = 0; r * K < n_col; ++r) {
                if (is_int) { for (int j = 0; j < K; ++j) stats[r * K + j] = s_ctot[r * K + j] / Wt; }
                else {
                    const double mean = s_ctot[r * K] / Wt;
                    stats[r * K] = mean;
                    stats[r * K + 1] = s_ctot[r * K + 1] / Wt - mean * mean;
                }
            }
        }
    }
    // my slab against the global max
    const double* pm = a.part; const double* psum = a.part + part_stride(a.nb);
    const int c0 = g * kSlabTiles + tid * 4;
    double e[4], v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + k;
        e[k] = 0.0; v[k] = 0.0;
        if (c < a.nb) {
            const double mc = pm[c];
            e[k] = rescale_factor(mc, M, a.grid_refs != 0);
            v[k] = psum[c] * e[k];
        }
    }
    __syncthreads();                                            // s_scr is about to be reused
    double tot;
    const double excl = block_excl_scan((v[0] + v[1]) + (v[2] + v[3]), s_scr, &tot);
    double run = before + excl;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + k;
        if (c < a.nb) { a.bc[c] = run; a.bf[c] = e[k]; }
        run += v[k];
    }
    if (g == 0 && tid == 0) {
        a.bc[a.nb] = Wt;
        if (a.phase == 1) {
            a.local_totals[0] = M; a.local_totals[1] = Wt; a.local_totals[2] = Qt;
        } else {
            a.ctrl->M = M; a.ctrl->W = Wt; a.ctrl->Q = Qt; a.ctrl->cdf_lo = 0.0; a.ctrl->w_local = Wt; a.ctrl->scale = 1.0;
            scan_tail(a);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Ancestor search for one tile of 1024 consecutive outputs.
//
// Tile-level CDF window: bc[] entries around the source tile the outputs are expected in
// (tile sums are nearly equal) are staged once in LDS; HBM is only searched when weights are so
// uneven that the target lies outside the window.
// ---------------------------------------------------------------------------------------------
constexpr int kWin = 256;

// in-lane inclusive prefix over the lane's kPPT values
template <class T> __device__ __forceinline__ void lane_prefix_sum(T (&w)[kPPT])
{
#pragma unroll
    for (int k = 1; k < kPPT; ++k) w[k] += w[k - 1];
}
// (lane_prefix_max, lane_fill: cpprob/detail/wave.hpp)
template <class T> __device__ __forceinline__ void lane_copy(T (&d)[kPPT], const T (&s)[kPPT])
{
#pragma unroll
    for (int k = 0; k < kPPT; ++k) d[k] = s[k];
}

#ifdef CPPROB_STAMPS
__device__ unsigned long long* g_stamps = nullptr;   // diagnostic build only: [nb][16] s_memrealtime stamps
#define CPH_STAMP(k) do { if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#elif defined(CPPROB_MARKS)
// static instruction budgets (tools/isa_budget.py): a comment line in the assembly at every stamp; instructions are counted between them
#define CPH_STAMP(k) asm volatile("; CPH_MARK " #k ::: "memory")
#else
#define CPH_STAMP(k) do {} while (0)
#endif

struct AncestorLds {
    union { double cdf[kTile]; int32_t slot[kTile]; } u;   // stratified: tile CDF; systematic: scatter slots
    double bcw[kWin + 1];    // bc[w0 .. w0 + kWin], rescaled to the global max
    double scr[2][kWaves];   // scan scratch, double-buffered across source tiles
    int iscr[kWaves];
    double pos[2];
    int cnt;
};

struct AncestorIn {
    const double* wrel; const double* bc; const double* bf; int nb; int64_t n_in;
    double W, scale, cdf_lo, u0, inv_stepw;
    uint64_t seed, step, gj_tile0, n_total_out; int n_valid_tile;
    uint64_t id0;      // RNG id of output 0 (stratified offsets are drawn per global particle id)
    int bc_in_lds;     // bc / bf already point at LDS copies of the whole tile-level CDF (fused step kernel)
    int guess;         // >= 0: source tile the first output is expected in (equal shard sizes: the output tile's own index)
    double g_end;      // systematic: first output owned by whatever follows this CDF (+inf unless another shard's sources do)
};

__device__ __forceinline__ int stage_window(const AncestorIn& in, AncestorLds& L, double q_guess)
{
    const int tid = threadIdx.x;
    if (in.bc_in_lds) return in.nb + 1;                      // empty window beyond the table: every lookup reads in.bc (LDS) directly
    const double w_local = in.bc[in.nb] * in.scale;
    const int guess = w_local > 0.0 ? (int)(fmin(fmax(q_guess / w_local, 0.0), 1.0) * in.nb) : 0;
    int w0 = guess - kWin / 2;
    if (w0 > in.nb - kWin) w0 = in.nb - kWin;
    if (w0 < 0) w0 = 0;
    const int wn = in.nb - w0 < kWin ? in.nb - w0 : kWin;    // window covers tiles [w0, w0 + wn), entries [0, wn]
    if (tid <= wn) L.bcw[tid] = in.bc[w0 + tid] * in.scale;
    if (tid == 0 && wn == kWin) L.bcw[kWin] = in.bc[w0 + kWin] * in.scale;
    return w0;
}

// ---- systematic: inverse (offspring-range) form ---------------------------------------------
// With positions (j + u0) * W/N, source k owns the outputs j in [G(C_{k-1}), G(C_k)),
// G(C) = ceil(C * N/W - u0), C the inclusive CDF: a partition of the outputs (G is monotone).
// Each visited source tile rebuilds its CDF from wrel (one scan), every source with a non-empty
// range writes its index into the LDS slot of its FIRST output, and one prefix-max over the
// 1024 slots hands every output its ancestor.  Cost per tile is independent of how the weights
// are spread (one heavy particle = one slot write), and there is no per-output search.
__device__ __forceinline__ double g_of(double c, double inv, double u0) { return ceil(c * inv - u0); }

// plain source of linear weights: the stored array
struct WrelSource {
    const double* wrel;
    __device__ __forceinline__ void load(int64_t i0, double (&w)[kPPT]) const { load4(wrel, i0, w); }
};

template <class WS>
__device__ __forceinline__ void ancestors_systematic(const AncestorIn& in, int32_t (&anc)[kPPT], AncestorLds& L, const WS& ws)
{
    const int tid = threadIdx.x;
    const double inv = in.inv_stepw, u0 = in.u0;
    const double gj_first = (double)in.gj_tile0, gj_last = (double)(in.gj_tile0 + (uint64_t)in.n_valid_tile - 1);
    const int w0 = stage_window(in, L, gj_first / inv - in.cdf_lo);
    const int wn = in.nb - w0 < kWin ? in.nb - w0 : kWin;
    {
        int32_t neg[kPPT];
        lane_fill(neg, (int32_t)-1);
        store4(L.u.slot, (int64_t)tid * kPPT, neg);
    }
    __syncthreads();
    CPH_STAMP(2);
    // tile-level value: start index of tile c's outputs
    auto bcv = [&](int c) -> double { return (c >= w0 && c <= w0 + wn) ? L.bcw[c - w0] : in.bc[c] * in.scale; };
    auto gt = [&](int c) -> double { return c >= in.nb ? in.g_end : g_of(in.cdf_lo + bcv(c), inv, u0); };
    // largest c in [lo, nb) with gt(c) <= g  (gt(lo) <= g guaranteed).  Tile masses are nearly equal, so the
    // answer is within a tile or two of `guess`: probe there first, binary search only when that fails.
    auto locate = [&](double g, int lo, int guess) -> int {
        int c = guess < lo ? lo : (guess >= in.nb ? in.nb - 1 : guess);
        int probes = 0;
        while (probes < 6 && c > lo && gt(c) > g) { --c; ++probes; }
        while (probes < 6 && c + 1 < in.nb && gt(c + 1) <= g) { ++c; ++probes; }
        if (gt(c) <= g && (c + 1 >= in.nb || gt(c + 1) > g)) return c;
        int a = lo, b = in.nb;
        if (!in.bc_in_lds) {
            if (w0 >= lo && gt(w0) <= g) a = w0;
            if (w0 + wn < in.nb && w0 + wn > a && gt(w0 + wn) > g) b = w0 + wn;
        }
        while (b - a > 1) { const int mid = (a + b) >> 1; if (gt(mid) <= g) a = mid; else b = mid; }
        return a;
    };
    const int guess0 = in.guess >= 0 ? in.guess : (int)((gj_first / inv - in.cdf_lo) / fmax(bcv(in.nb), 1e-300) * in.nb);
    const int c_lo = locate(gj_first, 0, guess0);
    const int c_hi = locate(gj_last, c_lo, c_lo + 1);
    CPH_STAMP(3);
    // linear weights of the 4 sources this lane owns in source tile c
    auto load_w = [&](int c, double (&w)[kPPT]) {
        const int64_t i0 = (int64_t)c * kTile + (int64_t)tid * kPPT;
        ws.load(i0, w);
    };
    // the two tiles an output tile normally overlaps are fetched together (one memory round trip)
    double w_first[kPPT], w_second[kPPT];
    lane_fill(w_second, 0.0);
    load_w(c_lo, w_first);
    if (c_lo + 1 <= c_hi) load_w(c_lo + 1, w_second);
    if (w_first[0] == -1.0) CPH_STAMP(15);
    CPH_STAMP(4);
    int it = 0;
    for (int c = c_lo; c <= c_hi; ++c) {
        const double b0 = bcv(c), b1 = bcv(c + 1);
        if (!(b1 > b0)) continue;                              // tile without mass: owns no output
        double w[kPPT];
        if (c == c_lo) lane_copy(w, w_first);
        else if (c == c_lo + 1) lane_copy(w, w_second);
        else load_w(c, w);
        lane_prefix_sum(w);
        double tot;
        const double excl = block_excl_scan(w[kPPT - 1], L.scr[it & 1], &tot);
        ++it;
        const double bfc = in.bf[c] * in.scale;
        const double off = in.cdf_lo + b0;
        const double g_lo = g_of(off, inv, u0), g_hi = (c + 1 >= in.nb) ? in.g_end : g_of(in.cdf_lo + b1, inv, u0);
        double g_prev = fmin(fmax(g_of(off + bfc * excl, inv, u0), g_lo), g_hi);
        if (tid == 0) g_prev = g_lo;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            double g = fmin(fmax(g_of(off + bfc * (excl + w[k]), inv, u0), g_lo), g_hi);
            if (tid == kThreads - 1 && k == kPPT - 1) g = g_hi;           // the tile ends where the next one starts
            if (g > g_prev) {
                const double s = g_prev - gj_first, e = g - gj_first;    // exact: integers
                if (e > 0.0 && s < (double)kTile) L.u.slot[s > 0.0 ? (int)s : 0] = (int32_t)((int64_t)c * kTile + tid * kPPT + k);
            }
            g_prev = g;
        }
    }
    CPH_STAMP(5);
    __syncthreads();
    CPH_STAMP(6);
    // inclusive prefix-max over the 1024 slots
    int32_t v[kPPT];
    load4(L.u.slot, (int64_t)tid * kPPT, v);
    lane_prefix_max(v);
    int32_t incl = wave_incl_max_i32(v[kPPT - 1]);
    if (lane_id() == kWave - 1) L.iscr[wave_id()] = incl;
    int32_t excl = dpp_or_i32<0x138 /* wave_shr:1 */>(incl, -1);
    if (lane_id() == 0) excl = -1;
    __syncthreads();
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv)
        if (wv < wave_id()) excl = max(excl, L.iscr[wv]);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) anc[k] = max(max(v[k], excl), 0);
    CPH_STAMP(7);
}

// Where the fused step kernel finds the linear weights of generation t-1: the stored array, or -- table-weight models
// on an every-step resampling schedule, where every weight is e_tab[t-1][state] -- the stored STATES (4 bytes instead
// of 8 read per source, and no wrel store at all between steps).
template <class Model>
struct WeightSource {
    using V = typename Model::value_t;
    static constexpr int K = Model::kWeightTable > 0 ? Model::kWeightTable : 1;
    using S = typename Model::store_t;
    const double* wrel; const S* states; int64_t n; double e[K]; bool from_states;
    __device__ __forceinline__ void load(int64_t i0, double (&w)[kPPT]) const
    {
        if (Model::kWeightTable > 0 && from_states) {
            V st[kPPT];
            load4_as(states, i0, st);
#pragma unroll
            for (int k = 0; k < kPPT; ++k) {
                const int idx = Model::weight_index(st[k]);
                double v = e[0];
#pragma unroll
                for (int s2 = 1; s2 < K; ++s2) v = idx == s2 ? e[s2] : v;
                w[k] = (i0 + k < n) ? v : 0.0;                       // padding slots weigh nothing
            }
        } else {
            load4(wrel, i0, w);
        }
    }
};

// Fast form for the fused step kernel: the tile-level CDF is in LDS (bc, bf), the slots were reset before the
// caller's last barrier, and the linear weights of the lane's own-index source tile (w_own: tile blockIdx.x, the one an
// output tile overlaps almost surely) were fetched at kernel entry.  The start indices of tiles b-1 .. b+2 are
// evaluated together; anything farther away (very uneven masses) falls back to a search of the LDS table.
template <class WS>
__device__ __forceinline__ void ancestors_systematic_fused(const WS& ws, const double* s_bc, const double* s_bf, int nb,
                                                            double u0, double inv, int n_valid_tile, const double (&w_own)[kPPT],
                                                            int32_t (&anc)[kPPT], AncestorLds& L, const int b /* this workgroup's output tile */)
{
    const int tid = threadIdx.x;
    const double gj_first = (double)((int64_t)b * kTile), gj_last = gj_first + (double)(n_valid_tile - 1);
    auto gt = [&](int c) -> double { return c >= nb ? INFINITY : (c < 0 ? -INFINITY : g_of(s_bc[c], inv, u0)); };
    const double g_m1 = gt(b - 1), g_0 = gt(b), g_p1 = gt(b + 1), g_p2 = gt(b + 2), g_p3 = gt(b + 3);
    int c_lo, c_hi;
    if (g_0 <= gj_first && gj_first < g_p1) c_lo = b;
    else if (g_m1 <= gj_first && gj_first < g_0) c_lo = b - 1;
    else if (g_p1 <= gj_first && gj_first < g_p2) c_lo = b + 1;
    else {                                                       // rare: search the table
        int lo = 0, hi = nb;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (gt(mid) <= gj_first) lo = mid; else hi = mid; }
        c_lo = lo;
    }
    {
        const double e1 = c_lo == b ? g_p1 : (c_lo == b - 1 ? g_0 : (c_lo == b + 1 ? g_p2 : gt(c_lo + 1)));
        const double e2 = c_lo == b ? g_p2 : (c_lo == b - 1 ? g_p1 : (c_lo == b + 1 ? g_p3 : gt(c_lo + 2)));
        if (gj_last < e1) c_hi = c_lo;
        else if (gj_last < e2) c_hi = c_lo + 1;
        else {
            int lo = c_lo, hi = nb;
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (gt(mid) <= gj_last) lo = mid; else hi = mid; }
            c_hi = lo;
        }
    }
    auto load_w = [&](int c, double (&w)[kPPT]) { ws.load((int64_t)c * kTile + (int64_t)tid * kPPT, w); };
    int it = 0;
    auto process = [&](int c, double (&w)[kPPT]) {
        const double b0 = s_bc[c], b1 = s_bc[c + 1];
        if (!(b1 > b0)) return;                                  // tile without mass: owns no output (workgroup-uniform)
        lane_prefix_sum(w);
        double tot;
        const double excl = block_excl_scan(w[kPPT - 1], L.scr[it & 1], &tot);
        ++it;
        const double bfc = s_bf[c];
        const double g_lo = g_of(b0, inv, u0), g_hi = (c + 1 >= nb) ? INFINITY : g_of(b1, inv, u0);
        double g_prev = fmin(fmax(g_of(b0 + bfc * excl, inv, u0), g_lo), g_hi);
        if (tid == 0) g_prev = g_lo;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            double g = fmin(fmax(g_of(b0 + bfc * (excl + w[k]), inv, u0), g_lo), g_hi);
            if (tid == kThreads - 1 && k == kPPT - 1) g = g_hi;   // the tile ends where the next one starts
            if (g > g_prev) {
                const double s = g_prev - gj_first, e = g - gj_first;    // exact: integers
                if (e > 0.0 && s < (double)kTile) L.u.slot[s > 0.0 ? (int)s : 0] = (int32_t)((int64_t)c * kTile + tid * kPPT + k);
            }
            g_prev = g;
        }
    };
    // the other tile's weights travel while the own-index tile is processed (slot writes of different source tiles
    // never collide and the prefix-max below does not care about their order)
    const bool own_in = b >= c_lo && b <= c_hi;
    const int c_other = (c_lo != b) ? c_lo : c_lo + 1;
    double w_other[kPPT];
    lane_fill(w_other, 0.0);
    const bool other_in = c_other <= c_hi && c_other != b;
    if (other_in) load_w(c_other, w_other);
    if (own_in) { double w[kPPT]; lane_copy(w, w_own); process(b, w); }
    if (other_in) process(c_other, w_other);
    for (int c = c_lo; c <= c_hi; ++c) {
        if (c == b || c == c_other) continue;
        double w[kPPT];
        load_w(c, w);
        process(c, w);
    }
    __syncthreads();
    // inclusive prefix-max over the 1024 slots
    int32_t v[kPPT];
    load4(L.u.slot, (int64_t)tid * kPPT, v);
    lane_prefix_max(v);
    int32_t incl = wave_incl_max_i32(v[kPPT - 1]);
    if (lane_id() == kWave - 1) L.iscr[wave_id()] = incl;
    int32_t excl = dpp_or_i32<0x138 /* wave_shr:1 */>(incl, -1);
    if (lane_id() == 0) excl = -1;
    __syncthreads();
#pragma unroll
    for (int wv = 0; wv < kWaves; ++wv)
        if (wv < wave_id()) excl = max(excl, L.iscr[wv]);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) anc[k] = max(max(v[k], excl), 0);
}

// ---- stratified: forward search --------------------------------------------------------------
// position of output j: (j + u_j) * W/N, u_j = 32-bit uniform of particle id j; ancestor =
// min{k : C_k > p}.  Positions are sorted, so the tile walks the source tiles they fall into,
// rebuilds each CDF in LDS and searches it (first output: binary search; the other three: the
// neighbour's result plus a short probe).
__device__ __forceinline__ void ancestors_stratified(const AncestorIn& in, int32_t (&anc)[kPPT], AncestorLds& L)
{
    const int tid = threadIdx.x;
    const double stepw = in.W / (double)in.n_total_out;
    const uint64_t gj0 = in.gj_tile0 + (uint64_t)tid * kPPT;
    double p[kPPT];
    {
        uint32_t wd[kPPT];
#pragma unroll
        for (int q = 0; q < kPPT; q += 4) draw_words4(in.seed, in.id0 + gj0 + q, kResampleDrawBase + in.step, reinterpret_cast<uint32_t(&)[4]>(wd[q]));
#pragma unroll
        for (int k = 0; k < kPPT; ++k) { p[k] = ((double)(gj0 + k) + u01_32(wd[k])) * stepw - in.cdf_lo; anc[k] = 0; }
    }
    const int w0 = stage_window(in, L, (double)in.gj_tile0 * stepw - in.cdf_lo);
    const int wn = in.nb - w0 < kWin ? in.nb - w0 : kWin;
    if (tid == 0) { L.pos[0] = p[0]; L.cnt = 0; }
    {
        const int last = in.n_valid_tile - 1;
        if (last / kPPT == tid) {
            double pl = p[0];
#pragma unroll
            for (int k = 1; k < kPPT; ++k) pl = (last % kPPT) == k ? p[k] : pl;
            L.pos[1] = pl;
        }
    }
    __syncthreads();
    const double p_first = fmax(L.pos[0], 0.0), p_last = L.pos[1];
    auto bcv = [&](int c) -> double { return (c >= w0 && c <= w0 + wn) ? L.bcw[c - w0] : in.bc[c] * in.scale; };
    auto locate = [&](double q, int lo) -> int {               // largest c in [lo, nb) with bc[c] <= q
        int a = lo, b = in.nb;
        if (!in.bc_in_lds) {
            if (w0 >= lo && bcv(w0) <= q) a = w0;
            if (w0 + wn < in.nb && w0 + wn > a && bcv(w0 + wn) > q) b = w0 + wn;
        }
        while (b - a > 1) { const int mid = (a + b) >> 1; if (bcv(mid) <= q) a = mid; else b = mid; }
        return a;
    };
    bool res[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) res[k] = (tid * kPPT + k) >= in.n_valid_tile;
    int c = locate(p_first, 0);
    int it = 0;
    for (;;) {
        double w[kPPT];
        load4(in.wrel, (int64_t)c * kTile + (int64_t)tid * kPPT, w);
        lane_prefix_sum(w);
        double tot;
        const double excl = block_excl_scan(w[kPPT - 1], L.scr[it & 1], &tot);
        ++it;
        const double bfc = in.bf[c] * in.scale;
        const double off = bcv(c);
        double cd[kPPT];
#pragma unroll
        for (int k = 0; k < kPPT; ++k) cd[k] = off + bfc * (excl + w[k]);
        store4(L.u.cdf, (int64_t)tid * kPPT, cd);
        __syncthreads();
        const double hi_c = (c == in.nb - 1) ? INFINITY : bcv(c + 1);
        const int64_t rem = in.n_in - (int64_t)c * kTile;
        const int n_src = rem < kTile ? (int)rem : kTile;
        auto lower = [&](double q, int from) -> int {          // first index >= from with cdf > q (n_src if none)
            int a = from, b = n_src;
            while (a < b) { const int mid = (a + b) >> 1; if (L.u.cdf[mid] > q) b = mid; else a = mid + 1; }
            return a;
        };
        int newly = 0, prev_pos = 0;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            if (!res[k] && p[k] < hi_c) {
                int pos = prev_pos;
                // short probe from the neighbour's ancestor, then binary search
                int probe = 0;
                while (probe < 3 && pos < n_src && L.u.cdf[pos] <= p[k]) { ++pos; ++probe; }
                if (pos < n_src && L.u.cdf[pos] <= p[k]) pos = lower(p[k], pos);
                if (pos >= n_src) pos = n_src - 1;
                prev_pos = pos;
                anc[k] = (int32_t)((int64_t)c * kTile + pos);
                res[k] = true;
                ++newly;
            }
        }
        if (p_last < hi_c) break;   // workgroup-uniform: every output of this tile is resolved
        if (newly) atomicAdd(&L.cnt, newly);
        __syncthreads();
        // outputs are sorted, so the resolved ones are a prefix: the next position is that of output #cnt
        const uint64_t gjn = in.gj_tile0 + (uint64_t)L.cnt;
        const double p_next = ((double)gjn + u01_32(draw_word(in.seed, in.id0 + gjn, kResampleDrawBase + in.step))) * stepw - in.cdf_lo;
        c = locate(p_next, c + 1);
    }
}

template <int RS, class WS>
__device__ __forceinline__ void find_ancestors(const AncestorIn& in, int32_t (&anc)[kPPT], AncestorLds& L, const WS& ws)
{
    if (RS == RS_SYSTEMATIC) ancestors_systematic(in, anc, L, ws);
    else ancestors_stratified(in, anc, L);
}

// Standalone resampler over an arbitrary weight array (building block).
struct ResampleArgs {
    const double* wrel; int64_t n_in;
    const double* bc; const double* bf; int nb; const StepCtrl* ctrl;
    uint64_t seed, step, j0, n_total_out; int64_t n_out;
    int32_t* anc;
    int run_ctrl;      // u0 / inv / g_end of a running SMC step (exchange scope) instead of (seed, step, n_total_out)
    int identity_unless_resampling;   // ctrl->do_resample == 0: every output is its own ancestor (device-side decision)
};

template <int RS>
__global__ __launch_bounds__(kThreads) void resample_kernel(ResampleArgs a)
{
    __shared__ AncestorLds L;
    const int64_t l0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    if (a.identity_unless_resampling && !a.ctrl->do_resample) {     // workgroup-uniform, before any barrier
#pragma unroll
        for (int k = 0; k < kPPT; ++k)
            if (l0 + k < a.n_out) a.anc[l0 + k] = (int32_t)(a.j0 + (uint64_t)(l0 + k));
        return;
    }
    const int64_t rem = a.n_out - (int64_t)blockIdx.x * kTile;
    AncestorIn in;
    in.wrel = a.wrel; in.bc = a.bc; in.bf = a.bf; in.nb = a.nb; in.n_in = a.n_in;
    in.W = a.ctrl->W; in.scale = a.ctrl->scale; in.cdf_lo = a.ctrl->cdf_lo;
    in.inv_stepw = (double)a.n_total_out / a.ctrl->W;
    in.g_end = INFINITY;
    if (a.run_ctrl) { in.u0 = a.ctrl->u0; in.inv_stepw = a.ctrl->inv_global; in.g_end = a.ctrl->g_end; }
    else {
        const u32x4 r = draw_block(a.seed, 0, kResampleDrawBase + a.step);
        in.u0 = u01_53(r.x, r.y);
    }
    in.seed = a.seed; in.step = a.step; in.gj_tile0 = a.j0 + (uint64_t)blockIdx.x * kTile; in.n_total_out = a.n_total_out;
    in.n_valid_tile = rem < kTile ? (int)rem : kTile;
    in.id0 = 0; in.bc_in_lds = 0; in.guess = -1;
    int32_t anc[kPPT];
    find_ancestors<RS>(in, anc, L, WrelSource{a.wrel});
#pragma unroll
    for (int k = 0; k < kPPT; ++k)
        if (l0 + k < a.n_out) a.anc[l0 + k] = anc[k];
}

// Multinomial (thesis Alg. 1 p.36, literal): independent positions u_j * W, unsorted, so the
// full-resolution CDF is materialised once (cdf_kernel) and searched per output.
__global__ __launch_bounds__(kThreads) void cdf_kernel(const double* __restrict__ wrel, const double* __restrict__ bc, const double* __restrict__ bf,
                                                        const StepCtrl* __restrict__ ctrl, double* __restrict__ cdf)
{
    __shared__ double s_scr[kWaves];
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    double w[kPPT];
    load4(wrel, j0, w);
    lane_prefix_sum(w);
    double tot;
    const double excl = block_excl_scan(w[kPPT - 1], s_scr, &tot);
    const double off = bc[blockIdx.x] * ctrl->scale, bfc = bf[blockIdx.x] * ctrl->scale;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) w[k] = off + bfc * (excl + w[k]);
    store4(cdf, j0, w);
}

__global__ __launch_bounds__(kThreads) void multinomial_kernel(const double* __restrict__ cdf, int64_t n_in, const StepCtrl* __restrict__ ctrl,
                                                                uint64_t seed, uint64_t step, uint64_t j0, int64_t n_out, int64_t n_pad,
                                                                int32_t* __restrict__ anc, int identity_unless_resampling)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_out) {
        if (i < n_pad) anc[i] = 0;          // padding slots of the particle store must hold a valid index
        return;
    }
    if (identity_unless_resampling && !ctrl->do_resample) { anc[i] = (int32_t)(j0 + (uint64_t)i); return; }
    const double p = draw_u01_53(seed, j0 + (uint64_t)i, kResampleDrawBase + step) * cdf[n_in - 1];
    int64_t lo = 0, hi = n_in;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] > p) hi = mid; else lo = mid + 1;
    }
    if (lo >= n_in) lo = n_in - 1;
    anc[i] = (int32_t)lo;
}

// ---------------------------------------------------------------------------------------------
// SIS: cpprob::inference(StateType::sis) for all particles at once
// (reference include/cpprob/cpprob.hpp:194-201).  One lane runs 4 particles to completion:
// draw priors -> sum logpdf(observe) -> record predicts.  Writes values[t][i], logw[i] and
// wrel[i] coalesced, plus the tile partial of the final weights.
// ---------------------------------------------------------------------------------------------
template <class Model>
struct SisArgs {
    ModelParams mp; const double* obs; int T; int64_t n, ld, rs;
    uint64_t seed, pid0;
    typename Model::store_t* values; double* logw; double* wrel; Partial* part;
    double* stile;     // READOUT: [T * kStats][gridDim.x] per-tile weighted sums (the read-out rides the normalisation)
};

constexpr int kMaxReadoutT = 2;

template <class Model, bool READOUT = false>
__global__ __launch_bounds__(kThreads) void sis_kernel(SisArgs<Model> a)
{
    using V = typename Model::value_t;
    constexpr int K = Model::kStats;
    __shared__ double s_scr[3 * kWaves];
    __shared__ double s_ro[2 * kWaves];
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    double lw[kPPT]; V x[kPPT];
    V xt[READOUT ? kMaxReadoutT : 1][kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) { lw[k] = 0.0; x[k] = V(0); }                              // start_trace(): log_w_ = 0
    for (int t = 0; t < a.T; ++t) {
        V nx[kPPT];
#pragma unroll
        for (int q = 0; q < kPPT / 4; ++q)                                                    // sample: distr(get_rng())  cpprob.hpp:72-74
            Model::propagate4(a.mp, a.seed, a.pid0 + (uint64_t)j0 + 4 * q, t, reinterpret_cast<const V(&)[4]>(x[4 * q]), reinterpret_cast<V(&)[4]>(nx[4 * q]));
#pragma unroll
        for (int k = 0; k < kPPT; ++k) { x[k] = nx[k]; lw[k] += Model::loglik(a.mp, x[k], t, a.obs); }   // observe: log_w_ += logpdf  state.cpp:212-223
        store4_as(a.values + (int64_t)t * a.rs, j0, x);                                       // predict: add_predict       state.hpp:312-327
        if (READOUT) {                                                                        // (static indices: registers)
#pragma unroll
            for (int tt = 0; tt < kMaxReadoutT; ++tt)
                if (t == tt) {
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) xt[READOUT ? tt : 0][k] = x[k];
                }
        }
    }
#pragma unroll
    for (int k = 0; k < kPPT; ++k)
        if (j0 + k >= a.n) lw[k] = -INFINITY;                                                 // padding slots
    store4(a.logw, j0, lw);                                                                   // finish_trace()
    double e[kPPT];
    tile_partial(lw, e, a.part, s_scr, (int)blockIdx.x, Model::kWeightTable == 0);
    if (!READOUT) store4(a.wrel, j0, e);          // the linear weights only serve the read-out pass, which READOUT replaces
    if (READOUT) {
        // StatsPrinter's sums for this tile, relative to the tile's reference: sum e f(x_t) per predict hit t (a.T <= kMaxReadoutT)
#pragma unroll
        for (int t = 0; t < kMaxReadoutT; ++t) {
            if (t >= a.T) break;                                    // workgroup-uniform
            double acc[K];
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = 0.0;
#pragma unroll
            for (int k = 0; k < kPPT; ++k) Model::accumulate(xt[READOUT ? t : 0][k], e[k], acc);
            for (int j = 0; j < K; j += 2) {
                double ca = acc[j], cb = j + 1 < K ? acc[j + 1] : 0.0;
                __syncthreads();                                    // s_ro is reused pair after pair
                block_sum2(ca, cb, s_ro);
                if (threadIdx.x == 0) {
                    a.stile[(size_t)(t * K + j) * gridDim.x + blockIdx.x] = ca;
                    if (j + 1 < K) a.stile[(size_t)(t * K + j + 1) * gridDim.x + blockIdx.x] = cb;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SIS for models whose log-weight has a host-known upper bound (Model::kBounded: the Gaussian models, whose weight is a quadratic
// in the prior's standard-normal variate): every weight is taken against ONE reference fixed before the launch (ModelParams::
// lw_ref), so nothing has to be reduced before the linear weights exist -- no tile maximum, no rescaling between tiles -- and a
// workgroup can carry its sums of e, e^2, e f(x) in registers over SEVERAL tiles (grid-stride, fixed tile set per workgroup: bitwise
// reproducible) and reduce across lanes once.  What the per-tile form paid per 1024 particles -- five DPP reductions, three
// barriers, the two logpdf evaluations -- is paid once per workgroup or not at all; the read-out needs one single-workgroup launch
// over <= 2048 partial rows instead of two launches over nb tile rows.  (cpprob.hpp:194-201 for all particles at once, as sis_kernel.)
// ---------------------------------------------------------------------------------------------
constexpr int kSisBoundedMaxGrid = 2048;

template <class Model>
struct SisBoundedArgs {
    ModelParams mp; int T; int64_t n, ld, rs; int nb;
    uint64_t seed, pid0;
    typename Model::store_t* values; double* logw;
    double* wpart;                                   // [2 + T * kStats][gridDim.x]: sums of e, e^2, then e f_j(x_t)
};

template <class Model, int TT>
__global__ __launch_bounds__(kThreads) void sis_bounded_kernel(SisBoundedArgs<Model> a)
{
    using V = typename Model::value_t;
    constexpr int K = Model::kStats;
    static_assert(K == 2 && TT >= 1 && TT <= kMaxReadoutT, "real-valued read-out: {sum e x, sum e x^2} per predict hit");
    __shared__ double s_ro[2 * kWaves];
    double acc[2 + TT * K];
#pragma unroll
    for (int j = 0; j < 2 + TT * K; ++j) acc[j] = 0.0;
    const double ref = a.mp.lw_ref;
    for (int tile = (int)blockIdx.x; tile < a.nb; tile += (int)gridDim.x) {
        const int64_t j0 = (int64_t)tile * kTile + (int64_t)threadIdx.x * kPPT;
        double lw[kPPT]; V xt[TT][kPPT];
#pragma unroll
        for (int k = 0; k < kPPT; ++k) lw[k] = 0.0;                                            // start_trace(): log_w_ = 0
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            typename Model::Rand r;
            const V zero[4] = {V(0), V(0), V(0), V(0)};
            Model::draw4(a.seed, a.pid0 + (uint64_t)j0, t, r);                                 // sample: distr(get_rng())  cpprob.hpp:72-74
            Model::apply4(a.mp, t, r, zero, xt[t]);
#pragma unroll
            for (int k = 0; k < kPPT; ++k) lw[k] += Model::logw_of_z(a.mp, t, r.z[k]);         // observe(s): log_w_ += logpdf  state.cpp:212-223
            store4_as(a.values + (int64_t)t * a.rs, j0, xt[t]);                               // predict: add_predict       state.hpp:312-327
        }
        if (tile == a.nb - 1) {                                                                // (workgroup-uniform: only the last tile has padding slots)
#pragma unroll
            for (int k = 0; k < kPPT; ++k)
                if (j0 + k >= a.n) lw[k] = -INFINITY;
        }
        store4(a.logw, j0, lw);                                                                // finish_trace()
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            // (branch-free: a padding slot's -inf is clamped to where the exponential underflows to exactly 0)
            const double e = exp_nonpos(fmax(lw[k] - ref, -1000.0));
            acc[0] += e; acc[1] = fma(e, e, acc[1]);
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const double ex = e * xt[t][k];
                acc[2 + t * K] += ex; acc[2 + t * K + 1] = fma(ex, xt[t][k], acc[2 + t * K + 1]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 2 + TT * K; j += 2) {
        double ca = acc[j], cb = acc[j + 1];
        if (j) __syncthreads();                                                                // s_ro is reused pair after pair
        block_sum2(ca, cb, s_ro);
        if (threadIdx.x == 0) { a.wpart[(size_t)j * gridDim.x + blockIdx.x] = ca; a.wpart[(size_t)(j + 1) * gridDim.x + blockIdx.x] = cb; }
    }
}

// One workgroup: the partial rows of sis_bounded_kernel -> ctrl (reference, W, Q, ESS, evidence) and StatsPrinter's numbers.
__global__ __launch_bounds__(kThreads) void sis_bounded_finish_kernel(const double* __restrict__ wpart, int G, int T, int K, double lw_ref, double n_pop,
                                                                       StepCtrl* __restrict__ ctrl, double* __restrict__ stats, double* __restrict__ ess_trace,
                                                                       int32_t* __restrict__ resampled, int normalise)
{
    constexpr int kCols = 2 + kMaxReadoutT * 2, kPer = kSisBoundedMaxGrid / kThreads;
    __shared__ double s_col[kCols][kWaves];
    const int n_col = 2 + T * K;
    // every load of every column first (one memory round trip), then the sums in a fixed order: bitwise reproducible
    double v[kCols][kPer];
#pragma unroll
    for (int j = 0; j < kCols; ++j)
#pragma unroll
        for (int i = 0; i < kPer; ++i) {
            const int g = i * kThreads + (int)threadIdx.x;
            v[j][i] = (j < n_col && g < G) ? wpart[(size_t)j * G + g] : 0.0;
        }
#pragma unroll
    for (int j = 0; j < kCols; ++j) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < kPer; ++i) s += v[j][i];
        s = wave_sum(s);
        if (lane_id() == 0) s_col[j][wave_id()] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot[kCols];
#pragma unroll
        for (int j = 0; j < kCols; ++j) { tot[j] = s_col[j][0]; for (int w = 1; w < kWaves; ++w) tot[j] += s_col[j][w]; }
        const double W = tot[0], Q = tot[1];
        ctrl->M = lw_ref; ctrl->W = W; ctrl->Q = Q; ctrl->ess = W * W / Q;
        ctrl->log_z = lw_ref + log(W / n_pop);                                                 // SIS evidence: the mean weight
        ctrl->n_resampled = 0; ctrl->do_resample = 0; ctrl->cdf_lo = 0.0; ctrl->w_local = W; ctrl->scale = 1.0; ctrl->lw_after = 0.0;
        if (ess_trace) ess_trace[T - 1] = W * W / Q;
        if (resampled) resampled[T - 1] = 0;
        for (int t = 0; t < T; ++t) {
            if (normalise) {
                const double mean = tot[2 + t * K] / W;
                stats[t * K] = mean;
                stats[t * K + 1] = tot[2 + t * K + 1] / W - mean * mean;                      // raw_moment(2) - mean^2, empirical_distribution.hpp:78-81
            } else { stats[t * K] = tot[2 + t * K]; stats[t * K + 1] = tot[2 + t * K + 1]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// SMC step t (fused): [normalise generation t-1] -> [resample it -> ancestors] -> gather ancestor
// state -> sample x_t -> weight by observe t -> record predict + ancestor -> tile partial + linear
// weights.
//
// FUSED = true (populations of <= kFuseMaxTiles tiles): the normalisation of generation t-1 --
// what scan_partials_kernel does -- runs in the prologue of EVERY workgroup, redundantly, from
// the nb tile partials (32 B each, L2-resident): global max, W, ESS, the resampling decision and
// the whole tile-level CDF, kept in LDS.  All workgroups execute identical arithmetic in identical
// order, so they take identical decisions; workgroup 0 records them in ctrl for the host.  This
// removes one launch + one kernel boundary (~5 us) per step.  FUSED = false reads what
// scan_partials_kernel left in ctrl / bc / bf (large populations, sharded runs).
// The LDS copy of the tile-level CDF (16 B per tile) caps the resident workgroups per CU: measured crossover against the
// un-fused form at ~1.7e6 particles (profiles/r01_ab_notes.md), hence 1664 tiles rather than the 2048 the prologue could hold.
// ---------------------------------------------------------------------------------------------
constexpr int kFuseMaxTiles = 1664;

// blockIdx -> tile such that the workgroups with equal blockIdx % 8 (one XCD under round-robin dispatch; only speed depends on
// that) own a contiguous range of tiles.  A bijection of [0, nb) for every nb.  Only while the whole grid is resident at once
// (<= 2048 workgroups): beyond that the waves of workgroups in flight should sweep ONE region of HBM together, and the plain
// order is 2-4 % faster (profiles/r01_ab_notes.md).
__device__ __forceinline__ int xcd_contiguous_tile(int b, int nb)
{
#ifdef CPPROB_NO_XCD_SWIZZLE
    return b;
#else
    if (nb > 2048) return b;
    const int x = b & 7, j = b >> 3, q = nb >> 3, r = nb & 7;
    return x * q + (x < r ? x : r) + j;
#endif
}

template <class Model>
struct StepArgs {
    ModelParams mp; const double* obs; int t, T; int64_t n, ld;
    uint64_t seed, pid0;   // pid0: RNG id of local slot 0
    typename Model::store_t* values; int32_t* anc;
    const double* logw_prev; double* logw_next;
    const double* wrel_prev; double* wrel_next;
    const Partial* part_prev; Partial* part;
    const double* bc; const double* bf; int nb; StepCtrl* ctrl;
    const int32_t* anc_pre;   // RS_PRECOMPUTED: ancestors computed by multinomial_kernel
    double n_pop, ess_frac;   // FUSED: ESS test
    double* ess_trace; int32_t* resampled;
    int store_logw;           // 0: every step resamples (known on the host), so only the last step's log-weights are ever read
    int part_counts;          // with wrel_from_state in the fused form: tile partials between steps are packed per-value counts
    int wrel_from_state;      // table-weight model, every step resamples systematically: weights of generation t-1 = e_tab[t-1][state] (WeightSource)
    int64_t rs;               // row stride of values[] / anc[] (ld plus the immigrant annex)
    int row_w, row_r;         // rows of values[] this step writes / reads: t and t - 1, or the two rows of a filtering-only run in turn
    // exchange scope (exact global resampling over shards): outputs [imm_l0, imm_l1) of this shard descend from local
    // sources, the others from immigrants whose lineages sit in annex columns imm_col0, imm_col0 + 1, ... in output order
    int exchange;
    const int64_t* imm_l01;       // device: {l0, l1} of the plan the previous step's exchange left (exchange.hpp: ExchangePlan::l0, l1)
    const int64_t* annex_base;    // device [T + 1]: annex columns in use before each step's immigrants
};

// Tile partial when every particle's log-weight is lwa + (one of K table values): no exp, no fp64
// reduction -- per-value counts by ballot + popcount (scalar unit), one barrier.
//
// {sum e, sum e^2} of a tile from its per-value counts: the one place this arithmetic lives (producer and consumers must agree
// to the bit).
template <int K>
__device__ __forceinline__ void table_sums(const int (&c)[K], const double (&e_tab)[K], double& sm, double& q)
{
    sm = 0.0; q = 0.0;
#pragma unroll
    for (int s2 = 0; s2 < K; ++s2) {
        sm += (double)c[s2] * e_tab[s2];
        q += (double)c[s2] * (e_tab[s2] * e_tab[s2]);
    }
}
// A tile partial as packed counts (16 bits per table value): 8 bytes per tile instead of 24 for the step kernels' all-to-all
static_assert(kTile <= 65535, "packed counts hold 16 bits per value");
template <int K>
__device__ __forceinline__ uint64_t pack_counts(const int (&c)[K])
{
    static_assert(K <= 4, "four 16-bit counts per word");
    uint64_t w = 0;
#pragma unroll
    for (int s2 = 0; s2 < K; ++s2) w |= (uint64_t)(uint32_t)c[s2] << (16 * s2);
    return w;
}
template <int K>
__device__ __forceinline__ void unpack_counts(uint64_t w, int (&c)[K])
{
#pragma unroll
    for (int s2 = 0; s2 < K; ++s2) c[s2] = (int)((w >> (16 * s2)) & 0xFFFFu);
}

template <int K>
__device__ __forceinline__ void tile_partial_table(const int (&idx)[kPPT], const bool (&valid)[kPPT], const double (&e_tab)[K], double m_ref,
                                                   double (&e)[kPPT], Partial* __restrict__ part, int* s_cnt /* kWaves*K ints */, bool counts_out, int tile)
{
    int cnt[K];
#pragma unroll
    for (int s2 = 0; s2 < K; ++s2) cnt[s2] = 0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        e[k] = 0.0;
#pragma unroll
        for (int s2 = 0; s2 < K; ++s2) {
            const bool hit = valid[k] && idx[k] == s2;
            cnt[s2] += __popcll(__ballot(hit));
            if (hit) e[k] = e_tab[s2];
        }
    }
    if (lane_id() == 0) {
#pragma unroll
        for (int s2 = 0; s2 < K; ++s2) s_cnt[wave_id() * K + s2] = cnt[s2];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int c[K];
#pragma unroll
        for (int s2 = 0; s2 < K; ++s2) {
            c[s2] = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) c[s2] += s_cnt[w * K + s2];
        }
        if (counts_out) {
            reinterpret_cast<uint64_t*>(part)[tile] = pack_counts<K>(c);   // the next step kernel rebuilds {sum, sum of squares}
        } else {
            double sm, q;
            table_sums<K>(c, e_tab, sm, q);
            put_partial(part, (int)gridDim.x, tile, m_ref, sm, q);
        }
    }
}

// FUSED: 0 = read ctrl / bc / bf (scan_partials_kernel ran); otherwise the number of tile partials each lane
// holds in registers in the normalisation prologue (2, 4 or 8: populations of <= 512, 1024, 2048 tiles)
// (launch bounds: the 8-partials-per-lane form of continuous-weight models is held to 6 waves per SIMD = 80 VGPRs, so that the
//  1025..1664 workgroups it serves are all resident: linear_gaussian_1d 1.25e6 particles 2399 -> 2211 us per run; the table-weight
//  form spills at that limit and stays unconstrained -- profiles/r01_ab_notes.md)
// COUNTS: the tile partials between steps are packed per-value counts (table-weight models on an every-step schedule; a compile-
// time form because the {max, sum, sum of squares} prologue it replaces costs 14-24 VGPRs: 73 -> 59 at 4 partials per lane, 101 -> 77 at 8)
template <class Model, int RS, int FUSED, bool COUNTS = false>
__global__ __launch_bounds__(kThreads, (FUSED == 8 && Model::kWeightTable == 0) ? 6 : 1) void smc_step_kernel(StepArgs<Model> a)
{
    using V = typename Model::value_t;
    extern __shared__ __attribute__((aligned(16))) double s_dyn[];   // FUSED: bc[nb+1] then bf[nb]
    __shared__ AncestorLds L;
    __shared__ double s_scr[3 * kWaves];
    __shared__ int s_cnt[kWaves * 4];
    const int tid = threadIdx.x;
    // output tile of this workgroup: workgroups that share an XCD (blockIdx % 8, round-robin dispatch) take a CONTIGUOUS range
    // of tiles, so the neighbouring source tiles a workgroup reads were requested by its XCD's other workgroups too (one L2)
    const int bid = xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x);
    const int64_t j0 = (int64_t)bid * kTile + (int64_t)tid * kPPT;
    const int t = a.t;
    double* s_bc = s_dyn;
    double* s_bf = s_dyn + (a.nb + 1);
    CPH_STAMP(0);

    // independent of everything this step waits for: the random part of sample #t, and (fused form) the linear weights of
    // the source tile with this workgroup's own index -- both issued before the first memory round trip completes
    typename Model::Rand rnd[kPPT / 4];
#pragma unroll
    for (int q = 0; q < kPPT / 4; ++q) Model::draw4(a.seed, a.pid0 + (uint64_t)j0 + 4 * q, t, rnd[q]);
    double w_own[kPPT];
    lane_fill(w_own, 0.0);
    WeightSource<Model> ws;
    ws.wrel = a.wrel_prev; ws.states = a.values + (int64_t)a.row_r * a.rs; ws.n = a.n; ws.from_states = a.wrel_from_state != 0;
    if (Model::kWeightTable > 0 && ws.from_states && t > 0) {
        constexpr int K = WeightSource<Model>::K;
        double ll[K], mref;
        Model::weight_table(a.mp, t - 1, ll, ws.e, mref);
    }
    if (FUSED && RS == RS_SYSTEMATIC && t > 0) {
        ws.load(j0, w_own);
        int32_t neg[kPPT];
        lane_fill(neg, (int32_t)-1);
        store4(L.u.slot, (int64_t)tid * kPPT, neg);
    }

    bool resample = false;
    double u0 = 0.0, inv_stepw = 0.0, lwa = 0.0;
    if (t > 0) {
        if (FUSED) {
            // ---- normalise generation t-1 from its tile partials (every workgroup, identically) ----
            // each lane owns FUSED consecutive tiles: one coalesced vector load per partial array, no LDS round trip
            constexpr int kPer = FUSED > 0 ? FUSED : 1;
            const int pst = part_stride(a.nb);
            const double* pm = a.part_prev; const double* psum = a.part_prev + pst; const double* pq = a.part_prev + 2 * pst;
            const int c0 = tid * kPer;
            double rm[kPer], rs[kPer], rq[kPer];
            double M;
            if (Model::kWeightTable > 0 && COUNTS) {
                // packed counts (8 B per tile): every tile shares the reference lwa + the step's largest table value, so
                // there is no max to reduce, and {sum, sum of squares} follow from the counts
                constexpr int K = Model::kWeightTable > 0 ? Model::kWeightTable : 1;
                const uint64_t* pc = reinterpret_cast<const uint64_t*>(a.part_prev);
                uint64_t wv[kPer];
#pragma unroll
                for (int i = 0; i < kPer; i += 2) {
                    const ulonglong2 v2 = *reinterpret_cast<const ulonglong2*>(pc + c0 + i);
                    wv[i] = v2.x; wv[i + 1] = v2.y;
                }
                double ll[K], et[K], mref;
                Model::weight_table(a.mp, t - 1, ll, et, mref);
                M = mref;
#pragma unroll
                for (int i = 0; i < kPer; ++i) {
                    int c[K];
                    unpack_counts<K>(wv[i], c);
                    table_sums<K>(c, et, rs[i], rq[i]);
                    rm[i] = M;
                    if (c0 + i >= a.nb) { rm[i] = -INFINITY; rs[i] = 0.0; rq[i] = 0.0; }
                }
            } else {
#pragma unroll
                for (int i = 0; i < kPer; i += 2) {                      // 16-B loads (the arrays are padded past nb)
                    const double2 vm = *reinterpret_cast<const double2*>(pm + c0 + i);
                    const double2 vs = *reinterpret_cast<const double2*>(psum + c0 + i);
                    const double2 vq = *reinterpret_cast<const double2*>(pq + c0 + i);
                    rm[i] = vm.x; rm[i + 1] = vm.y; rs[i] = vs.x; rs[i + 1] = vs.y; rq[i] = vq.x; rq[i + 1] = vq.y;
                }
                double m = -INFINITY;
#pragma unroll
                for (int i = 0; i < kPer; ++i) {
                    if (c0 + i >= a.nb) { rm[i] = -INFINITY; rs[i] = 0.0; rq[i] = 0.0; }
                    m = fmax(m, rm[i]);
                }
                M = block_max(m, s_scr);
            }
            double Q = 0.0, S = 0.0, ev[kPer];
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                double e = 1.0;
                if (rm[i] != M) e = rescale_factor(rm[i], M, Model::kWeightTable == 0);   // (table-weight models: mostly one shared reference)
                ev[i] = e;
                rs[i] *= e;                                          // tile mass
                S += rs[i];
                Q += rq[i] * (e * e);
            }
            double W;
            double run = block_sum_and_excl_scan(Q, S, s_scr + kWaves, &W);      // one barrier for both
#pragma unroll
            for (int i = 0; i < kPer; ++i) {
                if (c0 + i < a.nb) { s_bc[c0 + i] = run; s_bf[c0 + i] = ev[i]; }
                run += rs[i];
            }
            if (tid == 0) s_bc[a.nb] = W;
            const double ess = W * W / Q;
            resample = ess < a.ess_frac * a.n_pop;                            // ESS test, thesis p.37 (t-1 is never the last step here)
            inv_stepw = (double)a.n / W;
            lwa = 0.0;
            u0 = a.ctrl->u0_pp[t & 1];                                        // left there by workgroup 0 of step t-1
            if (bid == 0 && tid == 0) {                                       // bookkeeping for the host
                StepCtrl* c = a.ctrl;
                c->M = M; c->W = W; c->Q = Q; c->ess = ess; c->do_resample = resample ? 1 : 0;
                c->cdf_lo = 0.0; c->w_local = W; c->scale = 1.0; c->u0 = u0; c->inv_stepw = inv_stepw; c->lw_after = 0.0;
                double lz = (t == 1) ? 0.0 : c->log_z;
                int nr = (t == 1) ? 0 : c->n_resampled;
                if (resample) { lz += M + log(W / a.n_pop); nr += 1; }
                c->log_z = lz; c->n_resampled = nr;
                if (a.ess_trace) a.ess_trace[t - 1] = ess;
                if (a.resampled) a.resampled[t - 1] = resample ? 1 : 0;
            }
            __syncthreads();
        } else {
            resample = a.ctrl->do_resample != 0;                             // workgroup-uniform (scalar load)
            u0 = a.ctrl->u0; inv_stepw = a.ctrl->inv_stepw; lwa = a.ctrl->lw_after;
        }
    }

    CPH_STAMP(1);
    int32_t anc[kPPT]; double lw[kPPT];
    if (!resample) {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) { anc[k] = (int32_t)(j0 + k); lw[k] = 0.0; }
        if (t > 0) load4(a.logw_prev, j0, lw);               // weights carry over when no resampling happened
    } else {
        if (RS == RS_PRECOMPUTED) {
            load4(a.anc_pre, j0, anc);
        } else {
            const int64_t rem = a.n - (int64_t)bid * kTile;
            AncestorIn in;
            in.wrel = a.wrel_prev; in.nb = a.nb; in.n_in = a.n;
            if (FUSED) { in.bc = s_bc; in.bf = s_bf; in.bc_in_lds = 1; } else { in.bc = a.bc; in.bf = a.bf; in.bc_in_lds = 0; }
            in.W = in.bc[a.nb]; in.scale = 1.0; in.cdf_lo = 0.0; in.u0 = u0; in.inv_stepw = inv_stepw;
            in.seed = a.seed; in.step = (uint64_t)t; in.gj_tile0 = (uint64_t)bid * kTile; in.n_total_out = (uint64_t)a.n;
            in.id0 = a.pid0;
            in.n_valid_tile = rem < kTile ? (int)rem : kTile;
            in.guess = bid;
            in.g_end = INFINITY;
            if (!FUSED && a.exchange) {                       // positions and outputs in global terms
                in.scale = a.ctrl->scale; in.cdf_lo = a.ctrl->cdf_lo; in.inv_stepw = a.ctrl->inv_global; in.g_end = a.ctrl->g_end;
                in.gj_tile0 = a.pid0 + (uint64_t)bid * kTile; in.n_total_out = (uint64_t)a.n_pop; in.guess = -1;
            }
            if (FUSED && RS == RS_SYSTEMATIC) ancestors_systematic_fused(ws, s_bc, s_bf, a.nb, u0, inv_stepw, in.n_valid_tile, w_own, anc, L, bid);
            else find_ancestors<RS>(in, anc, L, ws);
        }
#pragma unroll
        for (int k = 0; k < kPPT; ++k) lw[k] = lwa;          // equal weights after resampling (the shard's mass share)
        if (!FUSED && a.exchange) {
            const int64_t l0 = a.imm_l01[0], l1 = a.imm_l01[1], col0 = a.ld + a.annex_base[t - 1];
#pragma unroll
            for (int k = 0; k < kPPT; ++k) {
                const int64_t j = j0 + k;
                if (j < l0) anc[k] = (int32_t)(col0 + j);
                else if (j >= l1 && j < a.n) anc[k] = (int32_t)(col0 + j - (l1 - l0));
            }
        }
    }

    CPH_STAMP(8);
    V prev[kPPT], x[kPPT];
    const typename Model::store_t* prev_row = a.values + (int64_t)a.row_r * a.rs;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) prev[k] = t > 0 ? static_cast<V>(prev_row[anc[k]]) : V(0);                 // ancestor's state (sorted gather)
#pragma unroll
    for (int q = 0; q < kPPT / 4; ++q)                                                        // sample #t
        Model::apply4(a.mp, t, rnd[q], reinterpret_cast<const V(&)[4]>(prev[4 * q]), reinterpret_cast<V(&)[4]>(x[4 * q]));
    CPH_STAMP(9);
    bool valid[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) valid[k] = j0 + k < a.n;
    {
        typename Model::store_t xs[kPPT];
#pragma unroll
        for (int k = 0; k < kPPT; ++k) xs[k] = static_cast<typename Model::store_t>(x[k]);
        store4(a.values + (int64_t)a.row_w * a.rs, j0, xs);                            // predict #t
    }
    if (a.anc) store4_write_through(a.anc + (int64_t)t * a.rs, j0, anc);                      // (a filtering-only run keeps no ancestors)
    double e[kPPT];
    const bool fresh = (t == 0) || resample;                                                  // every particle starts the step at log-weight lwa
    if (Model::kWeightTable > 0 && fresh) {
        // observe #t for models whose incremental weight takes one of K values: table look-ups only
        constexpr int K = Model::kWeightTable > 0 ? Model::kWeightTable : 1;
        double ll[K], et[K], mref;
        Model::weight_table(a.mp, t, ll, et, mref);
        int idx[kPPT];
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            idx[k] = Model::weight_index(x[k]);
            double l = ll[0];
#pragma unroll
            for (int s2 = 1; s2 < K; ++s2) l = idx[k] == s2 ? ll[s2] : l;
            lw[k] = valid[k] ? lw[k] + l : -INFINITY;
        }
        tile_partial_table<K>(idx, valid, et, (t == 0 ? 0.0 : lwa) + mref, e, a.part, s_cnt, FUSED && COUNTS && t + 1 < a.T, bid);
    } else {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            lw[k] += Model::loglik(a.mp, x[k], t, a.obs);                                     // observe #t
            if (!valid[k]) lw[k] = -INFINITY;                                                 // padding slots
        }
        tile_partial(lw, e, a.part, s_scr, bid, Model::kWeightTable == 0);
    }
    CPH_STAMP(10);
    if (a.store_logw || t + 1 == a.T) store4(a.logw_next, j0, lw);
    if (!a.wrel_from_state || t + 1 == a.T) store4(a.wrel_next, j0, e);   // otherwise the next step reads the states
    if (FUSED && bid == 0 && tid == 0 && t + 1 < a.T) {              // systematic offset of the resampling before step t+1
        const u32x4 r = draw_block(a.seed, 0, kResampleDrawBase + (uint64_t)(t + 1));
        a.ctrl->u0_pp[(t + 1) & 1] = u01_53(r.x, r.y);
    }
    CPH_STAMP(11);
}

// ---------------------------------------------------------------------------------------------
// Posterior read-out: StatsPrinter / EmpiricalDistribution over the final particles' full
// traces (reference stats_printer.hpp:88-120: the k-th hit of a predict address in a trace
// goes to the k-th distribution; empirical_distribution.hpp:30-40,52-81).  A final particle's
// trace is its ancestral line, so each lane walks anc[] backwards from its final slot and
// accumulates W_i * f(x_t) per step; ancestors are sorted, so the walk stays coalesced and
// collapses onto the surviving lineages (L2 hits).  Final weights are wrel * bf[tile]: no exp.
// ---------------------------------------------------------------------------------------------
template <class Model>
struct SmoothArgs {
    const typename Model::store_t* values; const int32_t* anc; const double* wrel; const double* bf; const StepCtrl* ctrl;
    const int32_t* resampled; int T; int64_t n, ld, rs; int identity;   // rs: row stride of values[] / anc[]
    double* stats_part;   // [T * kStats][gridDim.x]
    typename Model::value_t* paths;   // optional [T][ld]: materialised traces (dump / tests)
    const struct RemoteStores* rem;   // exchange scope with remote lineages (device pointer), or nullptr
};

template <class Model> struct SmoothArgs;
struct RemoteStores;
#ifndef CPPROB_SMOOTH_TILES
#define CPPROB_SMOOTH_TILES 2
#endif
constexpr int kSmoothTiles = CPPROB_SMOOTH_TILES;      // tiles a workgroup of the lineage walk follows at a time
#ifndef CPPROB_SMOOTH_ROWS
#define CPPROB_SMOOTH_ROWS 2
#endif
constexpr int kSmoothRows = CPPROB_SMOOTH_ROWS;        // rows of the store fetched together between two resamplings
constexpr int kSmoothMaxT = 2048;                       // (the read-out's LDS accumulators bound T x statistics per hit by 2048: cpprob_hip_infer_begin)
template <class Model, class WeightOf>
__device__ __forceinline__ void smooth_body_remote(const SmoothArgs<Model>& a, double* s_stat, WeightOf weight_of, const RemoteStores* __restrict__ rem);

// Remote lineages (exchange scope over peer-mapped stores): a lineage that reaches an annex column continues in the particle store
// of the rank the particle came from -- same generation, the slot recorded at its arrival.  One entry per rank, as THIS device
// addresses that rank's memory.
struct RemoteStores {
    const void* values[64]; const int32_t* anc[64]; const int64_t* origin[64];
    int64_t rs[64], ld[64];
    int world, rank;
    uint32_t* trace[2][64];              // trace words of every rank by the step's parity (trace_words.hpp), nullptr where not in use
};

// weight_of(tile, i) = the final weight of slot i (zero for padding slots), in the units finalize_kernel divides by.
template <class Model, class WeightOf>
__device__ __forceinline__ void smooth_body(const SmoothArgs<Model>& a, double* s_stat, WeightOf weight_of)
{
    if (a.rem) { smooth_body_remote<Model>(a, s_stat, weight_of, a.rem); return; }       // (workgroup-uniform)
    using V = typename Model::value_t;
    constexpr int K = Model::kStats;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int TK = a.T * K;
    for (int i = tid; i < kWaves * TK; i += kThreads) s_stat[i] = 0.0;
    // run[t]: how many rows, from t downwards, the lineages stay in their slots (no resampling between them)
    __shared__ uint16_t s_run[kSmoothMaxT];
    if (tid == 0) {
        int run = 0;
        for (int t = 0; t < a.T; ++t) { run = (t > 0 && !a.identity && a.resampled[t - 1]) ? 1 : run + 1; s_run[t] = (uint16_t)run; }
    }
    __syncthreads();
    const int64_t ntiles = (a.n + kTile - 1) / kTile;
    // (one XCD's workgroups walk neighbouring tiles: their lineages converge on the same ancestor rows)
    // Two tiles at a time (kSmoothTiles): eight lineages a lane -- a hop is a dependent gather, and what the walk is short of is loads
    // in flight (wait_frac 0.6-0.8 at four a lane: profiles/r04_pmc_traffic.json), not registers; the per-hop wavefront reduction is
    // shared by both tiles.
    auto walk = [&](auto tiles_tag) {
    constexpr int NT = decltype(tiles_tag)::value;
    for (int64_t tile0 = ntiles <= 2048 ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x; tile0 < ntiles; tile0 += (int64_t)NT * gridDim.x) {
        constexpr int L = NT * kPPT;
        int32_t idx[L]; double w[L];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int64_t tile = tile0 + (int64_t)(k / kPPT) * gridDim.x;
            const int64_t i = tile * kTile + (int64_t)(k % kPPT) * kThreads + tid;   // lane-strided: coalesced first touch
            const bool on = tile < ntiles;
            idx[k] = on ? (int32_t)i : 0;
            w[k] = on ? weight_of(tile, i) : 0.0;
        }
        // Rows between two resamplings are read at the SAME slots: up to kSmoothRows of them are fetched together (independent gathers
        // in flight instead of one dependent round trip a row: ESS-triggered schedules resample after a quarter of their steps), each
        // row's sums taken exactly as before -- the same numbers in the same order.
        auto rows = [&](auto rows_tag, int t_hi) {
            constexpr int U = decltype(rows_tag)::value;
            typename Model::store_t raw[U][L];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const typename Model::store_t* row = a.values + (int64_t)(t_hi - u) * a.rs;
#pragma unroll
                for (int k = 0; k < L; ++k) raw[u][k] = row[idx[k]];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t_hi - u;
                double acc[K];
#pragma unroll
                for (int j = 0; j < K; ++j) acc[j] = 0.0;
#pragma unroll
                for (int k = 0; k < L; ++k) {
                    const V x = static_cast<V>(raw[u][k]);
                    Model::accumulate(x, w[k], acc);
                    const int64_t tile = tile0 + (int64_t)(k / kPPT) * gridDim.x;
                    if (a.paths && tile < ntiles) a.paths[(int64_t)t * a.ld + tile * kTile + (int64_t)(k % kPPT) * kThreads + tid] = x;
                }
#pragma unroll
                for (int j = 0; j < K; ++j) acc[j] = wave_sum(acc[j]);
                if (lane == 0) {
#pragma unroll
                    for (int j = 0; j < K; ++j) s_stat[wv * TK + t * K + j] += acc[j];
                }
            }
        };
        int t = a.T - 1;
        while (t >= 0) {
            const int r = (int)s_run[t];                                  // rows t, t-1, .., t-r+1 share the lineages' slots
            int u = 0;
            // (measured, us per launch, rows 1 / 2 / 4: linear_gaussian_1d<100> at 1e7 1166 / 1092 / 1223; hmm<128> at 1.25e7 1179 / 1294 / 1557 --
            //  one-byte states gain nothing from more gathers in flight and lose the registers: profiles/r05_notes.md)
            constexpr int kRows = sizeof(typename Model::store_t) == 1 ? 1 : kSmoothRows;
            for (; u + kRows <= r && kRows > 1; u += kRows) rows(std::integral_constant<int, kRows>{}, t - u);
            for (; u < r; ++u) rows(std::integral_constant<int, 1>{}, t - u);
            t -= r;
            if (t >= 0) {                                                 // (generation t was resampled: the lineages hop)
                const int32_t* arow = a.anc + (int64_t)(t + 1) * a.rs;
#pragma unroll
                for (int k = 0; k < L; ++k) idx[k] = arow[idx[k]];
            }
        }
    }
    };
    // (pairs where a workgroup has several tiles to follow and a hop moves little -- one-byte states: measured on hmm<128> at
    //  1.25e7 particles 1.73 -> 1.40 ms; 8-byte values are bound by the bytes they gather: linear_gaussian_1d<100> at 1e7 1.14 -> 1.32 ms)
    if (kSmoothTiles > 1 && sizeof(typename Model::store_t) == 1 && ntiles > (int64_t)gridDim.x) walk(std::integral_constant<int, kSmoothTiles>{});
    else walk(std::integral_constant<int, 1>{});
    __syncthreads();
    for (int i = tid; i < TK; i += kThreads) {
        double s = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kWaves; ++w2) s += s_stat[w2 * TK + i];
        a.stats_part[(int64_t)i * gridDim.x + blockIdx.x] = s;
    }
}

template <class Model, class WeightOf>
__device__ __forceinline__ void smooth_body_remote(const SmoothArgs<Model>& a, double* s_stat, WeightOf weight_of, const RemoteStores* __restrict__ rem)
{
    using V = typename Model::value_t;
    using S = typename Model::store_t;
    constexpr int K = Model::kStats;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int TK = a.T * K;
    for (int i = tid; i < kWaves * TK; i += kThreads) s_stat[i] = 0.0;
    __syncthreads();
    const int64_t ntiles = (a.n + kTile - 1) / kTile;
    const int me = rem->rank;
    // (smooth_body's tile order and pairing: the workgroups' partial sums are then the very same numbers whichever way the lineages travelled)
    auto walk = [&](auto tiles_tag) {
    constexpr int NT = decltype(tiles_tag)::value;
    for (int64_t tile0 = ntiles <= 2048 ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x; tile0 < ntiles; tile0 += (int64_t)NT * gridDim.x) {
        constexpr int L = NT * kPPT;
        int32_t idx[L], rk[L]; double w[L];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            const int64_t tile = tile0 + (int64_t)(k / kPPT) * gridDim.x;
            const int64_t i = tile * kTile + (int64_t)(k % kPPT) * kThreads + tid;
            const bool on = tile < ntiles;
            idx[k] = on ? (int32_t)i : 0; rk[k] = me;
            w[k] = on ? weight_of(tile, i) : 0.0;
        }
        for (int t = a.T - 1; t >= 0; --t) {
            double acc[K];
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = 0.0;
            const bool hop = t > 0 && !a.identity && a.resampled[t - 1];
#pragma unroll
            for (int k = 0; k < L; ++k) {
                const int r = rk[k];
                // (nearly every hop stays on this rank -- O(sqrt N) lineages ever cross: its store comes from the launch's own arguments,
                //  not through the table of the ranks' stores, which would put a second dependent load on every hop)
                const bool here = r == me;
                const int64_t rs_r = here ? a.rs : rem->rs[r], ld_r = here ? a.ld : rem->ld[r];
                const S* vals = here ? a.values : static_cast<const S*>(rem->values[r]);
                const int32_t* ancs = here ? a.anc : rem->anc[r];
                const int64_t at = (int64_t)t * rs_r + idx[k];
                const V x = static_cast<V>(vals[at]);
                Model::accumulate(x, w[k], acc);
                const int64_t tile = tile0 + (int64_t)(k / kPPT) * gridDim.x;
                if (a.paths && tile < ntiles) a.paths[(int64_t)t * a.ld + tile * kTile + (int64_t)(k % kPPT) * kThreads + tid] = x;
                if (hop) {
                    // (a run whose transport overflowed is repeated, but its read-out still runs: whatever the tables hold then, the
                    //  walk stays inside the stores)
                    const int64_t p = min(max((int64_t)ancs[at], (int64_t)0), rs_r - 1);
                    if (p >= ld_r) {                                      // an immigrant of that rank: its history sits where it came from
                        const int64_t o = rem->origin[r][p - ld_r];
                        const int ro = min(max((int)(o >> 32), 0), rem->world - 1);
                        rk[k] = ro; idx[k] = (int32_t)min((int64_t)(uint32_t)o, rem->rs[ro] - 1);
                    } else idx[k] = (int32_t)p;
                }
            }
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = wave_sum(acc[j]);
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < K; ++j) s_stat[wv * TK + t * K + j] += acc[j];
            }
        }
    }
    };
    if (kSmoothTiles > 1 && sizeof(typename Model::store_t) == 1 && ntiles > (int64_t)gridDim.x) walk(std::integral_constant<int, kSmoothTiles>{});
    else walk(std::integral_constant<int, 1>{});
    __syncthreads();
    for (int i = tid; i < TK; i += kThreads) {
        double s = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kWaves; ++w2) s += s_stat[w2 * TK + i];
        a.stats_part[(int64_t)i * gridDim.x + blockIdx.x] = s;
    }
}

template <class Model>
__global__ __launch_bounds__(kThreads) void smooth_kernel(SmoothArgs<Model> a)
{
    extern __shared__ __attribute__((aligned(16))) double s_stat[];   // [kWaves][T*K]
    const double scale = a.ctrl->scale;
    smooth_body<Model>(a, s_stat, [&](int64_t tile, int64_t i) { return a.wrel[i] * (a.bf[tile] * scale); });   // padding slots: wrel = 0
}

// ---------------------------------------------------------------------------------------------
// Filtering-only runs (keep_history = 0): the particle store holds two rows and no ancestors, and predict hit t's
// statistics are those of generation t under ITS weights -- sum_i w_t,i f(x_t,i) / sum_i w_t,i -- taken right after
// step t.  One launch per step leaves, per workgroup, {reference m, sum e, sum e f_0 .. f_{K-1}} with e = exp(logw - m);
// one launch per run combines the workgroups in index order and normalises (filter_finalize_kernel).
// fpart layout: [T][K + 2][grid].
// ---------------------------------------------------------------------------------------------
template <class Model>
__global__ __launch_bounds__(kThreads) void filter_partials_kernel(const typename Model::store_t* __restrict__ row, const double* __restrict__ logw,
                                                                    int64_t n, double* __restrict__ fpart_t)
{
    using V = typename Model::value_t;
    constexpr int K = Model::kStats;
    __shared__ double s_scr[kWaves];
    __shared__ double s_sum[(K + 1) * kWaves];
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int64_t ntiles = (n + kTile - 1) / kTile;
    double M = -INFINITY, S = 0.0;                      // running reference; thread j <= K carries sum j (0: sum e, 1 + k: sum e f_k)
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t j0 = tile * kTile + (int64_t)tid * kPPT;
        double lw[kPPT]; V x[kPPT];
#pragma unroll
        for (int k = 0; k < kPPT; ++k) { lw[k] = j0 + k < n ? logw[j0 + k] : -INFINITY; x[k] = static_cast<V>(row[j0 + k]); }
        const double m = block_max(fmax(fmax(lw[0], lw[1]), fmax(lw[2], lw[3])), s_scr);
        double acc[K + 1];
#pragma unroll
        for (int j = 0; j <= K; ++j) acc[j] = 0.0;
        if (m != -INFINITY) {
#pragma unroll
            for (int k = 0; k < kPPT; ++k) {
                const double e = lw[k] == -INFINITY ? 0.0 : exp(lw[k] - m);
                acc[0] += e;
                Model::accumulate(x[k], e, reinterpret_cast<double(&)[K]>(acc[1]));
            }
        }
#pragma unroll
        for (int j = 0; j <= K; ++j) { const double w = wave_sum(acc[j]); if (lane == 0) s_sum[j * kWaves + wv] = w; }
        __syncthreads();
        if (tid <= K && m != -INFINITY) {
            double t = 0.0;
#pragma unroll
            for (int w2 = 0; w2 < kWaves; ++w2) t += s_sum[tid * kWaves + w2];
            const double nm = fmax(M, m);
            S = S * (M == -INFINITY ? 0.0 : exp(M - nm)) + t * exp(m - nm);
            M = nm;
        }
        __syncthreads();
    }
    if (tid <= K) fpart_t[(int64_t)(tid + 1) * gridDim.x + blockIdx.x] = S;
    if (tid == 0) fpart_t[blockIdx.x] = M;
}

// One workgroup per predict hit t: StatsPrinter's numbers from the workgroups' partials (fixed order: bitwise reproducible).
// normalise = 0 (a shard of a joint population): raw sums into stats, this shard's mass of generation t into masses[t].
__global__ __launch_bounds__(kThreads) void filter_finalize_kernel(const double* __restrict__ fpart, int grid, int K, int is_int, double* __restrict__ stats,
                                                                    int normalise, double* __restrict__ masses)
{
    __shared__ double s_scr[kWaves];
    __shared__ double s_out[9];
    const int t = blockIdx.x;
    const double* base = fpart + (int64_t)t * (K + 2) * grid;
    double m = -INFINITY;
    for (int g = threadIdx.x; g < grid; g += kThreads) m = fmax(m, base[g]);
    const double M = block_max(m, s_scr);
    for (int j = 0; j <= K; ++j) {
        double s = 0.0;
        for (int g = threadIdx.x; g < grid; g += kThreads) { const double mg = base[g]; if (mg != -INFINITY) s += base[(int64_t)(j + 1) * grid + g] * exp(mg - M); }
        __syncthreads();
        s = block_sum(s, s_scr);
        if (threadIdx.x == 0) s_out[j] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double W = s_out[0];
        if (!normalise) { for (int j = 0; j < K; ++j) stats[t * K + j] = s_out[j + 1]; if (masses) masses[t] = W; }
        else if (is_int) { for (int j = 0; j < K; ++j) stats[t * K + j] = s_out[j + 1] / W; }
        else { const double mean = s_out[1] / W; stats[t * K] = mean; stats[t * K + 1] = s_out[2] / W - mean * mean; }
    }
}

// Sums the per-workgroup partial statistics (layout [T*K][grid]) in a fixed order (bitwise
// reproducible) and normalises: real -> {mean, raw2 - mean^2}; int -> probabilities.
// One workgroup per predict hit t.  normalise = 0 leaves raw weighted sums (sharded runs
// all-reduce them before normalising).
__global__ __launch_bounds__(kThreads) void finalize_kernel(const double* __restrict__ stats_part, int grid, int T, int K, int is_int,
                                                             const StepCtrl* __restrict__ ctrl, double* __restrict__ stats, int normalise)
{
    __shared__ double s_scr[8][kWaves];
    __shared__ double s_out[8];
    const int t = blockIdx.x;
    const double W = normalise ? ctrl->W : 1.0;
    // (the K columns' loads travel together -- unconditional, from a valid column -- instead of one round trip per column; every
    //  thread's partial sums and the block's sums are those of one column after the other)
    double s[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s[j] = 0.0;
    for (int g = threadIdx.x; g < grid; g += kThreads) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = stats_part[(int64_t)(t * K + (j < K ? j : 0)) * grid + g];
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += v[j];
    }
    for (int j = 0; j < K; ++j) {
        double sj = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) sj = q == j ? s[q] : sj;
        sj = block_sum(sj, s_scr[j]);
        if (threadIdx.x == 0) s_out[j] = sj;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (is_int || !normalise) {
            for (int j = 0; j < K; ++j) stats[t * K + j] = s_out[j] / W;
        } else {
            const double mean = s_out[0] / W;
            stats[t * K] = mean;
            stats[t * K + 1] = s_out[1] / W - mean * mean;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Exchange scope: lineages that leave / enter a shard.
//   extract_lineages_kernel : record k = the trace x_0 .. x_{len-1} of the current-generation particle src[k]
//                             (walks anc[] like smooth_kernel; annex columns are ordinary columns to it)
//   annex_lineages_kernel   : record k becomes annex column col0 + k of rows 0 .. len-1, with identity ancestors,
//                             so that neither the step kernel's gather nor any lineage walk needs to know that the
//                             particle came from another GPU
// Records are [count][len], record-major (one contiguous block per destination rank for the all-to-all).
// ---------------------------------------------------------------------------------------------
template <class S, class V>
__global__ void extract_lineages_kernel(const S* __restrict__ values, const int32_t* __restrict__ anc, int64_t rs, const int32_t* __restrict__ resampled,
                                        int len, const int32_t* __restrict__ src, int64_t count, V* __restrict__ rec)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count) return;
    int32_t idx = src[k];
    for (int t = len - 1; t >= 0; --t) {
        rec[k * len + t] = static_cast<V>(values[(int64_t)t * rs + idx]);
        if (t > 0 && resampled[t - 1]) idx = anc[(int64_t)t * rs + idx];
    }
}

template <class S, class V>
__global__ void annex_lineages_kernel(const V* __restrict__ rec, int64_t count, int len, S* __restrict__ values, int32_t* __restrict__ anc, int64_t rs,
                                      int64_t col0)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count * len) return;
    const int64_t k = i / len; const int t = (int)(i - k * len);
    const int64_t col = col0 + k;
    values[(int64_t)t * rs + col] = static_cast<S>(rec[i]);
    anc[(int64_t)t * rs + col] = (int32_t)col;
}

// Traces from per-step records (cpprob_hip_lineage_gather): lane i walks final particle i's ancestral line; first_row[t] .. first_row[t+1]
// are the rows recorded in generation t's slots.
template <class V>
__global__ void lineage_gather_kernel(const int32_t* __restrict__ anc, const int32_t* __restrict__ resampled, int T, int64_t n, const V* __restrict__ cols,
                                      const int32_t* __restrict__ first_row, V* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t idx = i;
    for (int t = T - 1; t >= 0; --t) {
        for (int h = first_row[t]; h < first_row[t + 1]; ++h) out[(int64_t)h * n + i] = cols[(int64_t)h * n + idx];
        if (t > 0 && resampled[t - 1]) idx = anc[(int64_t)t * n + idx];
    }
}

// rows of a narrow particle store widened to the model's value type ([T][rs] -> [T][ld]): copy-out through the C ABI
template <class S, class V>
__global__ void widen_rows_kernel(const S* __restrict__ src, int64_t rs, int T, int64_t n, int64_t ld, V* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    for (int t = 0; t < T; ++t) dst[(int64_t)t * ld + i] = static_cast<V>(src[(int64_t)t * rs + i]);
}

// {log_evidence, ess, log_norm, max_logw, stats[n_stats]} of the finished run into one device buffer
__global__ void pack_results_kernel(const StepCtrl* __restrict__ ctrl, const double* __restrict__ stats, int n_stats, double* __restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { out[0] = ctrl->log_z; out[1] = ctrl->ess; out[2] = ctrl->M + log(ctrl->W); out[3] = ctrl->M; }
    if (i < n_stats) out[4 + i] = stats[i];
}

// ---------------------------------------------------------------------------------------------
// Elementwise building blocks (unit-parity surface)
// ---------------------------------------------------------------------------------------------
__global__ void philox_blocks_kernel(uint64_t seed, uint64_t group0, uint64_t draw, int64_t n, uint32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 r = draw_block(seed, group0 + (uint64_t)i, draw);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}

__global__ void draw_normal_kernel(uint64_t seed, uint64_t pid0, uint64_t draw, double mean, double sigma, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = draw_normal(seed, pid0 + (uint64_t)i, draw, mean, sigma);
}

__global__ void draw_smallint_kernel(uint64_t seed, uint64_t pid0, uint64_t draw, int64_t a, int64_t b, int64_t n, int32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)((int64_t)draw_smallint(seed, pid0 + (uint64_t)i, draw, 0, (uint64_t)(b - a)) + a);
}

struct DiscreteW { double w[8]; int k; };
__global__ void draw_discrete_kernel(uint64_t seed, uint64_t pid0, uint64_t draw, DiscreteW dw, int64_t n, int32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)discrete_from_u_dyn(u01_32(draw_word(seed, pid0 + (uint64_t)i, draw)), dw.w, dw.k);
}

__global__ void draw_uniform_real_kernel(uint64_t seed, uint64_t pid0, uint64_t draw, double a, double b, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = draw_uniform_real(seed, pid0 + (uint64_t)i, draw, a, b);
}

__global__ void draw_poisson_kernel(uint64_t seed, uint64_t pid0, uint64_t draw, double mean, int64_t n, int32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int32_t)draw_poisson(seed, pid0 + (uint64_t)i, draw, mean);
}

__global__ void logpdf_normal_kernel(const double* __restrict__ x, const double* __restrict__ mean, const double* __restrict__ sigma, int64_t n,
                                     double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = normal_logpdf(x[i], mean[i], sigma[i]);
}
__global__ void logpdf_uniform_real_kernel(const double* __restrict__ x, const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                           double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = uniform_real_logpdf(x[i], a[i], b[i]);
}
__global__ void logpdf_poisson_kernel(const int32_t* __restrict__ x, const double* __restrict__ l, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = poisson_logpdf(x[i], l[i]);
}
__global__ void logpdf_smallint_kernel(const int32_t* __restrict__ x, int64_t a, int64_t b, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = uniform_smallint_logpdf(x[i], a, b);
}
__global__ void logpdf_discrete_kernel(const int32_t* __restrict__ x, DiscreteW dw, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = discrete_logpdf(x[i], dw.w, dw.k);
}

// the lean elementary functions of cpprob/detail/fastmath.hpp, elementwise (which: 0 log01, 1 sincospi02 -> out0 = sin, out1 = cos, 2 exp_nonpos)
__global__ void fastmath_kernel(int which, const double* __restrict__ x, int64_t n, double* __restrict__ out0, double* __restrict__ out1)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (which == 0) out0[i] = log01(x[i]);
    else if (which == 1) { double sn, cs; sincospi02(x[i], sn, cs); out0[i] = sn; out1[i] = cs; }
    else out0[i] = exp_nonpos(x[i]);
}

template <class T>
__global__ void gather_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, T* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// Copy a user column into a tile-padded scratch column (building blocks operate on padded arrays).
template <class T>
__global__ void pad_copy_kernel(const T* __restrict__ src, int64_t n, int64_t ld, T* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < ld) dst[i] = i < n ? src[i] : T(0);
}

// The same for several columns at once: column k of src (k * src_stride) into row k of dst (k * ld).  grid = (ceil(ld / 256), columns).
template <class T>
__global__ void pad_copy_cols_kernel(const T* __restrict__ src, int64_t n, int64_t src_stride, int64_t ld, T* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t k = blockIdx.y;
    if (i < ld) dst[k * ld + i] = i < n ? src[k * src_stride + i] : T(0);
}

// Weighted moments / histogram of one column against one log-weight array (EmpiricalDistribution
// on device): the column is a 1-step "trace", so this is smooth_kernel with T = 1.
struct ColumnReal { using value_t = double; using store_t = double; static constexpr int kStats = 2; static constexpr bool kBins = false;
    __device__ static __forceinline__ void accumulate(double x, double w, double (&acc)[2]) { acc[0] += w * x; acc[1] += w * (x * x); }
    __device__ static __forceinline__ bool holds(const double (&)[kPPT], int) { return true; } };
struct ColumnInt8 { using value_t = int32_t; using store_t = int32_t; static constexpr int kStats = 8; static constexpr bool kBins = true;
    __device__ static __forceinline__ void accumulate(int32_t x, double w, double (&acc)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] += x == s ? w : 0.0; }
    __device__ static __forceinline__ bool holds(const int32_t (&x)[kPPT], int s) { return x[0] == s || x[1] == s || x[2] == s || x[3] == s; } };
static_assert(kPPT == 4, "ColumnInt8::holds names its particles");

// Statistics of per-step records along the lineages (cpprob_hip_lineage_moments / _hist): lineage_gather_kernel's walk with the
// columns' statistics taken on the way, in smooth_body's tile order and with its sums -- the numbers are those of gathering the
// traces first and reading the gathered columns (cpprob_hip_weighted_*_columns), without the traces' round trip through memory.
template <class Col>
struct LineageStatsArgs {
    const int32_t* anc; const int32_t* resampled; int T; int64_t n;
    const typename Col::value_t* cols; const int32_t* first_row; int h0, h1;        // records h0 .. h1 - 1 are this launch's
    const double* wrel; const double* bf; const StepCtrl* ctrl;
    double* stats_part;                                                             // [(h1 - h0) * kStats][gridDim.x]
};

template <class Col>
__global__ __launch_bounds__(kThreads) void lineage_stats_kernel(LineageStatsArgs<Col> a)
{
    extern __shared__ __attribute__((aligned(16))) double s_stat[];                 // [kWaves][(h1 - h0) * K]
    constexpr int K = Col::kStats;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int HK = (a.h1 - a.h0) * K;
    const double scale = a.ctrl->scale;
    for (int i = tid; i < kWaves * HK; i += kThreads) s_stat[i] = 0.0;
    __syncthreads();
    const int64_t ntiles = (a.n + kTile - 1) / kTile;
    for (int64_t tile = ntiles <= 2048 ? xcd_contiguous_tile((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t idx[kPPT]; double w[kPPT];
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const int64_t i = tile * kTile + (int64_t)k * kThreads + tid;
            idx[k] = i < a.n ? i : a.n - 1;                                          // (padding slots: wrel = 0)
            w[k] = a.wrel[i] * (a.bf[tile] * scale);
        }
        for (int t = a.T - 1; t >= 0; --t) {
            const int hb = max(a.first_row[t], a.h0), he = min(a.first_row[t + 1], a.h1);
            for (int h = hb; h < he; ++h) {
                double acc[K];
#pragma unroll
                for (int j = 0; j < K; ++j) acc[j] = 0.0;
                const typename Col::value_t* row = a.cols + (int64_t)h * a.n;
                typename Col::value_t x[kPPT];
#pragma unroll
                for (int k = 0; k < kPPT; ++k) x[k] = row[idx[k]];
                if constexpr (Col::kBins) {
                    // Which bins the wavefront holds at all, as ONE mask (a bit per bin, OR-ed over the lanes): a bin nobody holds sums to an
                    // exact zero and is neither accumulated nor reduced -- a three-state model pays for three of the eight bins.  The bins
                    // that are taken get the sums they always got (same terms, same order).
                    uint32_t m = 0;
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) m |= (uint32_t)x[k] < (uint32_t)K ? 1u << (x[k] & 31) : 0u;
                    int32_t mm = (int32_t)m;
                    mm |= dpp_or_i32<kDppRowShr1>(mm, 0); mm |= dpp_or_i32<kDppRowShr2>(mm, 0); mm |= dpp_or_i32<kDppRowShr4>(mm, 0); mm |= dpp_or_i32<kDppRowShr8>(mm, 0);
                    mm |= dpp_or_i32<kDppRowBcast15, 0xA>(mm, 0); mm |= dpp_or_i32<kDppRowBcast31, 0xC>(mm, 0);
                    const uint32_t present = (uint32_t)__builtin_amdgcn_readlane(mm, kWave - 1);
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if (!((present >> j) & 1u)) continue;                      // (scalar)
                        double aj = 0.0;
#pragma unroll
                        for (int k = 0; k < kPPT; ++k) aj += x[k] == j ? w[k] : 0.0;
                        const double sj = wave_sum(aj);
                        if (lane == 0) s_stat[wv * HK + (h - a.h0) * K + j] += sj;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < kPPT; ++k) Col::accumulate(x[k], w[k], acc);
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const double sj = wave_sum(acc[j]);
                        if (lane == 0) s_stat[wv * HK + (h - a.h0) * K + j] += sj;
                    }
                }
            }
            if (t > 0 && a.resampled[t - 1]) {
                const int32_t* arow = a.anc + (int64_t)t * a.n;
#pragma unroll
                for (int k = 0; k < kPPT; ++k) idx[k] = arow[idx[k]];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < HK; i += kThreads) {
        double s = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kWaves; ++w2) s += s_stat[w2 * HK + i];
        a.stats_part[(int64_t)i * gridDim.x + blockIdx.x] = s;
    }
}

}  // namespace cph
