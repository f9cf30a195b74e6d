// Particle kernels of the SIS / SMC engine (gfx950, wave64, fp64).
//
// Data layout in HBM (one context = one shard of the population):
//   values   [T][ld]   value of predict hit t in slot i of generation t (fp64 or int32);
//                      row t-1 doubles as the state the step-t kernel reads (no separate state)
//   anc      [T][ld]   int32 slot of generation t-1 extended by slot i of generation t
//   logw     [2][ld]   fp64 log-weights, ping-pong between steps
//   part     [nb]      per-tile {max, sum exp(lw-max), sum exp(2(lw-max))} written by the kernel
//                      that produced the weights (no separate normalisation pass over logw)
//   bc       [nb+1]    exclusive prefix of the tile sums rescaled to the global max: the
//                      resampling CDF at tile granularity
//   ctrl               device-resident control block: max, W, Q, ESS, log Z, resample decision
// ld = N rounded up to 4 so that every lane's 4 consecutive particles are one aligned 32-B
// (fp64) or 16-B (int32) access.  One particle per lane-slot, structure of arrays, every access
// of a step coalesced; the only gather (ancestor state) reads sorted indices.
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>
#include <stdint.h>

#include "dist.hpp"
#include "models.hpp"
#include "rng.hpp"
#include "wave.hpp"

namespace cph {

struct __attribute__((aligned(32))) Partial { double m, s, q, pad; };

struct StepCtrl {
    double M;         // max log-weight of the current generation (global over shards)
    double W;         // sum exp(logw - M)
    double Q;         // sum exp(2 (logw - M))
    double ess;       // W^2 / Q
    double log_z;     // accumulated log evidence
    double cdf_lo;    // sharded runs: global CDF offset of this shard (0 on one GPU)
    double w_local;   // this shard's sum rescaled to M
    double scale;     // exp(M_local - M): factor that rescales bc[] to the global max
    int32_t do_resample;  // decision taken after the last weighted step
    int32_t n_resampled;
    int32_t pad[2];
};

enum { RS_SYSTEMATIC = 0, RS_STRATIFIED = 1, RS_PRECOMPUTED = 2 };

// ---------------------------------------------------------------------------------------------
// 4-wide accesses
// ---------------------------------------------------------------------------------------------
template <class T> struct Vec4;
template <> struct Vec4<double> { using type = double __attribute__((ext_vector_type(4))); };
template <> struct Vec4<int32_t> { using type = int __attribute__((ext_vector_type(4))); };

template <class T>
__device__ __forceinline__ void load4(const T* __restrict__ p, int64_t i, int64_t n, T (&v)[kPPT], T fill)
{
    if (i + kPPT <= n) {
        const typename Vec4<T>::type x = *reinterpret_cast<const typename Vec4<T>::type*>(p + i);
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    } else {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) v[k] = (i + k < n) ? p[i + k] : fill;
    }
}

template <class T>
__device__ __forceinline__ void store4(T* __restrict__ p, int64_t i, int64_t n, const T (&v)[kPPT])
{
    if (i + kPPT <= n) {
        typename Vec4<T>::type x;
        x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
        *reinterpret_cast<typename Vec4<T>::type*>(p + i) = x;
    } else {
#pragma unroll
        for (int k = 0; k < kPPT; ++k)
            if (i + k < n) p[i + k] = v[k];
    }
}

// ---------------------------------------------------------------------------------------------
// Tile partial of the weights this workgroup just produced: {max, sum e, sum e^2}
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void write_partial(const double (&lw)[kPPT], const bool (&valid)[kPPT], Partial* __restrict__ part,
                                              double* s_scr)
{
    double m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kPPT; ++k)
        if (valid[k]) m = fmax(m, lw[k]);
    m = block_max(m, s_scr);
    double s = 0.0, q = 0.0;
    if (m != -INFINITY) {
#pragma unroll
        for (int k = 0; k < kPPT; ++k)
            if (valid[k]) { const double e = exp(lw[k] - m); s += e; q += e * e; }
    }
    s = block_sum(s, s_scr);
    q = block_sum(q, s_scr);
    if (threadIdx.x == 0) { Partial p; p.m = m; p.s = s; p.q = q; p.pad = 0.0; part[blockIdx.x] = p; }
}

// Standalone: partials of an arbitrary log-weight array (building block / tests).
__global__ __launch_bounds__(kThreads) void weights_partials_kernel(const double* __restrict__ logw, int64_t n, Partial* __restrict__ part)
{
    __shared__ double s_scr[16];
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    double lw[kPPT]; bool valid[kPPT];
    load4(logw, j0, n, lw, (double)-INFINITY);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) valid[k] = j0 + k < n;
    write_partial(lw, valid, part, s_scr);
}

// ---------------------------------------------------------------------------------------------
// scan_partials: one workgroup of 1024 threads turns the tile partials of the generation
// just weighted into: global max M, normaliser W, ESS, the tile-level CDF bc[], the evidence
// increment and the resampling decision (ESS < ess_frac * N; thesis p.37).  Launched once per
// step; everything stays on the device, so the host never waits inside a run.
// ---------------------------------------------------------------------------------------------
constexpr int kScanThreads = 1024;

struct ScanArgs {
    const Partial* part; int nb;
    double* bc; StepCtrl* ctrl;
    int t, T;
    double n_global, ess_frac;
    double* ess_trace; int32_t* resampled;
    int force_no_resample;     // SIS: never resample
    // sharded: when all_totals != nullptr the global (M, W, Q) come from the all-gathered
    // per-rank totals instead of the local partials
    const double* all_totals; int world, rank;
    double* local_totals;      // out: {M_local, W_local, Q_local} for the all-gather (phase 1)
    int phase;                 // 0: single GPU (everything); 1: local totals + local bc only; 2: combine
};

__global__ __launch_bounds__(kScanThreads) void scan_partials_kernel(ScanArgs a)
{
    __shared__ double s_scr[2 * (kScanThreads / kWave)];
    constexpr int NW = kScanThreads / kWave;
    const int tid = threadIdx.x;
    StepCtrl* ctrl = a.ctrl;

    if (a.phase != 2) {
        const int chunk = (a.nb + kScanThreads - 1) / kScanThreads;
        const int lo = tid * chunk, hi = min(a.nb, lo + chunk);
        double m = -INFINITY;
        for (int c = lo; c < hi; ++c) m = fmax(m, a.part[c].m);
        const double M = block_max<NW>(m, s_scr);
        double S = 0.0, Q = 0.0;
        for (int c = lo; c < hi; ++c) {
            const Partial p = a.part[c];
            const double e = (p.m == -INFINITY) ? 0.0 : exp(p.m - M);
            S += p.s * e;
            Q += p.q * (e * e);
        }
        double W;
        const double excl = block_excl_scan<NW>(S, s_scr, &W);
        const double Qt = block_sum<NW>(Q, s_scr);
        double run = excl;
        for (int c = lo; c < hi; ++c) {
            const Partial p = a.part[c];
            const double e = (p.m == -INFINITY) ? 0.0 : exp(p.m - M);
            a.bc[c] = run;
            run += p.s * e;
        }
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
        if (tid == 0) {
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
    }
    if (tid == 0) {
        const double W = ctrl->W, Q = ctrl->Q, M = ctrl->M;
        const double ess = W * W / Q;
        ctrl->ess = ess;
        const bool last = a.t + 1 == a.T;
        const bool rs = !a.force_no_resample && !last && (ess < a.ess_frac * a.n_global);
        ctrl->do_resample = rs ? 1 : 0;
        if (a.t == 0) { ctrl->log_z = 0.0; ctrl->n_resampled = 0; }
        if (rs || last) ctrl->log_z += M + log(W / a.n_global);
        if (rs) ctrl->n_resampled += 1;
        if (a.ess_trace) a.ess_trace[a.t] = ess;
        if (a.resampled) a.resampled[a.t] = rs ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Ancestor search for one tile of (sorted) output positions.
//   position of output j: systematic (j + u0) * W/N_out, stratified (j + u_j) * W/N_out
//   ancestor = min{k : C_k > p}, C = inclusive CDF of w_k = exp(logw_k - M)
// The tile walks the source tiles its positions fall into (usually 1-2: tile sums are nearly
// equal), rebuilds each source tile's CDF in LDS from logw (scan of 1024 doubles) and binary-
// searches it.  The full-resolution CDF is never written to HBM.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_search(const double* __restrict__ bc, int lo, int hi, double p)
{
    // largest c in [lo, hi) with bc[c] <= p (bc[lo] <= p is guaranteed by the caller)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (bc[mid] <= p) lo = mid; else hi = mid;
    }
    return lo;
}

template <int RS>
__device__ __forceinline__ void find_ancestors(const double* __restrict__ logw, int64_t n_in, const double* __restrict__ bc, int nb,
                                               double M, double W, double bc_scale, double cdf_lo, uint64_t seed, uint64_t step,
                                               uint64_t gj0, uint64_t n_total_out, int n_valid_tile, const bool (&valid)[kPPT],
                                               int32_t (&anc)[kPPT], double* s_cdf, double* s_scr, double* s_pos, int* s_idx)
{
    const int tid = threadIdx.x;
    const double stepw = W / (double)n_total_out;
    double p[kPPT];
    double u0 = 0.0;
    if (RS == RS_SYSTEMATIC) u0 = draw_u01(seed, 0, kResampleDrawBase + step);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        const double u = RS == RS_SYSTEMATIC ? u0 : draw_u01(seed, gj0 + k, kResampleDrawBase + step);
        // positions are relative to this shard's CDF segment [cdf_lo, cdf_lo + w_local)
        p[k] = ((double)(gj0 + k) + u) * stepw - cdf_lo;
        anc[k] = 0;
    }
    if (tid == 0) s_pos[0] = p[0];
    {
        const int last = n_valid_tile - 1;
        if ((last >> 2) == tid) s_pos[1] = (last & 3) == 0 ? p[0] : ((last & 3) == 1 ? p[1] : ((last & 3) == 2 ? p[2] : p[3]));
    }
    __syncthreads();
    const double p_first = fmax(s_pos[0], 0.0), p_last = s_pos[1];
    bool res[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) res[k] = !valid[k];

    int c = tile_search(bc, 0, nb, p_first / bc_scale);
    for (;;) {
        const int64_t base = (int64_t)c * kTile + (int64_t)tid * kPPT;
        double w[kPPT];
        load4(logw, base, n_in, w, (double)-INFINITY);
#pragma unroll
        for (int k = 0; k < kPPT; ++k) w[k] = exp(w[k] - M);   // exp(-inf) = 0 for padding
        w[1] += w[0]; w[2] += w[1]; w[3] += w[2];
        double tot;
        const double excl = block_excl_scan(w[3], s_scr, &tot);
        const double off = bc[c] * bc_scale + excl;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) s_cdf[tid * kPPT + k] = off + w[k];
        __syncthreads();
        const double hi_c = (c == nb - 1) ? INFINITY : bc[c + 1] * bc_scale;
        const int64_t rem = n_in - (int64_t)c * kTile;
        const int n_src = rem < kTile ? (int)rem : kTile;
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            if (!res[k] && p[k] < hi_c) {
                int lo = 0, hi = n_src;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (s_cdf[mid] > p[k]) hi = mid; else lo = mid + 1;
                }
                if (lo >= n_src) lo = n_src - 1;
                anc[k] = (int32_t)((int64_t)c * kTile + lo);
                res[k] = true;
            }
        }
        if (p_last < hi_c) break;   // workgroup-uniform
        if (tid == 0) *s_idx = INT_MAX;
        __syncthreads();
        int mine = INT_MAX;
#pragma unroll
        for (int k = kPPT - 1; k >= 0; --k)
            if (!res[k]) mine = tid * kPPT + k;
        if (mine != INT_MAX) atomicMin(s_idx, mine);
        __syncthreads();
        const int f = *s_idx;
        if ((f >> 2) == tid) s_pos[0] = (f & 3) == 0 ? p[0] : ((f & 3) == 1 ? p[1] : ((f & 3) == 2 ? p[2] : p[3]));
        __syncthreads();
        c = tile_search(bc, c + 1, nb, s_pos[0] / bc_scale);
    }
}

// Standalone resampler (building block; also the multinomial path's first half is elsewhere).
struct ResampleArgs {
    const double* logw; int64_t n_in;
    const double* bc; int nb; const StepCtrl* ctrl;
    uint64_t seed, step, j0, n_total_out; int64_t n_out;
    int32_t* anc;
};

template <int RS>
__global__ __launch_bounds__(kThreads) void resample_kernel(ResampleArgs a)
{
    __shared__ double s_cdf[kTile];
    __shared__ double s_scr[16];
    __shared__ double s_pos[2];
    __shared__ int s_idx;
    const int64_t l0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    bool valid[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) valid[k] = l0 + k < a.n_out;
    const int64_t rem = a.n_out - (int64_t)blockIdx.x * kTile;
    const int n_valid_tile = rem < kTile ? (int)rem : kTile;
    int32_t anc[kPPT];
    find_ancestors<RS>(a.logw, a.n_in, a.bc, a.nb, a.ctrl->M, a.ctrl->W, a.ctrl->scale, a.ctrl->cdf_lo, a.seed, a.step,
                       a.j0 + (uint64_t)l0, a.n_total_out, n_valid_tile, valid, anc, s_cdf, s_scr, s_pos, &s_idx);
    store4(a.anc, l0, a.n_out, anc);
}

// Multinomial (thesis Alg. 1 p.36, literal): independent positions u_j * W, unsorted, so the
// full-resolution CDF is materialised once (cdf_kernel) and searched per output.
__global__ __launch_bounds__(kThreads) void cdf_kernel(const double* __restrict__ logw, int64_t n, const double* __restrict__ bc,
                                                        const StepCtrl* __restrict__ ctrl, double* __restrict__ cdf)
{
    __shared__ double s_scr[16];
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    double w[kPPT];
    load4(logw, j0, n, w, (double)-INFINITY);
    const double M = ctrl->M;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) w[k] = exp(w[k] - M);
    w[1] += w[0]; w[2] += w[1]; w[3] += w[2];
    double tot;
    const double off = bc[blockIdx.x] * ctrl->scale + block_excl_scan(w[3], s_scr, &tot);
#pragma unroll
    for (int k = 0; k < kPPT; ++k) w[k] += off;
    store4(cdf, j0, n, w);
}

__global__ __launch_bounds__(kThreads) void multinomial_kernel(const double* __restrict__ cdf, int64_t n_in, const StepCtrl* __restrict__ ctrl,
                                                                uint64_t seed, uint64_t step, uint64_t j0, int64_t n_out, int32_t* __restrict__ anc)
{
    const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= n_out) return;
    const double p = draw_u01(seed, j0 + (uint64_t)i, kResampleDrawBase + step) * ctrl->W - ctrl->cdf_lo;
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
// draw priors -> sum logpdf(observe) -> record predicts.  Writes values[t][i] and logw[i]
// coalesced, plus the tile partial of the final weights.
// ---------------------------------------------------------------------------------------------
template <class Model>
struct SisArgs {
    ModelParams mp; const double* obs; int T; int64_t n, ld;
    uint64_t seed, pid0;
    typename Model::value_t* values; double* logw; Partial* part;
};

template <class Model>
__global__ __launch_bounds__(kThreads) void sis_kernel(SisArgs<Model> a)
{
    using V = typename Model::value_t;
    __shared__ double s_scr[16];
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    bool valid[kPPT]; double lw[kPPT]; V x[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) { valid[k] = j0 + k < a.n; lw[k] = 0.0; x[k] = V(0); }   // start_trace(): log_w_ = 0
    for (int t = 0; t < a.T; ++t) {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            x[k] = Model::propagate(a.mp, a.seed, a.pid0 + (uint64_t)(j0 + k), t, x[k]);   // sample: distr(get_rng())  cpprob.hpp:72-74
            lw[k] += Model::loglik(a.mp, x[k], t, a.obs);                                   // observe: log_w_ += logpdf  state.cpp:212-223
        }
        store4(a.values + (int64_t)t * a.ld, j0, a.n, x);                                   // predict: add_predict       state.hpp:312-327
    }
    store4(a.logw, j0, a.n, lw);                                                            // finish_trace()
    write_partial(lw, valid, a.part, s_scr);
}

// ---------------------------------------------------------------------------------------------
// SMC step t (fused): [resample generation t-1 -> ancestors] -> gather ancestor state ->
// sample x_t -> weight by observe t -> record predict + ancestor -> tile partial.
// ---------------------------------------------------------------------------------------------
template <class Model>
struct StepArgs {
    ModelParams mp; const double* obs; int t, T; int64_t n, ld;
    uint64_t seed, pid0, n_global;
    typename Model::value_t* values; int32_t* anc;
    const double* logw_prev; double* logw_next;
    const Partial* part_prev_unused; Partial* part;
    const double* bc; int nb; const StepCtrl* ctrl;
    const int32_t* anc_pre;   // RS_PRECOMPUTED: ancestors computed by multinomial_kernel
};

template <class Model, int RS>
__global__ __launch_bounds__(kThreads) void smc_step_kernel(StepArgs<Model> a)
{
    using V = typename Model::value_t;
    __shared__ double s_cdf[RS == RS_PRECOMPUTED ? 1 : kTile];
    __shared__ double s_scr[16];
    __shared__ double s_pos[2];
    __shared__ int s_idx;
    const int tid = threadIdx.x;
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)tid * kPPT;
    const int t = a.t;
    bool valid[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) valid[k] = j0 + k < a.n;
    const bool resample = t > 0 && a.ctrl->do_resample != 0;   // workgroup-uniform (scalar load)

    int32_t anc[kPPT]; double lw[kPPT];
    if (!resample) {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) { anc[k] = (int32_t)(j0 + k); lw[k] = 0.0; }
        if (t > 0) load4(a.logw_prev, j0, a.n, lw, 0.0);     // weights carry over when no resampling happened
    } else {
        if (RS == RS_PRECOMPUTED) {
            load4(a.anc_pre, j0, a.n, anc, 0);
        } else {
            const int64_t rem = a.n - (int64_t)blockIdx.x * kTile;
            const int n_valid_tile = rem < kTile ? (int)rem : kTile;
            find_ancestors<RS>(a.logw_prev, a.n, a.bc, a.nb, a.ctrl->M, a.ctrl->W, a.ctrl->scale, a.ctrl->cdf_lo, a.seed, (uint64_t)t,
                               a.pid0 + (uint64_t)j0, a.n_global, n_valid_tile, valid, anc, s_cdf, s_scr, s_pos, &s_idx);
        }
#pragma unroll
        for (int k = 0; k < kPPT; ++k) lw[k] = 0.0;          // equal weights after resampling
    }

    V x[kPPT];
    const V* prev_row = a.values + (int64_t)(t > 0 ? t - 1 : 0) * a.ld;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        V prev = V(0);
        if (t > 0 && valid[k]) prev = prev_row[anc[k]];                                   // ancestor's state (sorted gather)
        x[k] = Model::propagate(a.mp, a.seed, a.pid0 + (uint64_t)(j0 + k), t, prev);     // sample #t
        lw[k] += Model::loglik(a.mp, x[k], t, a.obs);                                     // observe #t
    }
    store4(a.values + (int64_t)t * a.ld, j0, a.n, x);                                     // predict #t
    store4(a.anc + (int64_t)t * a.ld, j0, a.n, anc);
    store4(a.logw_next, j0, a.n, lw);
    write_partial(lw, valid, a.part, s_scr);
}

// ---------------------------------------------------------------------------------------------
// Posterior read-out: StatsPrinter / EmpiricalDistribution over the final particles' full
// traces (reference stats_printer.hpp:88-120: the k-th hit of a predict address in a trace
// goes to the k-th distribution; empirical_distribution.hpp:30-40,52-81).  A final particle's
// trace is its ancestral line, so each lane walks anc[] backwards from its final slot and
// accumulates W_i * f(x_t) per step; ancestors are sorted, so the walk stays coalesced and
// collapses onto the surviving lineages (L2 hits).
// ---------------------------------------------------------------------------------------------
template <class Model>
struct SmoothArgs {
    const typename Model::value_t* values; const int32_t* anc; const double* logw; const StepCtrl* ctrl;
    const int32_t* resampled; int T; int64_t n, ld; int identity;
    double* stats_part;   // [gridDim.x][T * kStats]
    typename Model::value_t* paths;   // optional [T][ld]: materialised traces (dump / tests)
};

template <class Model>
__global__ __launch_bounds__(kThreads) void smooth_kernel(SmoothArgs<Model> a)
{
    using V = typename Model::value_t;
    constexpr int K = Model::kStats;
    extern __shared__ __attribute__((aligned(16))) double s_stat[];   // [kWaves][T*K]
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int TK = a.T * K;
    for (int i = tid; i < kWaves * TK; i += kThreads) s_stat[i] = 0.0;
    __syncthreads();
    const double M = a.ctrl->M;
    const int64_t ntiles = (a.n + kTile - 1) / kTile;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t idx[kPPT]; double w[kPPT]; int64_t self[kPPT];
#pragma unroll
        for (int k = 0; k < kPPT; ++k) {
            const int64_t i = tile * kTile + (int64_t)k * kThreads + tid;   // lane-strided: coalesced first touch
            self[k] = i;
            idx[k] = i < a.n ? i : 0;
            w[k] = i < a.n ? exp(a.logw[i] - M) : 0.0;
        }
        for (int t = a.T - 1; t >= 0; --t) {
            double acc[K];
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = 0.0;
            const V* row = a.values + (int64_t)t * a.ld;
#pragma unroll
            for (int k = 0; k < kPPT; ++k) {
                const V x = row[idx[k]];
                Model::accumulate(x, w[k], acc);
                if (a.paths && self[k] < a.n) a.paths[(int64_t)t * a.ld + self[k]] = x;
            }
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] = wave_sum(acc[j]);
            if (lane == 0) {
#pragma unroll
                for (int j = 0; j < K; ++j) s_stat[wv * TK + t * K + j] += acc[j];
            }
            if (t > 0 && !a.identity && a.resampled[t - 1]) {
                const int32_t* arow = a.anc + (int64_t)t * a.ld;
#pragma unroll
                for (int k = 0; k < kPPT; ++k) idx[k] = arow[idx[k]];
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < TK; i += kThreads) {
        double s = 0.0;
#pragma unroll
        for (int w2 = 0; w2 < kWaves; ++w2) s += s_stat[w2 * TK + i];
        a.stats_part[(int64_t)blockIdx.x * TK + i] = s;
    }
}

// Sums the per-workgroup partial statistics in a fixed order (bitwise reproducible) and
// normalises: real -> {mean, raw2 - mean^2}; int -> probabilities.
__global__ __launch_bounds__(kThreads) void finalize_kernel(const double* __restrict__ stats_part, int grid, int T, int K, int is_int,
                                                             const StepCtrl* __restrict__ ctrl, double* __restrict__ stats)
{
    const int TK = T * K;
    const double W = ctrl->W;
    if (is_int) {
        for (int i = blockIdx.x * kThreads + threadIdx.x; i < TK; i += gridDim.x * kThreads) {
            double s = 0.0;
            for (int g = 0; g < grid; ++g) s += stats_part[(int64_t)g * TK + i];
            stats[i] = s / W;
        }
    } else {
        for (int t = blockIdx.x * kThreads + threadIdx.x; t < T; t += gridDim.x * kThreads) {
            double s1 = 0.0, s2 = 0.0;
            for (int g = 0; g < grid; ++g) { s1 += stats_part[(int64_t)g * TK + t * K]; s2 += stats_part[(int64_t)g * TK + t * K + 1]; }
            const double mean = s1 / W;
            stats[t * K] = mean;
            stats[t * K + 1] = s2 / W - mean * mean;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Elementwise building blocks (unit-parity surface)
// ---------------------------------------------------------------------------------------------
__global__ void philox_blocks_kernel(uint64_t seed, uint64_t pid0, uint64_t draw, int64_t n, uint32_t* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32x4 r = draw_block(seed, pid0 + (uint64_t)i, draw);
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
    if (i < n) out[i] = (int32_t)discrete_from_u_dyn(draw_u01(seed, pid0 + (uint64_t)i, draw), dw.w, dw.k);
}

__global__ void draw_uniform_real_kernel(uint64_t seed, uint64_t pid0, uint64_t draw, double a, double b, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = draw_uniform_real(seed, pid0 + (uint64_t)i, draw, a, b);
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

template <class T>
__global__ void gather_kernel(const T* __restrict__ src, const int32_t* __restrict__ idx, int64_t n, T* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

// Weighted moments / histogram of one column against one log-weight array (EmpiricalDistribution
// on device): the column is a 1-step "trace", so this is smooth_kernel with T = 1.
struct ColumnReal { using value_t = double; static constexpr int kStats = 2;
    __device__ static __forceinline__ void accumulate(double x, double w, double (&acc)[2]) { acc[0] += w * x; acc[1] += w * (x * x); } };
struct ColumnInt8 { using value_t = int32_t; static constexpr int kStats = 8;
    __device__ static __forceinline__ void accumulate(int32_t x, double w, double (&acc)[8]) {
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[s] += x == s ? w : 0.0; } };

}  // namespace cph
