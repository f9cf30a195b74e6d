// SMC bookkeeping of ONE step on fixed-point weights, for callers that produce the log-weights themselves (the unchanged-model
// path: cpprob/gpu.hpp launches the model body, then cpprob_hip_smc_bookkeep_fixed).  Same arithmetic as step_fixed.hpp with the
// reference taken from the generation itself -- R = the exact maximum of the log-weights, found by an order-free reduction first:
//   bbf_max_kernel        tile maxima -> the hierarchy's M words (atomic max, a block's last tile forwards)
//   bbf_quantize_kernel   q_i = min(rint(exp(lw_i - R) 2^32), 2^32 - 1); tile masses / squares -> the hierarchy (atomic adds)
//   bbf_ancestors_kernel  totals -> ESS, decision, evidence (one thread); ancestors of the next generation by the integer comb
//                         (fixed_locate / fixed_walk), the identity when the step does not resample
// Three short launches in place of weights_partials + scan_partials + resample (kernels.hpp), no floating-point CDF anywhere.
// Two copies of the hierarchy alternate: a step writes one and clears the upper levels of the other.
#pragma once
#include "step_fixed.hpp"

namespace cph {

// (bbf_publish_max, bbf_top_max, bbf_publish_mass: cpprob/detail/fixed_mass.hpp -- the unchanged-model step kernel publishes through them too)
__global__ __launch_bounds__(kThreads) void bbf_max_kernel(const double* __restrict__ logw, int64_t n, FHier f)
{
    __shared__ uint64_t s_red[kWaves];
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    double m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) if (j0 + k < n) m = fmax(m, logw[j0 + k]);
    const uint64_t mw = wave_max_u64(dkey(m));
    if (lane_id() == 0) s_red[wave_id()] = mw;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t mk = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) mk = umax64(mk, s_red[w]);
        bbf_publish_max(f, (int)blockIdx.x, (int)gridDim.x, mk);
    }
}

// the largest log-weight of a generation held by several ranks: the maximum of the keys every rank left in word 0 of its three
__device__ __forceinline__ double ranks_max(const uint64_t* __restrict__ all_keys, int world)
{
    const int lane = lane_id();
    return dkey_inv(wave_max_u64(lane < world ? all_keys[3 * lane] : 0ull));
}
// all_keys != nullptr: one shard of a joint population -- the reference is the POPULATION's exact maximum (the ranks' all-gathered keys)
__global__ __launch_bounds__(kThreads) void bbf_quantize_kernel(const double* __restrict__ logw, int64_t n, FHier f, uint32_t* __restrict__ q,
                                                                 const uint64_t* __restrict__ all_keys = nullptr, int world = 0)
{
    __shared__ uint64_t s_red[2 * kWaves];
    __shared__ double s_ref;
    if (wave_id() == 0) { const double r = all_keys ? ranks_max(all_keys, world) : bbf_top_max(f); if (threadIdx.x == 0) s_ref = r; }
    __syncthreads();
    const double ref = s_ref;
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    U4 w;
    uint64_t s_l = 0, q_l = 0;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) {
        const uint32_t v = j0 + k < n ? fix_weight(logw[j0 + k], ref) : 0u;
        w[k] = v; s_l += v; q_l += fix_square(v);
    }
    *reinterpret_cast<U4*>(q + j0) = w;
    const uint64_t sw = wave_sum_u64(s_l), qw = wave_sum_u64(q_l);
    if (lane_id() == 0) { s_red[wave_id()] = sw; s_red[kWaves + wave_id()] = qw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t St = 0, Qt = 0;
#pragma unroll
        for (int k = 0; k < kWaves; ++k) { St += s_red[k]; Qt += s_red[kWaves + k]; }
        bbf_publish_mass(f, (int)blockIdx.x, (int)gridDim.x, St, Qt);
    }
}

struct BbfArgs {
    FHier f; const uint32_t* q; int64_t n; int nb;
    double u0, ess_frac; int step, last;
    double* ess; int32_t* resampled; double* log_z; int32_t* anc;
    uint64_t seed;                                   // stratified / multinomial: the outputs' own uniforms (draw kResampleDrawBase(2) + step + 1)
    const uint32_t* strata_offs; int strata_k;       // multinomial, strata form: first outputs of the strata at this resampling (multinomial_strata_kernel)
};

// RS: kFixSystematic / kFixStratified / kFixMultinomial (strata form) -- the built-in models' step kernels' resampling, on the same masses
template <int RS = kFixSystematic>
__global__ __launch_bounds__(kThreads) void bbf_ancestors_kernel(BbfArgs a)
{
    __shared__ FixedLdsT<RS> L;
    __shared__ __attribute__((aligned(16))) FixedFound s_found;
    const int tid = threadIdx.x;
    const int bid = (int)blockIdx.x, nb = a.nb;
    const int64_t j0 = (int64_t)bid * kTile + (int64_t)tid * kPPT;
    const int64_t rem = a.n - (int64_t)bid * kTile;
    const int n_out = rem < kTile ? (int)rem : kTile;
    const double gj_first = (double)((int64_t)bid * kTile);
    if constexpr (RS != kFixMultinomial) {
        int32_t neg[kPPT];
        lane_fill(neg, (int32_t)-1);
        store4(L.slot, (int64_t)tid * kPPT, neg);
    }
    FixedCdf fc;
    fc.u0 = a.u0; fc.n_pop = (double)a.n; fc.base = 0; fc.inv = 0.0;
    fc.seed = a.seed; fc.draw = kResampleDrawBase + (uint64_t)a.step + 1; fc.uid0 = 0;
    if constexpr (RS == kFixStratified) { if (!a.last) stratified_stage(L, fc.seed, fc.draw, (uint64_t)gj_first); }
    __shared__ int s_w0, s_w1;
    __shared__ uint64_t s_S;
    if (wave_id() == 0) {
        FTotWords tw; ftot_fetch(a.f, tw);
        const FTot tot = ftot_sum(a.f, tw);
        const double ref = tot.M;
        const FixedDecision d = fixed_decide(tot.S, tot.Q, (double)a.n, a.ess_frac, !a.last);
        fc.inv = d.inv;
        if (bid == 0 && tid == 0) {
            a.ess[a.step] = d.ess;
            a.resampled[a.step] = d.resample ? 1 : 0;
            double lz = a.step == 0 ? 0.0 : *a.log_z;
            if (d.resample || a.last) lz += ref + log(d.W / (double)a.n);
            *a.log_z = lz;
        }
        FLocated loc{0, 0, 0};
        int w0 = 0, w1 = 0;
        if (d.resample && !a.last) {
            if constexpr (RS == kFixMultinomial) {
                ProbeWords pw;
                probe_fetch(a.f.h, bid, nb, pw);
                const StrataLocated sl = strata_locate(a.f, a.strata_offs, a.strata_k, nb, strata_near(a.strata_k, nb, bid), bid, (uint32_t)((int64_t)bid * kTile),
                                                       (uint32_t)((int64_t)bid * kTile + n_out - 1), tot.S, 0ull, tot.S, &pw);
                loc = sl.loc; w0 = sl.w0; w1 = sl.w1;
            } else loc = fixed_locate<RS>(a.f, fc, nb, gj_first, n_out, bid, nullptr);
        }
        if (tid == 0) { s_found.loc = loc; s_found.inv = d.inv; s_found.resample = d.resample ? 1 : 0; s_w0 = w0; s_w1 = w1; s_S = tot.S; }
    }
    __syncthreads();
    if (a.last) return;
    int32_t anc[kPPT];
#pragma unroll
    for (int k = 0; k < kPPT; ++k) anc[k] = (int32_t)(j0 + k);
    if (s_found.resample) {
        fc.inv = s_found.inv;
        if constexpr (RS == kFixMultinomial) {
            StrataLocated sl;
            sl.loc = s_found.loc; sl.w0 = s_w0; sl.w1 = s_w1;
            bool mine[kPPT];
            strata_walk(a.strata_offs, a.strata_k, a.q, nb, sl, s_S, j0, a.seed, kResampleDrawBase2 + (uint64_t)a.step + 1, (uint64_t)j0, anc, L, 0ull, s_S, mine);
        } else {
            const U4 z = {0u, 0u, 0u, 0u};
            fixed_walk<RS>(fc, a.q, a.n, nb, true, gj_first, n_out, s_found.loc, bid, false, z, z, z, anc, L);
        }
#pragma unroll
        for (int k = 0; k < kPPT; ++k) anc[k] = max(anc[k], 0);
    } else {
#pragma unroll
        for (int k = 0; k < kPPT; ++k) anc[k] = (int32_t)(j0 + k);
    }
#pragma unroll
    for (int k = 0; k < kPPT; ++k) if (j0 + k < a.n) a.anc[j0 + k] = anc[k];
}

}  // namespace cph
