// Built-in model step functors: the bodies of the reference's target models in
// "one observe per step" form, so that the SIS kernel can run them to completion in
// registers and the SMC kernel can stop every particle at each observe.
//
// Statement order inside a step follows the reference model line by line; the draw index of a
// sample statement is its ordinal in the trace (t-th sample -> draw t).  Every functor advances the 4 consecutive
// particles a lane owns at once so that they share Philox blocks (rng.hpp).
#pragma once
#include "cpprob/detail/dist.hpp"
#include "cpprob/detail/rng.hpp"

namespace cph {

// Host-precomputed constants (identical IEEE fp64 operations to the device's own, so hoisting
// them does not change a single bit of the discrete draws).
struct ModelParams {
    // gaussian_unknown_mean: prior N(mu0, sigma0), likelihood N(mu, sigma)
    double mu0, sigma0, sigma, log_norm_lik;      // log_norm_lik = log(2*pi*sigma^2)
    double inv_sigma;                             // 1 / sigma (host-computed)
    // hmm: k = 3
    double hmm_mean[3];
    uint64_t hmm_thr[3][2];                       // ceil(2^32 * cumulative probability) of row s
    double log_norm_unit;                         // log(2*pi*1*1)
    const double* ll_tab;                         // hmm: [T][3] log N(y_t; mean[s], 1), device pointer
    const double* e_tab;                          // hmm: [T][4] exp(ll - max ll) for s = 0..2, then max ll
    // gaussian_2d_unk_mean: independent components, prior N(nd_mean[d], nd_sigma[d]); likelihood sigma / log_norm_lik as above
    double nd_mean[4], nd_sigma[4];
    // Gaussian models: the log-weight as a polynomial in the standard-normal variate z_d of row d (x_d = mean_d + sigma_d z_d),
    // logw = sum_d quad[d][0] z_d^2 + quad[d][1] z_d + quad[d][2], host-evaluated from the observes; lw_ref = the smallest grid
    // point {k ln 2} above its maximum over z: a reference every particle's weight can be taken against BEFORE any is known
    double quad[2][3];
    double lw_ref;
    // hmm over a caller-given table (cpprob_hip_set_hmm): k states, device tables
    int hk;                                       // 2..8
    const uint64_t* hk_thr;                       // [k][8]: ceil(2^32 * cumulative probability) of row s, entries 0..k-2
    const double* hk_ll;                          // [T][8] log N(y_t; mean[s], 1)
};

// reference include/models/models.hpp:22-35 and src/models/gaussian.cpp:6-17 (same body,
// different hyper-parameters).  One step, two observes, one real predict.
struct ModelGaussian {
    using value_t = double;
    using store_t = double;
    static constexpr bool kIsInt = false;
    static constexpr int kStats = 2;  // sum w x, sum w x^2
    static constexpr int kWeightTable = 0;   // incremental weights are continuous
    __device__ static __forceinline__ void weight_table(const ModelParams&, int, double (&)[1], double (&)[1], double&) {}
    __device__ static __forceinline__ int weight_index(value_t) { return 0; }
    // the random part of the step's sample statement depends on nothing but (seed, id, t): kernels draw it early,
    // in the shadow of their first memory round trip, and apply it once the ancestor's state is known
    struct Rand { double z[4]; };
    __device__ static __forceinline__ void draw4(uint64_t seed, uint64_t pid0, int /*t*/, Rand& r) { draw_std_normals4(seed, pid0, 0, r.z); }
    __device__ static __forceinline__ void apply4(const ModelParams& mp, int /*t*/, const Rand& r, const value_t (&)[4], value_t (&x)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = mp.mu0 + mp.sigma0 * r.z[k];  // mu = sample(prior, true)   models.hpp:26-27
    }
    __device__ static __forceinline__ void propagate4(const ModelParams& mp, uint64_t seed, uint64_t pid0, int t, const value_t (&prev)[4],
                                                      value_t (&x)[4])
    {
        Rand r; draw4(seed, pid0, t, r); apply4(mp, t, r, prev, x);
    }
    // the same weight from the variate: both observes' log-densities are one quadratic in z (models.hpp:32-33 expanded on the host)
    static constexpr bool kBounded = true;
    __device__ static __forceinline__ double logw_of_z(const ModelParams& mp, int /*d*/, double z) { return fma(fma(mp.quad[0][0], z, mp.quad[0][1]), z, mp.quad[0][2]); }
    __device__ static __forceinline__ double loglik(const ModelParams& mp, value_t mu, int /*t*/, const double* __restrict__ obs)
    {
        double lw = 0.0;                                                  // TraceInfer::log_w_ = 0     trace.hpp:59
        lw += normal_logpdf_scaled(obs[0], mu, mp.inv_sigma, mp.log_norm_lik);   // observe(likelihood, y1)  models.hpp:32
        lw += normal_logpdf_scaled(obs[1], mu, mp.inv_sigma, mp.log_norm_lik);   // observe(likelihood, y2)  models.hpp:33
        return lw;
    }
    __device__ static __forceinline__ void accumulate(value_t x, double w, double (&acc)[kStats])
    {
        acc[0] += w * x;
        acc[1] += w * (x * x);
    }
};

// reference include/models/models.hpp:38-49: ONE vector-valued sample (diagonal multivariate normal: component d is
// the d-th draw, multivariate_normal.hpp:268-274), ONE vector-valued observe (logpdf = sum over components,
// utils_multivariate_normal.hpp:22-33), ONE NDArray predict.  Variable-width SoA: component d is "step" d of the SIS
// kernel -- row d of values[] -- and the weight is the sum over rows, which is exactly the component sum.
struct ModelGaussianND {
    using value_t = double;
    using store_t = double;
    static constexpr bool kIsInt = false;
    static constexpr int kStats = 2;
    static constexpr int kWeightTable = 0;
    __device__ static __forceinline__ void weight_table(const ModelParams&, int, double (&)[1], double (&)[1], double&) {}
    __device__ static __forceinline__ int weight_index(value_t) { return 0; }
    struct Rand { double z[4]; };
    __device__ static __forceinline__ void draw4(uint64_t seed, uint64_t pid0, int d, Rand& r) { draw_std_normals4(seed, pid0, (uint64_t)d, r.z); }
    __device__ static __forceinline__ void apply4(const ModelParams& mp, int d, const Rand& r, const value_t (&)[4], value_t (&x)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = mp.nd_mean[d] + mp.nd_sigma[d] * r.z[k];           // models.hpp:42-43
    }
    __device__ static __forceinline__ void propagate4(const ModelParams& mp, uint64_t seed, uint64_t pid0, int d, const value_t (&prev)[4],
                                                      value_t (&x)[4])
    {
        Rand r; draw4(seed, pid0, d, r); apply4(mp, d, r, prev, x);
    }
    static constexpr bool kBounded = true;
    __device__ static __forceinline__ double logw_of_z(const ModelParams& mp, int d, double z) { return fma(fma(mp.quad[d][0], z, mp.quad[d][1]), z, mp.quad[d][2]); }
    __device__ static __forceinline__ double loglik(const ModelParams& mp, value_t mu, int d, const double* __restrict__ obs)
    {
        return normal_logpdf_scaled(obs[d], mu, mp.inv_sigma, mp.log_norm_lik);               // models.hpp:46-47, component d
    }
    __device__ static __forceinline__ void accumulate(value_t x, double w, double (&acc)[kStats])
    {
        acc[0] += w * x;
        acc[1] += w * (x * x);
    }
};

// reference include/models/models.hpp:67-80: x_0 = 0, x_t ~ N(x_{t-1}, 1), y_t ~ N(x_t, 1),
// predict(x_t, "State") after the observe.
struct ModelLinearGaussian1D {
    using value_t = double;
    using store_t = double;
    static constexpr bool kIsInt = false;
    static constexpr int kStats = 2;
    static constexpr int kWeightTable = 0;
    __device__ static __forceinline__ void weight_table(const ModelParams&, int, double (&)[1], double (&)[1], double&) {}
    __device__ static __forceinline__ int weight_index(value_t) { return 0; }
    struct Rand { double z[4]; };
    __device__ static __forceinline__ void draw4(uint64_t seed, uint64_t pid0, int t, Rand& r) { draw_std_normals4(seed, pid0, (uint64_t)t, r.z); }
    // half of draw4 for an EVEN particle id: the normals of particles pid, pid + 1 (one Philox block, one Box-Muller pair)
    static constexpr bool kHasDraw2 = true;
    __device__ static __forceinline__ void draw2(uint64_t seed, uint64_t pid_even, int t, double& z0, double& z1) { box_muller(draw_block(seed, pid_even >> 1, (uint64_t)t), z0, z1); }
    __device__ static __forceinline__ void apply4(const ModelParams&, int t, const Rand& r, const value_t (&prev)[4], value_t (&x)[4])
    {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double state = t == 0 ? 0.0 : prev[k];                  // models.hpp:72
            x[k] = state + 1.0 * r.z[k];                                  // :74-75  normal{state, 1}
        }
    }
    __device__ static __forceinline__ void propagate4(const ModelParams& mp, uint64_t seed, uint64_t pid0, int t, const value_t (&prev)[4],
                                                      value_t (&x)[4])
    {
        Rand r; draw4(seed, pid0, t, r); apply4(mp, t, r, prev, x);
    }
    static constexpr bool kBounded = false;
    __device__ static __forceinline__ double logw_of_z(const ModelParams&, int, double) { return 0.0; }
    __device__ static __forceinline__ double loglik(const ModelParams& mp, value_t x, int t, const double* __restrict__ obs)
    {
        return normal_logpdf_hoisted(obs[t], x, 1.0, mp.log_norm_unit);   // :76-77
    }
    __device__ static __forceinline__ void accumulate(value_t x, double w, double (&acc)[kStats])
    {
        acc[0] += w * x;
        acc[1] += w * (x * x);
    }
};

// reference include/models/models.hpp:114-141: 3 states, uniform initial state, rows T :123-125,
// emission N(state_mean[s], 1); predict(state, "State") before the observe.
struct ModelHmm3 {
    using value_t = int32_t;
    using store_t = int8_t;           // states 0..2: one byte per particle-step in the particle store (the C ABI widens on copy-out)
    static constexpr bool kIsInt = true;
    static constexpr int kStats = 3;  // sum w [x == s]
    // the incremental weight of a step takes one of 3 values: log N(y_t; mean[s], 1).  ll_tab row t holds them,
    // e_tab row t holds {exp(ll - max ll)} and max ll (host-computed once per run)
    static constexpr int kWeightTable = 3;
    static constexpr int kTraceBits = 2;   // a state in a trace word (trace_words.hpp: T <= 16 states ride in 32 bits)
    __device__ static __forceinline__ void weight_table(const ModelParams& mp, int t, double (&ll)[3], double (&e)[3], double& mref)
    {
        const double* r = mp.ll_tab + 3 * t;
        const double* q = mp.e_tab + 4 * t;
        ll[0] = r[0]; ll[1] = r[1]; ll[2] = r[2];
        e[0] = q[0]; e[1] = q[1]; e[2] = q[2]; mref = q[3];
    }
    __device__ static __forceinline__ int weight_index(value_t s) { return (int)s; }
    struct Rand { uint32_t w[4]; };
    __device__ static __forceinline__ void draw4(uint64_t seed, uint64_t pid0, int t, Rand& r) { draw_words4(seed, pid0, (uint64_t)t, r.w); }
    __device__ static __forceinline__ void propagate4(const ModelParams& mp, uint64_t seed, uint64_t pid0, int t, const value_t (&prev)[4],
                                                      value_t (&x)[4])
    {
        Rand r; draw4(seed, pid0, t, r); apply4(mp, t, r, prev, x);
    }
    __device__ static __forceinline__ void apply4(const ModelParams& mp, int t, const Rand& r, const value_t (&prev)[4], value_t (&x)[4])
    {
        const uint32_t (&w)[4] = r.w;
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = (value_t)smallint_from_word(w[k], 0, 2);   // uniform_smallint{0,2}   :126-127
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                                                  // discrete_distribution{T[state]} :135-136
                const uint64_t c0 = prev[k] == 0 ? mp.hmm_thr[0][0] : (prev[k] == 1 ? mp.hmm_thr[1][0] : mp.hmm_thr[2][0]);
                const uint64_t c1 = prev[k] == 0 ? mp.hmm_thr[0][1] : (prev[k] == 1 ? mp.hmm_thr[1][1] : mp.hmm_thr[2][1]);
                x[k] = (value_t)(((uint64_t)w[k] >= c0) + ((uint64_t)w[k] >= c1));
            }
        }
    }
    // The same transition with the rows' thresholds staged in LDS ({c0, c1} of row s at words 2s, 2s + 1; visible to the caller's
    // threads): one 16-byte LDS read per particle selects the row, where the register form above selects six 64-bit scalars
    // through execution-mask branches (~50 scalar instructions per particle in the step kernel's build).
    static constexpr int kStagedWords = 6;
    __device__ static __forceinline__ void stage(const ModelParams& mp, uint64_t* lds)       // one thread
    {
        lds[0] = mp.hmm_thr[0][0]; lds[1] = mp.hmm_thr[0][1];
        lds[2] = mp.hmm_thr[1][0]; lds[3] = mp.hmm_thr[1][1];
        lds[4] = mp.hmm_thr[2][0]; lds[5] = mp.hmm_thr[2][1];
    }
    __device__ static __forceinline__ void apply4_staged(const uint64_t* lds, int t, const Rand& r, const value_t (&prev)[4], value_t (&x)[4])
    {
        const uint32_t (&w)[4] = r.w;
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = (value_t)smallint_from_word(w[k], 0, 2);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const ulonglong2 c = *reinterpret_cast<const ulonglong2*>(lds + 2 * prev[k]);
                x[k] = (value_t)(((uint64_t)w[k] >= c.x) + ((uint64_t)w[k] >= c.y));
            }
        }
    }
    static constexpr bool kBounded = false;
    __device__ static __forceinline__ double logw_of_z(const ModelParams&, int, double) { return 0.0; }
    __device__ static __forceinline__ double loglik(const ModelParams& mp, value_t s, int t, const double* __restrict__ /*obs*/)
    {
        // log N(y_t; state_mean[s], 1) from the per-run table: three values per step    :130-131,138-139
        const double* row = mp.ll_tab + 3 * t;
        const double l0 = row[0], l1 = row[1], l2 = row[2];
        return s == 0 ? l0 : (s == 1 ? l1 : l2);
    }
    __device__ static __forceinline__ void accumulate(value_t x, double w, double (&acc)[kStats])
    {
        acc[0] += x == 0 ? w : 0.0;
        acc[1] += x == 1 ? w : 0.0;
        acc[2] += x == 2 ? w : 0.0;
    }
};

// The body of include/models/models.hpp:114-141 over a caller-given table: k states (2..8), uniform initial state, emission
// N(mean[s], 1), transition rows as weights (normalised like discrete_distribution does); predict(state, "State") before the
// observe.  Rows and emission log-densities live in device tables (a run-time index into kernel arguments would go through
// scratch memory).  Its steps run on fixed-point weights (step_fixed.hpp): bit-exact index work for any number of states.
struct ModelHmmK {
    using value_t = int32_t;
    using store_t = int8_t;
    static constexpr bool kIsInt = true;
    static constexpr int kStats = 8;
    static constexpr int kWeightTable = 0;        // (the prefix-count form is specialised for three values: models.hpp ModelHmm3)
    __device__ static __forceinline__ void weight_table(const ModelParams&, int, double (&)[1], double (&)[1], double&) {}
    __device__ static __forceinline__ int weight_index(value_t) { return 0; }
    struct Rand { uint32_t w[4]; };
    __device__ static __forceinline__ void draw4(uint64_t seed, uint64_t pid0, int t, Rand& r) { draw_words4(seed, pid0, (uint64_t)t, r.w); }
    __device__ static __forceinline__ void apply4(const ModelParams& mp, int t, const Rand& r, const value_t (&prev)[4], value_t (&x)[4])
    {
        if (t == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = (value_t)smallint_from_word(r.w[k], 0, (uint64_t)mp.hk - 1);      // uniform_smallint{0, k-1}
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t* row = mp.hk_thr + 8 * prev[k];
                int v = 0;
                for (int j = 0; j + 1 < mp.hk; ++j) v += (uint64_t)r.w[k] >= row[j] ? 1 : 0;                      // discrete_distribution{T[state]}
                x[k] = v;
            }
        }
    }
    __device__ static __forceinline__ void propagate4(const ModelParams& mp, uint64_t seed, uint64_t pid0, int t, const value_t (&prev)[4], value_t (&x)[4])
    {
        Rand r; draw4(seed, pid0, t, r); apply4(mp, t, r, prev, x);
    }
    static constexpr bool kBounded = false;
    __device__ static __forceinline__ double logw_of_z(const ModelParams&, int, double) { return 0.0; }
    __device__ static __forceinline__ double loglik(const ModelParams& mp, value_t s, int t, const double* __restrict__ /*obs*/) { return mp.hk_ll[8 * t + s]; }
    __device__ static __forceinline__ void accumulate(value_t x, double w, double (&acc)[kStats])
    {
#pragma unroll
        for (int s = 0; s < kStats; ++s) acc[s] += x == s ? w : 0.0;
    }
};

}  // namespace cph