/*
 * cpprob_oracle.c -- CPU restatement of the cpprob::inference(sis) hot path of
 * lezcano/CPProb, plus the SMC extension the reference never shipped.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may build, load or call it.  The product
 * (cpprob_amd/, include/) never links, imports or falls back to it.
 *
 * Parity status
 *   - logpdf(normal/uniform_smallint/discrete/uniform_real/poisson): PINNED against
 *     the grid of the reference's own test tests/cpprob/logpdf.cpp:23-35,61-78
 *     (tests/golden/logpdf_grid.npz, closed form from scipy) -- tests/test_oracle.py.
 *   - estimators (logsumexp / mean / variance / distribution): PINNED against the
 *     analytic posteriors the reference publishes (README.md:118, thesis p.85).
 *   - text dump grammar: PINNED against the reference's own printer/parser
 *     (include/cpprob/serialization.hpp compiled into oracle/_ref/).
 *   - sampler streams: UNPINNABLE.  The reference draws from Boost.Random 1.66
 *     (un-vendored, .travis.yml:74) on a std::mt19937 seeded from random_device
 *     (src/cpprob/utils.cpp:16-20); no seed exists in its API.  The generators
 *     below are a counter-based replacement (Philox4x32-10, rocRAND-compatible
 *     layout) pinned against rocRAND's own host-callable engine
 *     (tests/golden/philox_rocrand.json) and distributionally against the
 *     exact laws.
 *   - SMC: the reference has none (include/cpprob/state.hpp:28-33).  The driver
 *     follows thesis Alg. 1 p.36 / ESS p.37 and is pinned against exact
 *     posteriors (forward-backward, Kalman/RTS) in oracle/exact.py.
 *
 * Everything is strict IEEE fp64, compiled with -ffp-contract=off.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* Model ids, algorithms and resamplers (mirror include/cpprob_hip.h)         */
/* ------------------------------------------------------------------------- */
enum { ORC_MODEL_GAUSSIAN_UNKNOWN_MEAN = 0, /* include/models/models.hpp:22-35 */
       ORC_MODEL_GAUSSIAN_README = 1,       /* src/models/gaussian.cpp:6-17    */
       ORC_MODEL_LINEAR_GAUSSIAN_1D = 2,    /* include/models/models.hpp:67-80 */
       ORC_MODEL_HMM3 = 3,                  /* include/models/models.hpp:114-141 */
       ORC_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN = 4, /* include/models/models.hpp:38-49 (vector-valued statements) */
       ORC_MODEL_HMM_TABLE = 5 };           /* the body of models.hpp:114-141 with a caller-given table: k states (2..8), means, transition rows (orc_set_hmm) */
enum { ORC_RESAMPLE_SYSTEMATIC = 0, ORC_RESAMPLE_STRATIFIED = 1, ORC_RESAMPLE_MULTINOMIAL = 2,
       ORC_RESAMPLE_MULTINOMIAL_LITERAL = 3 }; /* (oracle-side name of CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL: one threshold per output against the whole CDF) */

#define ORC_RESAMPLE_DRAW_BASE (1ull << 40) /* draw index of the resampling uniforms */
#define ORC_RESAMPLE_DRAW_BASE2 ((1ull << 40) + (1ull << 39)) /* strata form of multinomial resampling: the outputs' uniforms inside their strata */
#define ORC_RESAMPLE_DRAW_BASE3 ((1ull << 40) + (1ull << 38)) /* ... and the bits that split the thresholds over the strata */
#define ORC_TILE 1024u                      /* sources per tile (cpprob_amd/include/cpprob/detail/wave.hpp: kTile): the strata form has one stratum per tile, rounded up to a power of two */

/* ------------------------------------------------------------------------- */
/* Philox4x32-10 (Salmon et al., SC'11; same constants as Random123/rocRAND)  */
/* Replaces get_rng(), src/cpprob/utils.cpp:16-20 (unseedable mt19937).       */
/* ------------------------------------------------------------------------- */
ORC_API void orc_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4])
{
    uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
    uint32_t k0 = key_in[0], k1 = key_in[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* One 128-bit block per (group, draw): key = seed, counter = (draw index, group id).
 * Identical to rocrand_init(seed, subsequence = group, offset = 4*draw); rocrand4().
 * Neighbouring particles share a block (mirrors cpprob_amd/csrc/rng.hpp):
 *   32-bit variates : word (pid & 3) of block(group = pid >> 2)
 *   normal variates : rocRAND box_muller_double(block(group = pid >> 1)), component pid & 1
 *   53-bit uniforms : words (2(pid&1), 2(pid&1)+1) of block(group = pid >> 1)            */
ORC_API void orc_draw_block(uint64_t seed, uint64_t group, uint64_t draw, uint32_t out[4])
{
    uint32_t ctr[4] = { (uint32_t)draw, (uint32_t)(draw >> 32), (uint32_t)group, (uint32_t)(group >> 32) };
    uint32_t key[2] = { (uint32_t)seed, (uint32_t)(seed >> 32) };
    orc_philox4x32_10(ctr, key, out);
}

static const double TWO_POW_M53 = 1.1102230246251565e-16; /* 2^-53 */
static const double TWO_POW_M32 = 2.3283064365386963e-10; /* 2^-32 */

/* 53-bit integer from two words, rocRAND's uniform_distribution_double(v1, v2). */
static inline uint64_t bits53(uint32_t lo, uint32_t hi) { return (uint64_t)lo | ((uint64_t)(hi >> 11) << 32); }

/* [0, 1) from 53 bits / from 32 bits */
ORC_API double orc_u01_53(uint32_t lo, uint32_t hi) { return (double)bits53(lo, hi) * TWO_POW_M53; }
ORC_API double orc_u01_32(uint32_t w) { return (double)w * TWO_POW_M32; }

/* sin(pi*w), cos(pi*w) for w in (0, 2]: exact range reduction, then libm. */
static double sinpi_02(double w)
{
    double sign = 1.0;
    if (w > 1.0) { w -= 1.0; sign = -1.0; }   /* exact: w in (1,2] */
    if (w > 0.5) w = 1.0 - w;                  /* exact */
    if (w <= 0.25) return sign * sin(M_PI * w);
    return sign * cos(M_PI * (0.5 - w));
}
static double cospi_02(double w)
{
    double sign = 1.0;
    if (w > 1.0) { w -= 1.0; sign = -1.0; }   /* cos(pi(w+1)) = -cos(pi w) */
    if (w > 0.5) { w = 1.0 - w; sign = -sign; } /* cos(pi(1-r)) = -cos(pi r) */
    if (w <= 0.25) return sign * cos(M_PI * w);
    return sign * sin(M_PI * (0.5 - w));
}

/* Both outputs of rocRAND's box_muller_double(uint4):
 * u = 2^-53 + v1*2^-53, v1 = x ^ (y << 21); w = 2^-52 + v2*2^-52, v2 = z ^ (w << 21);
 * s = sqrt(-2 log u); out = (s sin(pi w), s cos(pi w)).
 * Stands in for boost::random::normal_distribution::operator() (row a14). */
ORC_API void orc_box_muller(const uint32_t r[4], double out[2])
{
    uint64_t v1 = (uint64_t)r[0] ^ ((uint64_t)r[1] << 21);
    uint64_t v2 = (uint64_t)r[2] ^ ((uint64_t)r[3] << 21);
    double u = TWO_POW_M53 + (double)v1 * TWO_POW_M53;
    double w = (TWO_POW_M53 * 2.0) + (double)v2 * (TWO_POW_M53 * 2.0);
    double s = sqrt(-2.0 * log(u));
    out[0] = s * sinpi_02(w);
    out[1] = s * cospi_02(w);
}

ORC_API uint32_t orc_draw_word(uint64_t seed, uint64_t pid, uint64_t draw)
{
    uint32_t r[4];
    orc_draw_block(seed, pid >> 2, draw, r);
    return r[pid & 3];
}

ORC_API double orc_draw_std_normal(uint64_t seed, uint64_t pid, uint64_t draw)
{
    uint32_t r[4];
    double z[2];
    orc_draw_block(seed, pid >> 1, draw, r);
    orc_box_muller(r, z);
    return z[pid & 1];
}

ORC_API double orc_draw_u01_53(uint64_t seed, uint64_t pid, uint64_t draw)
{
    uint32_t r[4];
    orc_draw_block(seed, pid >> 1, draw, r);
    return (pid & 1) ? orc_u01_53(r[2], r[3]) : orc_u01_53(r[0], r[1]);
}

ORC_API double orc_draw_normal(uint64_t seed, uint64_t pid, uint64_t draw, double mean, double sigma)
{
    return mean + sigma * orc_draw_std_normal(seed, pid, draw);
}

/* uniform_smallint<size_t>{a, b}: a + floor(word * range / 2^32) */
ORC_API uint64_t orc_draw_smallint(uint64_t seed, uint64_t pid, uint64_t draw, uint64_t a, uint64_t b)
{
    uint64_t range = b - a + 1;
    return a + (((uint64_t)orc_draw_word(seed, pid, draw) * range) >> 32);
}

/* discrete_distribution over k weights: inverse CDF on the normalised cumulative
 * sums, u = word * 2^-32 in [0,1).  (Boost uses an alias table; law is identical.) */
static uint64_t discrete_from_u(double u, const double *w, int k)
{
    double tot = 0.0;
    for (int i = 0; i < k; ++i) tot += w[i];
    double acc = 0.0;
    uint64_t idx = 0;
    for (int i = 0; i < k - 1; ++i) {
        acc += w[i];
        if (u >= acc / tot) idx = (uint64_t)(i + 1);
    }
    return idx;
}

ORC_API uint64_t orc_draw_discrete(uint64_t seed, uint64_t pid, uint64_t draw, const double *w, int k)
{
    return discrete_from_u(orc_u01_32(orc_draw_word(seed, pid, draw)), w, k);
}

/* uniform_real_distribution{a,b}: a + (b-a)*u, u in [0,1) from 53 bits */
ORC_API double orc_draw_uniform_real(uint64_t seed, uint64_t pid, uint64_t draw, double a, double b)
{
    return a + (b - a) * orc_draw_u01_53(seed, pid, draw);
}

/* poisson_distribution{mean}: inversion by sequential search on the 53-bit uniform */
ORC_API int64_t orc_draw_poisson(uint64_t seed, uint64_t pid, uint64_t draw, double mean)
{
    double u = orc_draw_u01_53(seed, pid, draw);
    int64_t k = 0;
    double p = exp(-mean), F = p;
    while (u > F && k < 100000) { ++k; p *= mean / (double)k; F += p; }
    return k;
}

/* ------------------------------------------------------------------------- */
/* logpdf functors                                                            */
/* ------------------------------------------------------------------------- */
/* include/cpprob/distributions/utils_normal_distribution.hpp:20-45 */
ORC_API double orc_normal_logpdf(double x, double mean, double std)
{
    if (std == 0) {                                   /* :28-32 Dirac delta */
        return x == mean ? 0 : -INFINITY;
    }
    if (fabs(x) == INFINITY) {                        /* :34-36 */
        return -INFINITY;
    }
    double result = (x - mean) / std;                 /* :38 */
    result *= result;                                 /* :39 */
    result += log(2 * M_PI * std * std);              /* :40 */
    result *= -0.5;                                   /* :41 */
    return result;
}

/* include/cpprob/distributions/utils_uniform_smallint.hpp:17-27 */
ORC_API double orc_uniform_smallint_logpdf(int64_t x, int64_t a, int64_t b)
{
    if (x < a || x > b) return -INFINITY;
    return -log((double)(b - a) + 1.0);
}

/* include/cpprob/distributions/utils_discrete.hpp:17-27 (probabilities() are normalised) */
ORC_API double orc_discrete_logpdf(int64_t x, const double *w, int k)
{
    if (x < 0 || x > k - 1) return -INFINITY;
    double tot = 0.0;
    for (int i = 0; i < k; ++i) tot += w[i];
    return log(w[x] / tot);
}

/* include/cpprob/distributions/utils_uniform_real.hpp:21-31 */
ORC_API double orc_uniform_real_logpdf(double x, double a, double b)
{
    if (x < a || x > b) return -INFINITY;
    return -log(b - a);
}

/* include/cpprob/distributions/utils_poisson.hpp:17-36 */
ORC_API double orc_poisson_logpdf(int64_t x, double l)
{
    if (l == 0.0) return -INFINITY;
    double ret = (double)x * log(l) - l;
    for (int64_t i = 1; i <= x; ++i) ret -= log((double)i);
    return ret;
}

/* ------------------------------------------------------------------------- */
/* Trace record: what TraceInfer holds for one particle                       */
/* (include/cpprob/trace.hpp:34-63): predict lists + double log_w_ = 0.        */
/* In memory the predict lists become columns: real[t][i] / ints[t][i].       */
/* ------------------------------------------------------------------------- */
typedef struct {
    uint64_t seed, pid;      /* replaces the global mt19937 */
    uint64_t n_sample;       /* ordinal of the next sample statement (draw index) */
    double log_w;            /* trace.hpp:59 */
    double *real; int32_t *ints; /* predict columns of THIS particle (strided) */
    size_t stride; int n_pred;
} orc_trace;

/* cpprob::sample, SIS branch: `return distr(get_rng())`  cpprob.hpp:33-35,72-74 */
static double tr_sample_normal(orc_trace *tr, double mean, double sigma)
{ return orc_draw_normal(tr->seed, tr->pid, tr->n_sample++, mean, sigma); }
static uint64_t tr_sample_smallint(orc_trace *tr, uint64_t a, uint64_t b)
{ return orc_draw_smallint(tr->seed, tr->pid, tr->n_sample++, a, b); }
static uint64_t tr_sample_discrete(orc_trace *tr, const double *w, int k)
{ return orc_draw_discrete(tr->seed, tr->pid, tr->n_sample++, w, k); }

/* cpprob::observe -> StateInfer::increment_log_prob: trace_.log_w_ += logpdf
 * cpprob.hpp:79-90, state.cpp:212-223 */
static void tr_observe_normal(orc_trace *tr, double mean, double sigma, double x)
{ tr->log_w += orc_normal_logpdf(x, mean, sigma); }

/* cpprob::predict -> StateInfer::add_predict (real / int lists) cpprob.hpp:92-98,
 * state.hpp:312-327.  All target models use ONE address, so the k-th hit is column k. */
static void tr_predict_real(orc_trace *tr, double x) { tr->real[(size_t)tr->n_pred++ * tr->stride] = x; }
static void tr_predict_int(orc_trace *tr, uint64_t x) { tr->ints[(size_t)tr->n_pred++ * tr->stride] = (int32_t)x; }

/* ------------------------------------------------------------------------- */
/* Models (row a12)                                                           */
/* ------------------------------------------------------------------------- */
/* include/models/models.hpp:22-35 */
static void model_gaussian_unknown_mean(orc_trace *tr, const double *y)
{
    const double mu = tr_sample_normal(tr, 1, sqrt(5));   /* :26-27 prior {1, sqrt(5)} */
    const double var = sqrt(2);                            /* :28 (used as sigma)      */
    tr_observe_normal(tr, mu, var, y[0]);                  /* :32 */
    tr_observe_normal(tr, mu, var, y[1]);                  /* :33 */
    tr_predict_real(tr, mu);                               /* :34 "Mu" */
}
/* src/models/gaussian.cpp:6-17 */
static void model_gaussian_readme(orc_trace *tr, const double *y)
{
    const double mu0 = 1, sigma0 = 1.5, sigma = 2;         /* :8 */
    const double mu = tr_sample_normal(tr, mu0, sigma0);   /* :10-11 */
    tr_observe_normal(tr, mu, sigma, y[0]);                /* :14 */
    tr_observe_normal(tr, mu, sigma, y[1]);                /* :15 */
    tr_predict_real(tr, mu);                               /* :16 "Mean" */
}
/* include/models/models.hpp:67-80 */
static void model_linear_gaussian_1d(orc_trace *tr, const double *obs, size_t T)
{
    double state = 0;                                      /* :72 */
    for (size_t t = 0; t < T; ++t) {                       /* :73 */
        state = tr_sample_normal(tr, state, 1);            /* :74-75 */
        tr_observe_normal(tr, state, 1, obs[t]);           /* :76-77 */
        tr_predict_real(tr, state);                        /* :78 "State" */
    }
}
/* include/models/models.hpp:114-141 */
static const double HMM_MEAN[3] = { -1, 0, 1 };                               /* :122 */
static const double HMM_T[3][3] = { { 0.1, 0.5, 0.4 }, { 0.2, 0.2, 0.6 }, { 0.15, 0.15, 0.7 } }; /* :123-125 */
/* the same model body over a caller-given table (k states, emission means, transition rows): orc_set_hmm */
static int HMMK_K = 0;
static double HMMK_MEAN[8], HMMK_T[8][8];
ORC_API int orc_set_hmm(int k, const double *means, const double *trans)
{
    if (k < 2 || k > 8) return -1;
    HMMK_K = k;
    for (int s = 0; s < k; ++s) { HMMK_MEAN[s] = means[s]; for (int j = 0; j < k; ++j) HMMK_T[s][j] = trans[s * k + j]; }
    return 0;
}
static int is_hmm(int model) { return model == ORC_MODEL_HMM3 || model == ORC_MODEL_HMM_TABLE; }
static int hmm_k(int model) { return model == ORC_MODEL_HMM3 ? 3 : HMMK_K; }
static void model_hmm_table(orc_trace *tr, const double *obs, size_t T)
{
    uint64_t state = tr_sample_smallint(tr, 0, (uint64_t)HMMK_K - 1);     /* :126-127 with k states */
    tr_predict_int(tr, state);
    tr_observe_normal(tr, HMMK_MEAN[state], 1, obs[0]);
    for (size_t t = 1; t < T; ++t) {
        state = tr_sample_discrete(tr, HMMK_T[state], HMMK_K);
        tr_predict_int(tr, state);
        tr_observe_normal(tr, HMMK_MEAN[state], 1, obs[t]);
    }
}
static void model_hmm3(orc_trace *tr, const double *obs, size_t T)
{
    uint64_t state = tr_sample_smallint(tr, 0, 2);         /* :126-127 */
    tr_predict_int(tr, state);                             /* :128 */
    tr_observe_normal(tr, HMM_MEAN[state], 1, obs[0]);     /* :130-131 */
    for (size_t t = 1; t < T; ++t) {                       /* :134 */
        state = tr_sample_discrete(tr, HMM_T[state], 3);   /* :135-136 */
        tr_predict_int(tr, state);                         /* :137 */
        tr_observe_normal(tr, HMM_MEAN[state], 1, obs[t]); /* :138-139 */
    }
}

/* include/models/models.hpp:38-49: prior = multivariate_normal {{1,2},{sqrt 5, sqrt 3}} (independent components,
 * multivariate_normal.hpp:41-50), ONE vector-valued sample, ONE vector-valued observe, ONE NDArray predict.
 * sample: the components draw from the generator in order (multivariate_normal.hpp:268-274) -> draw ordinals 0, 1;
 * observe: logpdf = sum of the components' normal logpdfs (utils_multivariate_normal.hpp:22-33);
 * predict: NDArray -> the real list (state.hpp:330-337); stored here as D consecutive columns. */
static void model_gaussian_2d_unk_mean(orc_trace *tr, const double *y)
{
    const double m0[2] = { 1, 2 }, s0[2] = { sqrt(5), sqrt(3) };   /* :42 */
    double mu[2];
    for (int d = 0; d < 2; ++d) mu[d] = tr_sample_normal(tr, m0[d], s0[d]);   /* :43 */
    const double var = sqrt(2);                                     /* :44 (used as sigma) */
    for (int d = 0; d < 2; ++d) tr_observe_normal(tr, mu[d], var, y[d]);      /* :46-47 */
    for (int d = 0; d < 2; ++d) tr_predict_real(tr, mu[d]);                   /* :48 "Mu" */
}

static int model_is_int(int model) { return model == ORC_MODEL_HMM3 || model == ORC_MODEL_HMM_TABLE; }
ORC_API int orc_model_num_predicts(int model, size_t n_obs)
{ return (model == ORC_MODEL_GAUSSIAN_UNKNOWN_MEAN || model == ORC_MODEL_GAUSSIAN_README) ? 1 : (int)n_obs; }   /* 2-D model: one hit of 2 columns */

static int run_model(int model, orc_trace *tr, const double *obs, size_t n_obs)
{
    switch (model) {
    case ORC_MODEL_GAUSSIAN_UNKNOWN_MEAN: if (n_obs != 2) return -1; model_gaussian_unknown_mean(tr, obs); return 0;
    case ORC_MODEL_GAUSSIAN_README:       if (n_obs != 2) return -1; model_gaussian_readme(tr, obs); return 0;
    case ORC_MODEL_LINEAR_GAUSSIAN_1D:    if (n_obs < 1) return -1;  model_linear_gaussian_1d(tr, obs, n_obs); return 0;
    case ORC_MODEL_HMM3:                  if (n_obs < 1) return -1;  model_hmm3(tr, obs, n_obs); return 0;
    case ORC_MODEL_HMM_TABLE:             if (n_obs < 1 || HMMK_K < 2) return -1;  model_hmm_table(tr, obs, n_obs); return 0;
    case ORC_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN: if (n_obs != 2) return -1; model_gaussian_2d_unk_mean(tr, obs); return 0;
    }
    return -2;
}

/* ------------------------------------------------------------------------- */
/* cpprob::inference(StateType::sis, ...)  include/cpprob/cpprob.hpp:173-203  */
/* In-memory form: finish_trace() appends to column arrays, not to files.     */
/*   val_real / val_int : [n_pred][n] column-major predict values             */
/*   logw               : [n]                                                 */
/* pid0 = global id of local particle 0 (sharding).                           */
/* ------------------------------------------------------------------------- */
ORC_API int orc_sis(int model, const double *obs, size_t n_obs, uint64_t n, uint64_t seed, uint64_t pid0,
                    double *val_real, int32_t *val_int, double *logw)
{
    for (uint64_t i = 0; i < n; ++i) {                    /* cpprob.hpp:194 */
        orc_trace tr;                                      /* start_trace(): trace_ = TraceInfer() state.cpp:188-191 */
        memset(&tr, 0, sizeof tr);
        tr.seed = seed; tr.pid = pid0 + i; tr.log_w = 0;
        tr.real = val_real ? val_real + i : NULL;
        tr.ints = val_int ? val_int + i : NULL;
        tr.stride = n;
        int rc = run_model(model, &tr, obs, n_obs);        /* call_f_tuple(f, observes) cpprob.hpp:199 */
        if (rc) return rc;
        logw[i] = tr.log_w;                                /* finish_trace() state.cpp:193-202 */
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Faithful on-disk form (row a10): StateInfer::dump_predicts state.cpp:262-267*/
/* `([(id v) (id v) ...] logw)`, std::scientific, precision digits10 = 15;     */
/* grammar serialization.hpp:41-46,71-98.  One open/append/close per particle  */
/* per file, exactly like finish_trace() state.cpp:193-202 (all three files).  */
/* ------------------------------------------------------------------------- */
static void dump_line(const char *path, int n_pred, const double *real, const int32_t *ints, size_t stride, double logw)
{
    FILE *f = fopen(path, "a");                           /* std::ios::app state.cpp:264 */
    if (!f) return;
    fputs("([", f);
    for (int k = 0; k < n_pred; ++k) {
        if (k) fputc(' ', f);
        if (real) fprintf(f, "(0 %.15e)", real[(size_t)k * stride]);
        else      fprintf(f, "(0 %d)", ints[(size_t)k * stride]);
    }
    fprintf(f, "] %.15e)\n", logw);
    fclose(f);
}

ORC_API int orc_sis_faithful(int model, const double *obs, size_t n_obs, uint64_t n, uint64_t seed,
                             const char *file_prefix, const char *address, int progress)
{
    char p_real[4096], p_int[4096], p_any[4096], p_ids[4096];
    snprintf(p_real, sizeof p_real, "%s.real", file_prefix);
    snprintf(p_int, sizeof p_int, "%s.int", file_prefix);
    snprintf(p_any, sizeof p_any, "%s.any", file_prefix);
    snprintf(p_ids, sizeof p_ids, "%s.ids", file_prefix);
    int n_pred = orc_model_num_predicts(model, n_obs);
    double *real = (double *)calloc((size_t)n_pred, sizeof(double));
    int32_t *ints = (int32_t *)calloc((size_t)n_pred, sizeof(int32_t));
    int is_int = model_is_int(model);
    for (uint64_t i = 0; i < n; ++i) {
        if (progress && i % 100 == 0) { printf("Generating trace %llu\n", (unsigned long long)i); fflush(stdout); } /* cpprob.hpp:195-197 (endl flushes) */
        orc_trace tr; memset(&tr, 0, sizeof tr);
        tr.seed = seed; tr.pid = i; tr.real = real; tr.ints = ints; tr.stride = 1;
        int rc = run_model(model, &tr, obs, n_obs);
        if (rc) { free(real); free(ints); return rc; }
        /* finish_trace(): three dumps, empty lists print as `([] logw)` */
        dump_line(p_int, is_int ? n_pred : 0, NULL, ints, 1, tr.log_w);
        dump_line(p_real, is_int ? 0 : n_pred, real, NULL, 1, tr.log_w);
        dump_line(p_any, 0, real, NULL, 1, tr.log_w);
    }
    /* finish_infer(): ids file, then delete the all-empty files state.cpp:164-180 */
    FILE *f = fopen(p_ids, "w");
    if (f) { fprintf(f, "%s\n", address); fclose(f); }
    if (is_int) remove(p_real); else remove(p_int);
    remove(p_any);
    free(real); free(ints);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Estimators (row a13) include/cpprob/postprocess/empirical_distribution.hpp  */
/* ------------------------------------------------------------------------- */
/* logsumexp :125-143 -- max-shifted, sequential accumulate */
ORC_API double orc_logsumexp(const double *logw, uint64_t n)
{
    if (n == 0) return 0.0;                               /* :131-133 value-initialised */
    double max = logw[0];
    for (uint64_t i = 1; i < n; ++i) if (logw[i] > max) max = logw[i];  /* supremum :135 */
    double acc = 0.0;
    for (uint64_t i = 0; i < n; ++i) acc += exp(logw[i] - max);         /* :136-139 */
    return log(acc) + max;                                               /* :140 */
}

/* raw_moment(1), raw_moment(2) - mean^2  :52-81 ; ESS = (sum W^2)^-1 thesis p.37 */
ORC_API void orc_weighted_moments(const double *x, const double *logw, uint64_t n, double out[4])
{
    double log_norm = orc_logsumexp(logw, n);
    double m1 = 0.0, m2 = 0.0, q = 0.0;
    for (uint64_t i = 0; i < n; ++i) {
        double w = exp(logw[i] - log_norm);               /* :63 */
        m1 += w * x[i];
        m2 += w * (x[i] * x[i]);
        q += w * w;
    }
    out[0] = m1;                 /* mean()              :68-71 */
    out[1] = m2 - m1 * m1;       /* variance(mean)      :78-81 */
    out[2] = log_norm;
    out[3] = 1.0 / q;            /* ESS */
}

/* distribution() :30-40 for integer predicts with support {0..k-1} */
ORC_API void orc_weighted_hist(const int32_t *x, const double *logw, uint64_t n, int k, double *out)
{
    double log_norm = orc_logsumexp(logw, n);
    for (int j = 0; j < k; ++j) out[j] = 0.0;
    for (uint64_t i = 0; i < n; ++i)
        if (x[i] >= 0 && x[i] < k) out[x[i]] += exp(logw[i] - log_norm);
}

/* ------------------------------------------------------------------------- */
/* Resampling (row a15; spec: thesis Alg. 1 p.36, remark on systematic p.36)   */
/* weights w_k = exp(logw_k - max), inclusive CDF C_k, total W.                */
/* position of output j:  systematic (j + u0) * W/N, stratified (j + u_j) * W/N,*/
/* multinomial u_j * W.   ancestor a_j = min{k : C_k > p_j} (clamped to N-1).   */
/* u0 = draw(seed, pid 0, RESAMPLE_BASE + step), u_j = draw(seed, pid j, same). */
/* ------------------------------------------------------------------------- */
static uint64_t upper_bound_d(const double *c, uint64_t n, double p)
{
    uint64_t lo = 0, hi = n;          /* first k with c[k] > p */
    while (lo < hi) { uint64_t mid = lo + (hi - lo) / 2; if (c[mid] > p) hi = mid; else lo = mid + 1; }
    return lo < n ? lo : n - 1;
}

/* systematic: one 53-bit uniform of group 0; stratified: 32-bit uniform of output j;
 * multinomial: 53-bit uniform of output j */
static double resample_u0(uint64_t seed, uint64_t step)
{
    uint32_t r[4];
    orc_draw_block(seed, 0, ORC_RESAMPLE_DRAW_BASE + step, r);
    return orc_u01_53(r[0], r[1]);
}

/* n_out outputs [j0, j0+n_out) of a population of n_total_out positions drawn
 * over the n_in weights.  (Sharded runs call this with j0 != 0.) */
ORC_API int orc_resample(int kind, const double *logw, uint64_t n_in, uint64_t seed, uint64_t step,
                         uint64_t j0, uint64_t n_out, uint64_t n_total_out, int32_t *anc, double *cdf_scratch)
{
    double *cdf = cdf_scratch ? cdf_scratch : (double *)malloc(n_in * sizeof(double));
    if (!cdf) return -1;
    double max = logw[0];
    for (uint64_t i = 1; i < n_in; ++i) if (logw[i] > max) max = logw[i];
    double acc = 0.0;
    for (uint64_t i = 0; i < n_in; ++i) { acc += exp(logw[i] - max); cdf[i] = acc; }
    const double W = acc;
    const double step_w = W / (double)n_total_out;
    const double u0 = resample_u0(seed, step);
    for (uint64_t jj = 0; jj < n_out; ++jj) {
        uint64_t j = j0 + jj;
        double p;
        if (kind == ORC_RESAMPLE_SYSTEMATIC) p = ((double)j + u0) * step_w;
        else if (kind == ORC_RESAMPLE_STRATIFIED) p = ((double)j + orc_u01_32(orc_draw_word(seed, j, ORC_RESAMPLE_DRAW_BASE + step))) * step_w;
        else p = orc_draw_u01_53(seed, j, ORC_RESAMPLE_DRAW_BASE + step) * W;
        anc[jj] = (int32_t)upper_bound_d(cdf, n_in, p);
    }
    if (!cdf_scratch) free(cdf);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Systematic resampling of a TABLE-WEIGHT generation, order-independent form. */
/* When every particle entered the step at the same log-weight (t = 0, or the   */
/* previous step resampled) and the model's incremental weight takes one of K   */
/* values (HMM: log N(y_t; mean[s], 1), K = 3; models.hpp:130-131,138-139), the  */
/* weight of particle k is e[x_k] with e[s] = exp(ll_s - max ll), and the        */
/* inclusive CDF is a function of INTEGER prefix counts c_s(k) = #{i <= k:        */
/* x_i = s}:   C_k = fma(c_2, e_2, fma(c_1, e_1, c_0 * e_0))   (this order).       */
/* Output j (position (j + u0) W/N, thesis p.36 remark) descends from            */
/* a_j = min{k : C_k > (j + u0) W/N} = min{k : G_k > j},                          */
/* G_k = ceil(fma(C_k, N/W, -u0)) clamped to [0, N], G of the last source = N.    */
/* No running floating-point sum exists, so any evaluation order -- a serial     */
/* loop here, tiles and wavefronts on the GPU, shards on several GPUs -- yields   */
/* the same integers: this is the form whose ancestors are compared bit for bit. */
/* before[s] = count of state s in the shards that precede this one (0 on one     */
/* GPU), total[s] = count over the whole population; outputs [j0, j0 + n_out).     */
/* Outputs whose ancestor is not among these n_in sources get -1.                  */
/* ------------------------------------------------------------------------- */
static double table_cdf(const uint64_t c[3], const double e[3])
{
    return fma((double)c[2], e[2], fma((double)c[1], e[1], (double)c[0] * e[0]));
}

ORC_API int orc_resample_table_systematic(const int32_t *x, uint64_t n_in, const double e[3], const uint64_t before[3],
                                          const uint64_t total[3], int last_shard, uint64_t seed, uint64_t step,
                                          uint64_t j0, uint64_t n_out, uint64_t n_total_out, int32_t *anc)
{
    const double N = (double)n_total_out;
    const double W = table_cdf(total, e);
    const double inv = N / W;
    const double u0 = resample_u0(seed, step);
    uint64_t c[3] = { before[0], before[1], before[2] };
    for (uint64_t jj = 0; jj < n_out; ++jj) anc[jj] = -1;
    double g_prev = ceil(fma(table_cdf(c, e), inv, -u0));     /* first output owned by this shard's sources */
    if (g_prev < 0.0 || (before[0] + before[1] + before[2]) == 0) g_prev = 0.0;
    if (g_prev > N) g_prev = N;
    for (uint64_t k = 0; k < n_in; ++k) {
        if (x[k] < 0 || x[k] > 2) return -2;
        c[x[k]] += 1;
        double g = ceil(fma(table_cdf(c, e), inv, -u0));
        if (g < 0.0) g = 0.0;
        if (g > N) g = N;
        if (last_shard && k + 1 == n_in) g = N;
        for (double j = g_prev; j < g; j += 1.0) {
            if (j >= (double)j0 && j < (double)(j0 + n_out)) anc[(uint64_t)j - j0] = (int32_t)k;
        }
        if (g > g_prev) g_prev = g;
    }
    return 0;
}

/* The same generation under STRATIFIED resampling (positions j + u_j, u_j the   */
/* 32-bit uniform of output j): the sources up to CDF value C_k reach            */
/* H_k = C_k * (N / W) (one rounded product) and own the outputs with            */
/* j + u_j < H_k, a prefix of A_k = F + [u_F < H_k - F] outputs, F = floor(H_k)  */
/* (orc_resample_fixed_stratified with the table CDF in place of the mass).      */
static double table_stratified_first(const uint64_t c[3], const double e[3], double inv, double N, uint64_t seed, uint64_t step)
{
    const double H = table_cdf(c, e) * inv;
    const double F = floor(H);
    if (F >= N) return N;
    const double u = orc_u01_32(orc_draw_word(seed, (uint64_t)F, ORC_RESAMPLE_DRAW_BASE + step));
    return u < H - F ? F + 1.0 : F;
}

ORC_API int orc_resample_table_stratified(const int32_t *x, uint64_t n_in, const double e[3], const uint64_t before[3],
                                          const uint64_t total[3], int last_shard, uint64_t seed, uint64_t step,
                                          uint64_t j0, uint64_t n_out, uint64_t n_total_out, int32_t *anc)
{
    const double N = (double)n_total_out;
    const double inv = N / table_cdf(total, e);
    uint64_t c[3] = { before[0], before[1], before[2] };
    for (uint64_t jj = 0; jj < n_out; ++jj) anc[jj] = -1;
    double g_prev = table_stratified_first(c, e, inv, N, seed, step);
    for (uint64_t k = 0; k < n_in; ++k) {
        if (x[k] < 0 || x[k] > 2) return -2;
        c[x[k]] += 1;
        double g = table_stratified_first(c, e, inv, N, seed, step);
        if (last_shard && k + 1 == n_in) g = N;
        for (double j = g_prev; j < g; j += 1.0)
            if (j >= (double)j0 && j < (double)(j0 + n_out)) anc[(uint64_t)j - j0] = (int32_t)k;
        if (g > g_prev) g_prev = g;
    }
    return 0;
}

/* ... and under MULTINOMIAL resampling, strata form (orc_resample_fixed_multinomial_strata below with the table CDF in place of  */
/* the integer mass): the strata's bounds are B_w = w * (W 2^-k) (exact scaling, one rounded product; B_K = W), output s of        */
/* stratum w takes tau_s = fma(v_s, B_w+1 - B_w, B_w) with v_s = the 53-bit uniform of output s, ancestor = min{k : C_k > tau_s}   */
/* (the population's last particle if no C_k exceeds it: tau_s may round up to W).  The counts per stratum: orc_multinomial_strata. */
ORC_API int orc_strata_levels(uint64_t n_particles);
ORC_API void orc_multinomial_strata(uint64_t seed, uint64_t step, uint64_t n_out, int k, uint32_t *offs);
ORC_API int orc_resample_table_multinomial(const int32_t *x, uint64_t n_in, const double e[3], uint64_t seed, uint64_t step, uint64_t n_out, int32_t *anc)
{
    const int k = orc_strata_levels(n_in);
    const uint64_t K = (uint64_t)1 << k;
    double *cdf = (double *)malloc((n_in ? n_in : 1) * sizeof(double));
    uint32_t *offs = (uint32_t *)malloc((K + 1) * sizeof(uint32_t));
    if (!cdf || !offs) { free(cdf); free(offs); return -1; }
    uint64_t c[3] = { 0, 0, 0 };
    for (uint64_t i = 0; i < n_in; ++i) {
        if (x[i] < 0 || x[i] > 2) { free(cdf); free(offs); return -2; }
        c[x[i]] += 1;
        cdf[i] = table_cdf(c, e);
    }
    const double W = table_cdf(c, e);
    const double unit = ldexp(W, -k);
    orc_multinomial_strata(seed, step, n_out, k, offs);
    for (uint64_t w = 0; w < K; ++w) {
        const double b0 = (double)w * unit, b1 = (double)(w + 1) * unit;
        for (uint64_t s = offs[w]; s < offs[w + 1]; ++s) {
            uint32_t r[4];
            orc_draw_block(seed, s >> 1, ORC_RESAMPLE_DRAW_BASE2 + step, r);
            const double v = (s & 1) ? orc_u01_53(r[2], r[3]) : orc_u01_53(r[0], r[1]);
            const double tau = fma(v, b1 - b0, b0);
            uint64_t lo = 0, hi = n_in;               /* first i with cdf[i] > tau */
            while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (cdf[mid] > tau) hi = mid; else lo = mid + 1; }
            anc[s] = (int32_t)(lo < n_in ? lo : n_in - 1);
        }
    }
    free(cdf); free(offs);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Systematic resampling on FIXED-POINT weights, order-independent form for     */
/* continuous weights and ESS-triggered schedules (the build's own arithmetic:   */
/* the reference has no SMC; cpprob_amd/csrc/step_fixed.hpp).          */
/* The linear weight of particle i is the integer                                */
/*     q_i = min(rint(exp(lw_i - R) * 2^32), 2^32 - 1),                           */
/* R >= max lw known before the generation exists: R_t = B_t when every particle  */
/* enters step t at log-weight 0 (t = 0, or the previous step resampled), else    */
/* R_t = M_{t-1} + B_t, with B_t the upper bound of the step's incremental        */
/* log-weight (the emission's density at its mode, models.hpp:76-77,130-139) and  */
/* M_{t-1} the exact maximum of the previous log-weights.  Inclusive CDF          */
/* C_k = sum_{i<=k} q_i (exact, 64-bit), G_k = ceil(fma((double)C_k, N/(double)C_N,*/
/* -u0)), ancestor of output j = min{k : G_k > j}, G of the last source = N;       */
/* W = C_N 2^-32, ESS = min(N, W^2 / (2^-32 sum floor(floor(q_i / 2^8)^2 / 2^16))).  Integers sum exactly in */
/* any order: tiles, wavefronts and shards all produce these ancestors.            */
/* exp() is the kernel's own range-specific form (fma arithmetic, so that the      */
/* integers agree bit for bit): cpprob/detail/fastmath.hpp exp_nonpos.             */
/* ------------------------------------------------------------------------- */
static double orc_exp_nonpos(double x)
{
    static const double c[11] = { 0x1.1f8b4cd99e7aap-29, 0x1.af4dea2bc3f25p-26, 0x1.27e4cccda6fcfp-22, 0x1.71de023137276p-19, 0x1.a01a01acfae99p-16,
                                  0x1.a01a01abe8206p-13, 0x1.6c16c16c151fcp-10, 0x1.11111111100dbp-7, 0x1.5555555555558p-5, 0x1.5555555555557p-3, 0.5 };
    const double k = rint(x * 0x1.71547652b82fep+0);
    double r = fma(-k, 0x1.62e42fefa3800p-1, x);
    r = fma(-k, 0x1.ef35793c76730p-45, r);
    double p = c[0];
    for (int i = 1; i < 11; ++i) p = fma(p, r, c[i]);
    p = fma(p, r * r, r);
    return ldexp(p + 1.0, (int)k);
}

/* a weight's square for the ESS: cpprob/detail/fixed_mass.hpp fix_square */
static uint64_t orc_fix_square(uint32_t q)
{
    const uint64_t h = q >> 8;
    return (h * h) >> 16;
}

ORC_API uint32_t orc_fix_weight(double lw, double ref)
{
    double d = lw - ref;
    if (!(d > -1000.0)) d = -1000.0;                    /* -inf (and the clamp of the kernel): an exact 0 */
    const double s = rint(orc_exp_nonpos(d) * 4294967296.0);
    return s >= 4294967295.0 ? 0xffffffffu : (uint32_t)s;
}

ORC_API void orc_fix_weights(const double *logw, uint64_t n, double ref, uint32_t *q)
{
    for (uint64_t i = 0; i < n; ++i) q[i] = orc_fix_weight(logw[i], ref);
}

/* before = mass of the shards that precede this one (0 on one GPU), total = mass of the whole population;
 * outputs [j0, j0 + n_out); outputs whose ancestor is not among these n_in sources get -1. */
ORC_API int orc_resample_fixed_systematic(const uint32_t *q, uint64_t n_in, uint64_t before, uint64_t total, int last_shard,
                                          uint64_t seed, uint64_t step, uint64_t j0, uint64_t n_out, uint64_t n_total_out, int32_t *anc)
{
    const double N = (double)n_total_out;
    const double inv = N / (double)total;
    const double u0 = resample_u0(seed, step);
    uint64_t c = before;
    for (uint64_t jj = 0; jj < n_out; ++jj) anc[jj] = -1;
    double g_prev = ceil(fma((double)c, inv, -u0));
    if (g_prev < 0.0 || before == 0) g_prev = 0.0;
    if (g_prev > N) g_prev = N;
    for (uint64_t k = 0; k < n_in; ++k) {
        c += q[k];
        double g = ceil(fma((double)c, inv, -u0));
        if (g < 0.0) g = 0.0;
        if (g > N) g = N;
        if (last_shard && k + 1 == n_in) g = N;
        for (double j = g_prev; j < g; j += 1.0)
            if (j >= (double)j0 && j < (double)(j0 + n_out)) anc[(uint64_t)j - j0] = (int32_t)k;
        if (g > g_prev) g_prev = g;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Stratified resampling on the same integer masses (thesis p.36 remark; the   */
/* build's own arithmetic, cpprob_amd/include/cpprob/detail/fixed_mass.hpp).    */
/* Output j sits at position j + u_j, u_j = word_j 2^-32 the 32-bit uniform of  */
/* OUTPUT j (word j & 3 of block j >> 2, draw RESAMPLE_BASE + step), and source */
/* k reaches up to H_k = double(C_k) * (N / double(C_N)) (one rounded product). */
/* j + u_j increases with j, so the outputs below H_k form a prefix:            */
/*     A_k = #{j : j + u_j < H_k} = F + [u_F < H_k - F],  F = floor(H_k)         */
/* (H_k - F is exact, the comparison is exact), ancestor of j = min{k : A_k > j},*/
/* A of the population's last source = N.  A_k is a function of the exact        */
/* integer C_k alone: tiles, wavefronts and shards may take the sources in any   */
/* order.  The sequential-CDF form this restates: orc_resample(STRATIFIED).      */
/* ------------------------------------------------------------------------- */
static double fixed_stratified_first(uint64_t c, double inv, double N, uint64_t seed, uint64_t step)
{
    const double H = (double)c * inv;
    const double F = floor(H);
    if (F >= N) return N;
    const double u = orc_u01_32(orc_draw_word(seed, (uint64_t)F, ORC_RESAMPLE_DRAW_BASE + step));
    return u < H - F ? F + 1.0 : F;
}

ORC_API int orc_resample_fixed_stratified(const uint32_t *q, uint64_t n_in, uint64_t before, uint64_t total, int last_shard,
                                          uint64_t seed, uint64_t step, uint64_t j0, uint64_t n_out, uint64_t n_total_out, int32_t *anc)
{
    const double N = (double)n_total_out;
    const double inv = N / (double)total;
    uint64_t c = before;
    for (uint64_t jj = 0; jj < n_out; ++jj) anc[jj] = -1;
    double g_prev = fixed_stratified_first(c, inv, N, seed, step);
    for (uint64_t k = 0; k < n_in; ++k) {
        c += q[k];
        double g = fixed_stratified_first(c, inv, N, seed, step);
        if (last_shard && k + 1 == n_in) g = N;
        for (double j = g_prev; j < g; j += 1.0)
            if (j >= (double)j0 && j < (double)(j0 + n_out)) anc[(uint64_t)j - j0] = (int32_t)k;
        if (g > g_prev) g_prev = g;
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Multinomial resampling (thesis Alg. 1 p.36: a_j ~ Categorical(W)) on the     */
/* same integer masses.  Output j draws the 53-bit uniform of OUTPUT j           */
/* (orc_draw_u01_53's words: u = b 2^-53) and its threshold is the integer       */
/*     tau_j = floor(u C_N) = floor(b C_N / 2^53)     (a 128-bit product),        */
/* ancestor of j = min{k : C_k > tau_j}: no rounding anywhere, tau_j < C_N       */
/* always.  before = mass of the shards that precede these n_in sources; outputs */
/* whose ancestor is not among them get -1.  The sequential-CDF form:            */
/* orc_resample(MULTINOMIAL).                                                    */
/* ------------------------------------------------------------------------- */
ORC_API uint64_t orc_multinomial_threshold(uint64_t seed, uint64_t j, uint64_t step, uint64_t total)
{
    uint32_t r[4];
    orc_draw_block(seed, j >> 1, ORC_RESAMPLE_DRAW_BASE + step, r);
    const uint64_t b = (j & 1) ? bits53(r[2], r[3]) : bits53(r[0], r[1]);
    return (uint64_t)(((unsigned __int128)(b << 11) * total) >> 64);
}

ORC_API int orc_resample_fixed_multinomial(const uint32_t *q, uint64_t n_in, uint64_t before, uint64_t total,
                                           uint64_t seed, uint64_t step, uint64_t j0, uint64_t n_out, int32_t *anc)
{
    uint64_t *cdf = (uint64_t *)malloc((n_in ? n_in : 1) * sizeof(uint64_t));
    if (!cdf) return -1;
    uint64_t c = before;
    for (uint64_t k = 0; k < n_in; ++k) { c += q[k]; cdf[k] = c; }
    for (uint64_t jj = 0; jj < n_out; ++jj) {
        const uint64_t tau = orc_multinomial_threshold(seed, j0 + jj, step, total);
        anc[jj] = -1;
        if (tau < before || tau >= c) continue;
        uint64_t lo = 0, hi = n_in;               /* first k with cdf[k] > tau */
        while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (cdf[mid] > tau) hi = mid; else lo = mid + 1; }
        anc[jj] = (int32_t)lo;
    }
    free(cdf);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Multinomial resampling, STRATA form: the same law -- N iid uniform           */
/* thresholds on [0, C_N) -- generated in nearly sorted order, so that an        */
/* output's ancestor sits next to it as under systematic resampling.  The form    */
/* the device runs by default (csrc/step_fixed.hpp: multinomial_strata_kernel,   */
/* strata_walk); the literal form above is N searches of the whole population    */
/* at random addresses.                                                          */
/*   N iid uniforms = how many fall into each of K = 2^k equal strata            */
/*   (m_w) ~ Multinomial(N; 1/K, .., 1/K), and iid uniforms inside each stratum.  */
/*   1  the counts do not depend on the weights: a binary tree over the strata,   */
/*      the n thresholds of a node go left with probability 1/2 each --           */
/*      left = popcount of the first n bits of the node's own Philox stream       */
/*      (block group = (2^l + i) << 32 | chunk, draw BASE3 + step): exact          */
/*      Binomial(n, 1/2) in integers.  o_w = m_0 + .. + m_w-1.                     */
/*   2  output s in [o_w, o_w+1) draws the 53-bit uniform v_s of OUTPUT s          */
/*      (draw BASE2 + step) and its threshold is                                   */
/*          tau_s = B_w + floor(v_s (B_w+1 - B_w)),  B_w = floor(C_N w / K),        */
/*      ancestor = min{k : C_k > tau_s}.                                          */
/* K = the smallest power of two >= FOUR times the number of 1024-source tiles (a */
/* stratum holds 128 .. 256 thresholds on average).  Integers throughout.         */
/* ------------------------------------------------------------------------- */
ORC_API int orc_strata_levels(uint64_t n_particles)
{
    const uint64_t nb = (n_particles + ORC_TILE - 1) / ORC_TILE;
    int k = 0;
    while (((uint64_t)1 << k) < nb) ++k;
    return k + 2;
}

/* popcount of the first n bits of the stream of tree node `node` (heap order: 2^l + i): words x, y, z, w of block (node << 32 | chunk), low bits first */
static uint64_t strata_left(uint64_t seed, uint64_t draw, uint64_t node, uint64_t n)
{
    uint64_t left = 0;
    for (uint64_t chunk = 0; chunk * 128 < n; ++chunk) {
        uint32_t r[4];
        orc_draw_block(seed, (node << 32) | chunk, draw, r);
        uint64_t rem = n - chunk * 128;
        for (int j = 0; j < 4 && rem > 0; ++j) {
            const uint32_t take = rem >= 32 ? 32u : (uint32_t)rem;
            const uint32_t m = take == 32 ? 0xffffffffu : ((1u << take) - 1u);
            left += (uint64_t)__builtin_popcount(r[j] & m);
            rem -= take;
        }
    }
    return left;
}

/* offs[0 .. K]: first output of every stratum's thresholds (offs[K] = n_out).
 * k <= 6: one tree over all n_out thresholds.  Beyond, two parts (the sum of independent multinomial counts is multinomial):
 *   top     the thresholds are dealt to G = clamp(K / 64 / 2, 1, 64) groups of consecutive outputs [g n / G, (g + 1) n / G); every group
 *           sends its own through the top 6 levels of a tree of ITS OWN (stream of node `node` of group g: block group
 *           1 << 63 | g << 40 | node << 32 | chunk); the groups' counts of the 64 level-6 nodes are added;
 *   bottom  each level-6 node splits its total down the remaining k - 6 levels (streams by heap index, as in the one-tree form).
 * (That is how the device spreads the work over the chip: groups x steps workgroups for the top, 64 x steps for the bottom.) */
#define ORC_STRATA_TOP 6
ORC_API int orc_strata_groups(int k)
{
    if (k <= ORC_STRATA_TOP) return 1;
    const int64_t g = ((int64_t)1 << k) / 128;
    return (int)(g < 1 ? 1 : (g > 64 ? 64 : g));
}

static void strata_split(uint64_t seed, uint64_t draw, uint64_t key_hi, uint64_t heap0, int levels, uint64_t stride, uint32_t *cnt)
{
    /* cnt[0] holds the root's count; node i of level l sits at index i * (stride >> l); heap index of the root: heap0 */
    for (int l = 0; l < levels; ++l) {
        const uint64_t span = stride >> l, half = span >> 1;
        for (uint64_t i = 0; i < ((uint64_t)1 << l); ++i) {
            const uint64_t n = cnt[i * span];
            const uint64_t left = strata_left(seed, draw, key_hi | ((heap0 << l) + i), n);
            cnt[i * span] = (uint32_t)left;
            cnt[i * span + half] = (uint32_t)(n - left);
        }
    }
}

ORC_API void orc_multinomial_strata(uint64_t seed, uint64_t step, uint64_t n_out, int k, uint32_t *offs)
{
    const uint64_t K = (uint64_t)1 << k;
    const uint64_t draw = ORC_RESAMPLE_DRAW_BASE3 + step;
    if (k <= ORC_STRATA_TOP) {
        offs[0] = (uint32_t)n_out;
        strata_split(seed, draw, 0, 1, k, K, offs);
    } else {
        const int G = orc_strata_groups(k);
        const uint64_t sub = K >> ORC_STRATA_TOP;               /* strata below a level-6 node */
        uint32_t top[64], acc[64];
        memset(acc, 0, sizeof acc);
        for (int g = 0; g < G; ++g) {
            const uint64_t n_g = (uint64_t)(((unsigned __int128)n_out * (uint64_t)(g + 1)) / (uint64_t)G) - (uint64_t)(((unsigned __int128)n_out * (uint64_t)g) / (uint64_t)G);
            memset(top, 0, sizeof top);
            top[0] = (uint32_t)n_g;
            strata_split(seed, draw, ((uint64_t)1 << 31) | ((uint64_t)g << 8), 1, ORC_STRATA_TOP, 64, top);
            for (int i = 0; i < 64; ++i) acc[i] += top[i];
        }
        for (uint64_t i = 0; i < 64; ++i) {
            offs[i * sub] = acc[i];
            strata_split(seed, draw, 0, 64 + i, k - ORC_STRATA_TOP, sub, offs + i * sub);
        }
    }
    uint64_t a = 0;
    for (uint64_t w = 0; w < K; ++w) { const uint64_t m = offs[w]; offs[w] = (uint32_t)a; a += m; }
    offs[K] = (uint32_t)a;
}

static uint64_t strata_bound(uint64_t total, uint64_t w, int k) { return (uint64_t)(((unsigned __int128)total * w) >> k); }

ORC_API int orc_resample_fixed_multinomial_strata(const uint32_t *q, uint64_t n_in, uint64_t seed, uint64_t step, uint64_t n_out, int32_t *anc)
{
    const int k = orc_strata_levels(n_in);
    const uint64_t K = (uint64_t)1 << k;
    uint64_t *cdf = (uint64_t *)malloc((n_in ? n_in : 1) * sizeof(uint64_t));
    uint32_t *offs = (uint32_t *)malloc((K + 1) * sizeof(uint32_t));
    if (!cdf || !offs) { free(cdf); free(offs); return -1; }
    uint64_t c = 0;
    for (uint64_t i = 0; i < n_in; ++i) { c += q[i]; cdf[i] = c; }
    const uint64_t total = c;
    orc_multinomial_strata(seed, step, n_out, k, offs);
    for (uint64_t w = 0; w < K; ++w) {
        const uint64_t b0 = strata_bound(total, w, k), b1 = strata_bound(total, w + 1, k);
        for (uint64_t s = offs[w]; s < offs[w + 1]; ++s) {
            uint32_t r[4];
            orc_draw_block(seed, s >> 1, ORC_RESAMPLE_DRAW_BASE2 + step, r);
            const uint64_t v = (s & 1) ? bits53(r[2], r[3]) : bits53(r[0], r[1]);
            const uint64_t tau = b0 + (uint64_t)(((unsigned __int128)(v << 11) * (b1 - b0)) >> 64);
            uint64_t lo = 0, hi = n_in;               /* first i with cdf[i] > tau */
            while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (cdf[mid] > tau) hi = mid; else lo = mid + 1; }
            anc[s] = (int32_t)(lo < n_in ? lo : n_in - 1);
        }
    }
    free(cdf); free(offs);
    return 0;
}

/* The strata form over SHARDS of one population (exchange scope; thesis Alg. 1 p.36 drawn by several devices): the strata, their
 * counts and every output's uniform are the POPULATION's (k from n_pop, output ids population-wide), so the threshold of output s is
 * the same integer on whichever rank looks at it; these n_in sources hold the mass range [before, before + own) of `total`, and
 * output s descends from them iff before <= tau_s < before + own -- its ancestor the first local k with before + C_k > tau_s.
 * anc[0 .. n_pop): -1 where the ancestor is another shard's.  One shard with before = 0 is orc_resample_fixed_multinomial_strata. */
ORC_API int orc_resample_fixed_multinomial_strata_shard(const uint32_t *q, uint64_t n_in, uint64_t before, uint64_t total, uint64_t seed, uint64_t step,
                                                        uint64_t n_pop, int32_t *anc)
{
    const int k = orc_strata_levels(n_pop);
    const uint64_t K = (uint64_t)1 << k;
    uint64_t *cdf = (uint64_t *)malloc((n_in ? n_in : 1) * sizeof(uint64_t));
    uint32_t *offs = (uint32_t *)malloc((K + 1) * sizeof(uint32_t));
    if (!cdf || !offs) { free(cdf); free(offs); return -1; }
    uint64_t c = before;
    for (uint64_t i = 0; i < n_in; ++i) { c += q[i]; cdf[i] = c; }
    const uint64_t end = c;
    if (end > total) { free(cdf); free(offs); return -2; }
    orc_multinomial_strata(seed, step, n_pop, k, offs);
    for (uint64_t w = 0; w < K; ++w) {
        const uint64_t b0 = strata_bound(total, w, k), b1 = strata_bound(total, w + 1, k);
        for (uint64_t s = offs[w]; s < offs[w + 1]; ++s) {
            uint32_t r[4];
            orc_draw_block(seed, s >> 1, ORC_RESAMPLE_DRAW_BASE2 + step, r);
            const uint64_t v = (s & 1) ? bits53(r[2], r[3]) : bits53(r[0], r[1]);
            const uint64_t tau = b0 + (uint64_t)(((unsigned __int128)(v << 11) * (b1 - b0)) >> 64);
            if (tau < before || tau >= end) { anc[s] = -1; continue; }
            uint64_t lo = 0, hi = n_in;               /* first i with cdf[i] > tau (exists: cdf[n_in - 1] = end > tau) */
            while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (cdf[mid] > tau) hi = mid; else lo = mid + 1; }
            anc[s] = (int32_t)lo;
        }
    }
    free(cdf); free(offs);
    return 0;
}

/* ... and the table form over shards: before[s] / total[s] = counts of state s in the shards that precede this one / in the whole
 * population (orc_resample_table_systematic's arguments); the CDF values are table_cdf of GLOBAL prefix counts, the rank's range is
 * [table_cdf(before), table_cdf(before + own)) -- the population's last shard also takes the thresholds that round up to W. */
ORC_API int orc_resample_table_multinomial_shard(const int32_t *x, uint64_t n_in, const double e[3], const uint64_t before[3], const uint64_t total[3], int last_shard,
                                                 uint64_t seed, uint64_t step, uint64_t n_pop, int32_t *anc)
{
    const int k = orc_strata_levels(n_pop);
    const uint64_t K = (uint64_t)1 << k;
    double *cdf = (double *)malloc((n_in ? n_in : 1) * sizeof(double));
    uint32_t *offs = (uint32_t *)malloc((K + 1) * sizeof(uint32_t));
    if (!cdf || !offs) { free(cdf); free(offs); return -1; }
    uint64_t c[3] = { before[0], before[1], before[2] };
    const double c_lo = table_cdf(c, e);
    for (uint64_t i = 0; i < n_in; ++i) {
        if (x[i] < 0 || x[i] > 2) { free(cdf); free(offs); return -2; }
        c[x[i]] += 1;
        cdf[i] = table_cdf(c, e);
    }
    const double c_hi = table_cdf(c, e);
    const double W = table_cdf(total, e);
    const double unit = ldexp(W, -k);
    orc_multinomial_strata(seed, step, n_pop, k, offs);
    for (uint64_t w = 0; w < K; ++w) {
        const double b0 = (double)w * unit, b1 = (double)(w + 1) * unit;
        for (uint64_t s = offs[w]; s < offs[w + 1]; ++s) {
            uint32_t r[4];
            orc_draw_block(seed, s >> 1, ORC_RESAMPLE_DRAW_BASE2 + step, r);
            const double v = (s & 1) ? orc_u01_53(r[2], r[3]) : orc_u01_53(r[0], r[1]);
            const double tau = fma(v, b1 - b0, b0);
            if (tau < c_lo || (!last_shard && tau >= c_hi)) { anc[s] = -1; continue; }
            uint64_t lo = 0, hi = n_in;
            while (lo < hi) { const uint64_t mid = lo + (hi - lo) / 2; if (cdf[mid] > tau) hi = mid; else lo = mid + 1; }
            anc[s] = (int32_t)(lo < n_in ? lo : n_in - 1);
        }
    }
    free(cdf); free(offs);
    return 0;
}

/* The thresholds themselves (tests of the exchange plan: which rank's mass range holds output s's threshold): tau[0 .. n_pop) as the
 * two forms above draw them -- integers against a total mass, doubles against a table CDF's W -- and the stratum of every output. */
ORC_API int orc_strata_thresholds_fixed(uint64_t total, uint64_t seed, uint64_t step, uint64_t n_pop, uint64_t *tau, int32_t *stratum)
{
    const int k = orc_strata_levels(n_pop);
    const uint64_t K = (uint64_t)1 << k;
    uint32_t *offs = (uint32_t *)malloc((K + 1) * sizeof(uint32_t));
    if (!offs) return -1;
    orc_multinomial_strata(seed, step, n_pop, k, offs);
    for (uint64_t w = 0; w < K; ++w) {
        const uint64_t b0 = strata_bound(total, w, k), b1 = strata_bound(total, w + 1, k);
        for (uint64_t s = offs[w]; s < offs[w + 1]; ++s) {
            uint32_t r[4];
            orc_draw_block(seed, s >> 1, ORC_RESAMPLE_DRAW_BASE2 + step, r);
            const uint64_t v = (s & 1) ? bits53(r[2], r[3]) : bits53(r[0], r[1]);
            tau[s] = b0 + (uint64_t)(((unsigned __int128)(v << 11) * (b1 - b0)) >> 64);
            if (stratum) stratum[s] = (int32_t)w;
        }
    }
    free(offs);
    return 0;
}
ORC_API int orc_strata_thresholds_table(double W, uint64_t seed, uint64_t step, uint64_t n_pop, double *tau, int32_t *stratum)
{
    const int k = orc_strata_levels(n_pop);
    const uint64_t K = (uint64_t)1 << k;
    uint32_t *offs = (uint32_t *)malloc((K + 1) * sizeof(uint32_t));
    if (!offs) return -1;
    const double unit = ldexp(W, -k);
    orc_multinomial_strata(seed, step, n_pop, k, offs);
    for (uint64_t w = 0; w < K; ++w) {
        const double b0 = (double)w * unit, b1 = (double)(w + 1) * unit;
        for (uint64_t s = offs[w]; s < offs[w + 1]; ++s) {
            uint32_t r[4];
            orc_draw_block(seed, s >> 1, ORC_RESAMPLE_DRAW_BASE2 + step, r);
            const double v = (s & 1) ? orc_u01_53(r[2], r[3]) : orc_u01_53(r[0], r[1]);
            tau[s] = fma(v, b1 - b0, b0);
            if (stratum) stratum[s] = (int32_t)w;
        }
    }
    free(offs);
    return 0;
}
ORC_API double orc_table_cdf(const uint64_t c[3], const double e[3]) { return table_cdf(c, e); }

/* e[s] = exp(ll_s - max ll) of step t of the HMM: the table the weights of generation t are drawn from */
static void hmm_weight_table(double y, double e[3], double *mref)
{
    double l[3], mx;
    for (int s = 0; s < 3; ++s) l[s] = orc_normal_logpdf(y, HMM_MEAN[s], 1);
    mx = l[0] > l[1] ? l[0] : l[1];
    if (l[2] > mx) mx = l[2];
    for (int s = 0; s < 3; ++s) e[s] = exp(l[s] - mx);
    if (mref) *mref = mx;
}

/* ------------------------------------------------------------------------- */
/* SMC driver (row a15).  Markov step form of the three state-space models:   */
/* step t: x_t ~ p(.|x_{t-1}) [sample #t], predict, logw += log p(y_t|x_t).    */
/* After weighting step t (t < T-1): ESS_t = W^2/Q; resample iff ESS_t <       */
/* ess_frac*N (ess_frac > 1: every step); then logw <- 0 and                   */
/* logZ += max + log(W/N).  No resampling after the last step.                 */
/* hist_real/hist_int [T][n]: value of predict t in slot i of generation t.    */
/* hist_anc [T][n]: slot of generation t-1 that slot i of generation t extends */
/* (row 0 = identity).                                                         */
/* ------------------------------------------------------------------------- */
/* filter_stats (optional, [T][K]): predict hit t under generation t's OWN weights -- P(x_t = s) (HMM, K = 3) or {mean, variance}
 * (K = 2) -- what a filtering-only run (keep_history = 0) reports instead of the whole-trace posterior. */
/* ref_mode (cpprob_amd/include/cpprob/gpu.hpp: the unchanged-model path's step forms):
 *   0  as above: the model's own bound (table form for the 3-state HMM on an every-step schedule);
 *   1  fixed-point form for every model, B_t = the observe statement's density at its mode, logpdf(N(m, 1), m) -- what the host's
 *      structural dry run records for `observe(normal_distribution<>{m, 1}, y_t)` (models.hpp:76-77,138-139);
 *   2  fixed-point form for every model, R_t = the generation's exact maximum (no bound: cpprob_hip_smc_bookkeep_fixed,
 *      cpprob_hip_generic_quantize);
 *   3  the floating-point form of every resampler (orc_resample: a sequential fp64 CDF) -- what CPPROB_HIP_FLAG_FLOATING_POINT_STEP
 *      runs, up to the summation order of its parallel scan. */
static int orc_smc_impl(int model, const double *obs, size_t T, uint64_t n, uint64_t seed,
                        int resampler, double ess_frac,
                        double *hist_real, int32_t *hist_int, int32_t *hist_anc,
                        double *logw_final, double *log_z, double *ess_trace, int32_t *resampled, double *filter_stats, int ref_mode)
{
    if (model != ORC_MODEL_LINEAR_GAUSSIAN_1D && !is_hmm(model)) return -2;
    if (is_hmm(model) != (hist_int != NULL)) return -3;
    if (model == ORC_MODEL_HMM_TABLE && HMMK_K < 2) return -5;
    const int K = is_hmm(model) ? hmm_k(model) : 0;
    const double *hmean = model == ORC_MODEL_HMM3 ? HMM_MEAN : HMMK_MEAN;
    double *logw = (double *)calloc(n, sizeof(double));
    double *cdf = (double *)malloc(n * sizeof(double));
    int32_t *anc = (int32_t *)malloc(n * sizeof(int32_t));
    if (!logw || !cdf || !anc) return -1;
    double lz = 0.0;
    int do_resample = 0;
    /* every resampler runs on integer masses (the fixed-point forms above) -- except systematic / stratified resampling of the 3-state
     * HMM on an every-step schedule, which run on integer prefix COUNTS (the table forms), and ref_mode 3, the floating-point CDF */
    const int table = (resampler == ORC_RESAMPLE_SYSTEMATIC || resampler == ORC_RESAMPLE_STRATIFIED || resampler == ORC_RESAMPLE_MULTINOMIAL) &&
                      model == ORC_MODEL_HMM3 && ess_frac > 1.0 && ref_mode == 0;
    const int fixed = ref_mode != 3 && !table;
    uint32_t *qw = fixed ? (uint32_t *)malloc(n * sizeof(uint32_t)) : NULL;
    uint64_t q_total = 0;
    double m_prev = 0.0;
    for (size_t t = 0; t < T; ++t) {
        if (do_resample && fixed) {
            int rc;
            if (resampler == ORC_RESAMPLE_SYSTEMATIC) rc = orc_resample_fixed_systematic(qw, n, 0, q_total, 1, seed, (uint64_t)t, 0, n, n, anc);
            else if (resampler == ORC_RESAMPLE_STRATIFIED) rc = orc_resample_fixed_stratified(qw, n, 0, q_total, 1, seed, (uint64_t)t, 0, n, n, anc);
            else if (resampler == ORC_RESAMPLE_MULTINOMIAL) rc = orc_resample_fixed_multinomial_strata(qw, n, seed, (uint64_t)t, n, anc);
            else rc = orc_resample_fixed_multinomial(qw, n, 0, q_total, seed, (uint64_t)t, 0, n, anc);
            if (rc) return -4;
        } else if (do_resample && table) {
            /* every step resamples: generation t-1 carries table weights -> the order-independent form */
            double e[3];
            uint64_t before[3] = { 0, 0, 0 }, total[3] = { 0, 0, 0 };
            hmm_weight_table(obs[t - 1], e, NULL);
            for (uint64_t i = 0; i < n; ++i) total[hist_int[(t - 1) * n + i]] += 1;
            if (resampler == ORC_RESAMPLE_SYSTEMATIC ? orc_resample_table_systematic(hist_int + (t - 1) * n, n, e, before, total, 1, seed, (uint64_t)t, 0, n, n, anc)
              : resampler == ORC_RESAMPLE_STRATIFIED ? orc_resample_table_stratified(hist_int + (t - 1) * n, n, e, before, total, 1, seed, (uint64_t)t, 0, n, n, anc)
                                                      : orc_resample_table_multinomial(hist_int + (t - 1) * n, n, e, seed, (uint64_t)t, n, anc)) return -4;
        } else if (do_resample) {
            orc_resample(resampler == ORC_RESAMPLE_MULTINOMIAL_LITERAL ? ORC_RESAMPLE_MULTINOMIAL : resampler, logw, n, seed, (uint64_t)t, 0, n, n, anc, cdf);
        } else {
            for (uint64_t i = 0; i < n; ++i) anc[i] = (int32_t)i;
        }
        for (uint64_t i = 0; i < n; ++i) {
            const uint64_t a = (uint64_t)anc[i];
            hist_anc[t * n + i] = anc[i];
            double lw = do_resample ? 0.0 : logw[a];      /* a == i when not resampling */
            if (model == ORC_MODEL_LINEAR_GAUSSIAN_1D) {
                double prev = t == 0 ? 0.0 : hist_real[(t - 1) * n + a];
                double x = orc_draw_normal(seed, i, (uint64_t)t, prev, 1);
                hist_real[t * n + i] = x;
                lw += orc_normal_logpdf(obs[t], x, 1);
            } else {
                uint64_t s;
                if (t == 0) s = orc_draw_smallint(seed, i, 0, 0, (uint64_t)K - 1);
                else s = orc_draw_discrete(seed, i, (uint64_t)t, model == ORC_MODEL_HMM3 ? HMM_T[hist_int[(t - 1) * n + a]] : HMMK_T[hist_int[(t - 1) * n + a]], K);
                hist_int[t * n + i] = (int32_t)s;
                lw += orc_normal_logpdf(obs[t], hmean[s], 1);
            }
            cdf[i] = lw;                                  /* staging: new logw */
        }
        memcpy(logw, cdf, n * sizeof(double));
        /* weights of generation t */
        double max = logw[0];
        for (uint64_t i = 1; i < n; ++i) if (logw[i] > max) max = logw[i];
        double W = 0.0, Q = 0.0;
        for (uint64_t i = 0; i < n; ++i) { double w = exp(logw[i] - max); W += w; Q += w * w; }
        double ess = W * W / Q;
        if (fixed) {
            /* reference known before the generation exists; integer weights, masses and squares */
            double bound;
            if (model == ORC_MODEL_LINEAR_GAUSSIAN_1D) bound = orc_normal_logpdf(obs[t], obs[t], 1);
            else { bound = orc_normal_logpdf(obs[t], hmean[0], 1); for (int s2 = 1; s2 < K; ++s2) { const double l = orc_normal_logpdf(obs[t], hmean[s2], 1); if (l > bound) bound = l; } }
            if (ref_mode == 1) bound = orc_normal_logpdf(0.0, 0.0, 1);
            double ref = (t == 0 || do_resample) ? bound : m_prev + bound;
            if (ref_mode == 2) ref = max;
            /* a generation whose heaviest particle sits more than 6 nats below the reference known in advance keeps too few of its 32
             * bits: it is weighed against its exact maximum instead (cpprob_hip.hip: settle_fixed / repair_fixed_generation) */
            if (ref - max > 6.0 && max > -INFINITY) ref = max;
            uint64_t S = 0, Q16 = 0;
            for (uint64_t i = 0; i < n; ++i) { qw[i] = orc_fix_weight(logw[i], ref); S += qw[i]; Q16 += orc_fix_square(qw[i]); }
            q_total = S;
            m_prev = max;
            max = ref;                                   /* sums below are relative to exp(ref) */
            W = (double)S * (1.0 / 4294967296.0);
            Q = (double)Q16 * (1.0 / 4294967296.0);
            ess = W * W / Q;
            if (ess > (double)n) ess = (double)n;        /* the floored squares under-count Q: the estimate is kept in [.., N] */
        }
        if (ess_trace) ess_trace[t] = ess;
        if (filter_stats) {
            if (is_hmm(model)) {
                double acc[8] = { 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0 };
                for (uint64_t i = 0; i < n; ++i) acc[hist_int[t * n + i]] += fixed ? (double)qw[i] * (1.0 / 4294967296.0) : exp(logw[i] - max);
                for (int k = 0; k < K; ++k) filter_stats[t * K + k] = acc[k] / W;
            } else {
                double s1 = 0.0, s2 = 0.0;
                for (uint64_t i = 0; i < n; ++i) { const double w = fixed ? (double)qw[i] * (1.0 / 4294967296.0) : exp(logw[i] - max), x = hist_real[t * n + i]; s1 += w * x; s2 += w * x * x; }
                filter_stats[t * 2] = s1 / W;
                filter_stats[t * 2 + 1] = s2 / W - (s1 / W) * (s1 / W);
            }
        }
        do_resample = (t + 1 < T) && (ess < ess_frac * (double)n);
        if (resampled) resampled[t] = do_resample;
        if (do_resample) lz += max + log(W / (double)n);
        else if (t + 1 == T) lz += max + log(W / (double)n);
    }
    memcpy(logw_final, logw, n * sizeof(double));
    if (log_z) *log_z = lz;
    free(logw); free(cdf); free(anc); free(qw);
    return 0;
}

ORC_API int orc_smc(int model, const double *obs, size_t T, uint64_t n, uint64_t seed,
                    int resampler, double ess_frac,
                    double *hist_real, int32_t *hist_int, int32_t *hist_anc,
                    double *logw_final, double *log_z, double *ess_trace, int32_t *resampled)
{
    return orc_smc_impl(model, obs, T, n, seed, resampler, ess_frac, hist_real, hist_int, hist_anc, logw_final, log_z, ess_trace, resampled, NULL, 0);
}

ORC_API int orc_smc_ref(int model, const double *obs, size_t T, uint64_t n, uint64_t seed,
                        int resampler, double ess_frac, int ref_mode,
                        double *hist_real, int32_t *hist_int, int32_t *hist_anc,
                        double *logw_final, double *log_z, double *ess_trace, int32_t *resampled)
{
    if (ref_mode < 0 || ref_mode > 3) return -6;
    return orc_smc_impl(model, obs, T, n, seed, resampler, ess_frac, hist_real, hist_int, hist_anc, logw_final, log_z, ess_trace, resampled, NULL, ref_mode);
}

ORC_API int orc_smc_filter(int model, const double *obs, size_t T, uint64_t n, uint64_t seed,
                           int resampler, double ess_frac,
                           double *hist_real, int32_t *hist_int, int32_t *hist_anc,
                           double *logw_final, double *log_z, double *ess_trace, int32_t *resampled, double *filter_stats)
{
    return orc_smc_impl(model, obs, T, n, seed, resampler, ess_frac, hist_real, hist_int, hist_anc, logw_final, log_z, ess_trace, resampled, filter_stats, 0);
}

/* Lineage read-out: path[t][i] = slot of generation t on the ancestral line of
 * final particle i.  This is what makes each surviving particle a full trace
 * (list of T predicts + one weight) as in TraceInfer / stats_printer.hpp:106-118. */
ORC_API void orc_trace_lineage(const int32_t *hist_anc, size_t T, uint64_t n, int32_t *path)
{
    for (uint64_t i = 0; i < n; ++i) path[(T - 1) * n + i] = (int32_t)i;
    for (size_t t = T - 1; t > 0; --t)
        for (uint64_t i = 0; i < n; ++i)
            path[(t - 1) * n + i] = hist_anc[t * n + (uint64_t)path[t * n + i]];
}

/* Smoothing estimators over lineages: per predict hit t, StatsPrinter's numbers.
 * real: out[t*2+0] = mean, out[t*2+1] = variance; int: out[t*k + s] = P(x_t = s). */
ORC_API void orc_smoothing_real(const double *hist_real, const int32_t *hist_anc, const double *logw,
                                size_t T, uint64_t n, double *out)
{
    int32_t *path = (int32_t *)malloc(T * n * sizeof(int32_t));
    double *col = (double *)malloc(n * sizeof(double));
    orc_trace_lineage(hist_anc, T, n, path);
    for (size_t t = 0; t < T; ++t) {
        for (uint64_t i = 0; i < n; ++i) col[i] = hist_real[t * n + (uint64_t)path[t * n + i]];
        double m[4];
        orc_weighted_moments(col, logw, n, m);
        out[t * 2] = m[0]; out[t * 2 + 1] = m[1];
    }
    free(path); free(col);
}

ORC_API void orc_smoothing_int(const int32_t *hist_int, const int32_t *hist_anc, const double *logw,
                               size_t T, uint64_t n, int k, double *out)
{
    int32_t *path = (int32_t *)malloc(T * n * sizeof(int32_t));
    int32_t *col = (int32_t *)malloc(n * sizeof(int32_t));
    orc_trace_lineage(hist_anc, T, n, path);
    for (size_t t = 0; t < T; ++t) {
        for (uint64_t i = 0; i < n; ++i) col[i] = hist_int[t * n + (uint64_t)path[t * n + i]];
        orc_weighted_hist(col, logw, n, k, out + t * k);
    }
    free(path); free(col);
}
