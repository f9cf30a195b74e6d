/*
 * cpprob_hip.h -- C ABI of the MI355X (gfx950) SIS / SMC inference engine.
 *
 * Drop-in boundary for ONE path of lezcano/CPProb: cpprob::inference(StateType::sis, ...)
 * (reference include/cpprob/cpprob.hpp:173-203) and the StateType::smc mode the reference never
 * shipped.  The reference has no FFI for this path -- it is a header-level C++14 template API
 *  -- so the entry points below are what the C++14 compatibility layer
 * in cpprob_amd/include/cpprob/ binds, and what a maintainer of the reference would call from
 * cpprob::inference instead of the per-particle loop.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only.  No C++/torch types.
 *   - Every function returns 0 on success, a negative CPPROB_HIP_E* code otherwise;
 *     cpprob_hip_last_error(ctx) then describes the failure.  Library code never exits
 *     (the reference's exit()/terminate() paths, SURVEY section 5, are not reproduced).
 *   - A context owns one device, one HIP stream and all device buffers of a run.  No process
 *     globals (the reference's State::state_, StateInfer::trace_, TraceInfer::ids_predict_,
 *     src/cpprob/state.cpp:20-21,148-155, are per-context here).  One run at a time per context;
 *     contexts are independent and may live on different threads.
 *   - The context's stream is NON-blocking (no implicit ordering with the null stream or any other stream): d_* buffers the
 *     caller produced elsewhere must be complete before the call that reads them.
 *   - Pointers named d_* are DEVICE pointers valid on the context's device, h_* are host
 *     pointers.  Work is enqueued on the context's stream; functions that fill host memory
 *     synchronise that stream, all others are asynchronous.
 *   - All arithmetic is fp64 (reference: double log_w_, include/cpprob/trace.hpp:59); discrete
 *     values and ancestor indices are int32.
 */
#ifndef CPPROB_HIP_H_
#define CPPROB_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

#define CPPROB_HIP_ABI_VERSION 3

/* error codes */
#define CPPROB_HIP_OK 0
#define CPPROB_HIP_EINVAL (-1)     /* bad argument                                   */
#define CPPROB_HIP_EDEVICE (-2)    /* HIP runtime error (no GPU, launch failure, OOM) */
#define CPPROB_HIP_ESTATE (-3)     /* call out of order (e.g. stats before run)       */
#define CPPROB_HIP_EUNSUPPORTED (-4)
#define CPPROB_HIP_EPRECISION (-5)  /* a caller-driven sharded run (step protocol) left the fixed-point weights too few bits: repeat it
                                        with CPPROB_HIP_FLAG_FLOATING_POINT_STEP (runs the library drives itself are repeated by it) */

/* StateType -- reference include/cpprob/state.hpp:28-33 {compile, csis, sis, dryrun}; smc is new. */
#define CPPROB_HIP_ALG_SIS 2
#define CPPROB_HIP_ALG_SMC 4

/* Built-in models: hand-fused kernels for the models of namespace models on this path. */
#define CPPROB_HIP_MODEL_GAUSSIAN_UNKNOWN_MEAN 0 /* include/models/models.hpp:22-35  obs (y1,y2), predict "Mu"    */
#define CPPROB_HIP_MODEL_GAUSSIAN_README 1       /* src/models/gaussian.cpp:6-17     obs (x1,x2), predict "Mean"  */
#define CPPROB_HIP_MODEL_LINEAR_GAUSSIAN_1D 2    /* include/models/models.hpp:67-80  obs[T],      predict "State" */
#define CPPROB_HIP_MODEL_HMM3 3                  /* include/models/models.hpp:114-141 obs[T],     predict "State" */
#define CPPROB_HIP_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN 4 /* include/models/models.hpp:38-49 vector-valued statements: one multivariate
                                                    normal sample, one vector observe y[2], one NDArray predict "Mu".  The D = 2
                                                    components are rows of the particle store (variable-width SoA):
                                                    n_predict = 2 = one predict hit of width 2, stats row d = component d.
                                                    One observe statement => smc runs as sis.                             */

#define CPPROB_HIP_MODEL_HMM_TABLE 5             /* the model body of include/models/models.hpp:114-141 over a caller-given table
                                                    (cpprob_hip_set_hmm before cpprob_hip_infer_begin): k states, 2 <= k <= 8, uniform
                                                    initial state, emission N(mean[s], 1), transition row s as weights; predict "State"
                                                    before every observe.  stats_per_predict = 8 (P(x_t = s), zeros beyond k)        */

/* Resamplers (thesis Alg. 1 p.36: multinomial; remark p.36: systematic / stratified).  One population per context: all three run on
 * integers inside the step launch (prefix counts or fixed-point masses), bit-identical over tilings.
 *   SYSTEMATIC   one shared offset u0: output j at j + u0;
 *   STRATIFIED   output j at j + u_j, u_j the 32-bit uniform of output j;
 *   MULTINOMIAL  a_j ~ Categorical(W) iid, evaluated in two stages so that offspring stay next to their parent (strata form): the N
 *                iid thresholds as counts per equal stratum of the total mass (K = 2^k strata, K >= four times the 1024-particle tiles;
 *                exact Binomial splits down a tree of Philox popcounts: independent of the weights, drawn once per run) and iid
 *                53-bit uniforms inside a stratum: output s of stratum w takes tau_s = B_w + floor(v_s (B_w+1 - B_w)), its ancestor
 *                is min{k : C_k > tau_s}  (CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL: one threshold floor(u_j C_N) per output searched
 *                against the whole population -- the same law, N scattered searches). */
#define CPPROB_HIP_RESAMPLE_SYSTEMATIC 0
#define CPPROB_HIP_RESAMPLE_STRATIFIED 1
#define CPPROB_HIP_RESAMPLE_MULTINOMIAL 2

#define CPPROB_HIP_SCOPE_GLOBAL 0
#define CPPROB_HIP_SCOPE_ISLAND 1
#define CPPROB_HIP_SCOPE_EXCHANGE 2   /* one population, resampled jointly AND exactly: offspring of remote sources migrate (below) */

typedef struct cpprob_hip_ctx cpprob_hip_ctx;

/* Run configuration.  Replaces the arguments of cpprob::inference (cpprob.hpp:173-180):
 * algorithm <- StateType, model <- const Func& f, n_particles <- std::size_t n; the observes
 * tuple is passed flattened to cpprob_hip_infer_begin.  The reference has no seed (its RNG is
 * a random_device-seeded global, src/cpprob/utils.cpp:16-20); here runs are reproducible. */
typedef struct cpprob_hip_config {
    int32_t algorithm;        /* CPPROB_HIP_ALG_*                                              */
    int32_t model;            /* CPPROB_HIP_MODEL_*                                            */
    int32_t resampler;        /* CPPROB_HIP_RESAMPLE_* (SMC only)                              */
    int32_t resample_scope;   /* CPPROB_HIP_SCOPE_GLOBAL: one population of n_global particles,
                                 resampled jointly (sharded runs use the step_begin/step_end
                                 protocol); CPPROB_HIP_SCOPE_ISLAND: this shard is an independent
                                 population of n_particles (particle_offset only selects the
                                 RNG streams); shards are combined by their evidence estimates;
                                 CPPROB_HIP_SCOPE_EXCHANGE: like GLOBAL, with particle migration
                                 (cpprob_hip_exchange_*) so that resampling is exact over shards */
    int32_t keep_history;     /* 1: keep per-step values + ancestors: statistics over whole traces (what the reference's
                                 posterior files hold), dumps.  0 (SMC; one population per context, or a shard of a joint
                                 population in the exchange scope, whose migrants are then their current state alone): filtering only --
                                 two rows of values, no ancestors (memory O(N) instead of O(N T)); predict hit t's
                                 statistics are those of generation t under its own weights; cpprob_hip_copy_values /
                                 _ancestors / _paths return CPPROB_HIP_ESTATE                                    */
    int32_t annex_kcols;      /* exchange scope: immigrant-annex capacity in units of 1024 columns per row (0 = default: the larger of
                                 1/16 of the shard and sqrt(n_global) x T -- what a well-mixed run's O(sqrt N) immigrants per step
                                 add up to --, at least 4096); stream-ordered runs cannot grow it mid-run and report overflow instead */
    uint32_t flags;           /* CPPROB_HIP_FLAG_* below; 0 = the measured optimum.  Every switch that selects another kernel form
                                 lives HERE, in the caller's hands -- the library reads no environment variable */
    int32_t fuse_max_tiles;   /* floating-point step: largest population (in 1024-particle tiles) whose step kernel normalises the
                                 previous generation in its own prologue; 0 = default (1664), capped there */
    double ess_threshold;     /* SMC: resample after a step iff ESS < ess_threshold * N_global;
                                 > 1 resamples after every step (thesis p.37 uses 0.5)        */
    uint64_t seed;            /* Philox key                                                    */
    uint64_t n_particles;     /* particles held by THIS context (the shard)                    */
    uint64_t particle_offset; /* global id of local particle 0 (RNG counters use global ids)   */
    uint64_t n_global;        /* total particles over all shards (= n_particles for 1 GPU)     */
} cpprob_hip_config;

/* cpprob_hip_config::flags -- A/B and diagnostic forms (results agree within the stated tolerances) */
#define CPPROB_HIP_FLAG_FLOATING_POINT_STEP 1u   /* table-weight models on an every-step schedule: the floating-point step instead of the
                                                    integer prefix-count step (ancestors then agree up to CDF-boundary flips, not bit for bit) */
#define CPPROB_HIP_FLAG_NO_SKIP_ROWS 2u          /* exchange scope, long traces: extract migrating lineages hop by hop */
#define CPPROB_HIP_FLAG_SIS_PER_TILE 4u          /* SIS of bounded-weight models through the per-tile kernel */
#define CPPROB_HIP_FLAG_SIS_SEPARATE_READOUT 8u  /* SIS read-out as a pass over the particle store instead of riding the normalisation */
#define CPPROB_HIP_FLAG_WREL_STORED 16u          /* floating-point step of table-weight models: read stored linear weights, not states */
#define CPPROB_HIP_FLAG_FP_TILE_PARTIALS 32u     /* ... and fp64 tile partials instead of packed per-value counts */
#define CPPROB_HIP_FLAG_WALK_READOUT 64u         /* short discrete traces: read the posterior out by the lineage walk, not from trace words */
#define CPPROB_HIP_FLAG_PAIRED_STEP_LAUNCH 512u  /* fixed-point form, ESS-triggered schedules (A/B): a step as two launches {carry, resampling} of which one ends at once */
#define CPPROB_HIP_FLAG_REPEAT_IN_FLOATING_POINT 256u /* fixed-point form: a run with a generation that lost its bits is repeated whole in the floating-point form
                                                        (the r03 / r04 behaviour) instead of being repaired from that generation on, in integers */
#define CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL 128u /* multinomial resampling: ancestor of output j = min{k : C_k > floor(u_j C_N)}, one search per output */

/* Posterior summary of a finished run -- what StatsPrinter prints
 * (include/cpprob/postprocess/stats_printer.hpp:42-79) plus SMC diagnostics. */
typedef struct cpprob_hip_summary {
    double log_evidence;  /* log Z-hat: logsumexp(logw) - log N for SIS; SMC product estimate     */
    double ess_final;     /* (sum W_i^2)^-1 of the final weights, thesis p.37                      */
    double log_norm;      /* logsumexp of the final log-weights (EmpiricalDistribution :117-123)  */
    double max_logw;      /* reference of the final weights: their max, or (continuous-weight models; bounded-weight SIS: an upper bound of it)
                             the smallest multiple of ln 2 above it; sums are relative to exp(max_logw) */
    int32_t n_predict;    /* predict hits per trace (T)                                           */
    int32_t stats_per_predict; /* 2 for real predicts (mean, variance); k for int predicts (P(x=s)) */
    int32_t is_int;       /* 1: predicts are integral (.int file), 0: real (.real)                */
    int32_t n_resampled;  /* number of steps after which resampling happened                      */
    int32_t step_form;    /* arithmetic the run's steps ran in: CPPROB_HIP_FORM_*                          */
    int32_t n_requantised; /* fixed-point form: generations whose weights were taken again against their exact maximum, because their heaviest
                              particle sat more than 6 nats below the reference known in advance (the steps behind such a generation ran again) */
} cpprob_hip_summary;
#define CPPROB_HIP_FORM_FLOAT 0   /* fp64 linear weights, floating-point CDF (stratified / multinomial resampling, locally resampled
                                     shards, SIS, runs repeated because the fixed-point weights lost their bits)          */
#define CPPROB_HIP_FORM_COUNTS 1  /* integer prefix counts of the states (table-weight models, every-step schedule)        */
#define CPPROB_HIP_FORM_FIXED 2   /* fixed-point weights q = rint(exp(lw - max_logw) 2^32), exact 64-bit prefix masses     */

/* ---- lifecycle ------------------------------------------------------------------------- */
int cpprob_hip_abi_version(void);
/* Hash of the sources this binary was built from (cpprob_amd/build.py embeds it; "unknown" for a hand-rolled build):
 * lets a loader refuse a stale libcpprob_hip.so that is newer than, but different from, the sources next to it. */
const char* cpprob_hip_build_id(void);
/* Number of visible HIP devices, or a negative error code (no GPU -> CPPROB_HIP_EDEVICE). */
int cpprob_hip_device_count(void);
/* Creates a context on `device` with its own stream.  Fails loudly without a GPU. */
int cpprob_hip_create(int device, cpprob_hip_ctx** out);
void cpprob_hip_destroy(cpprob_hip_ctx* ctx);
const char* cpprob_hip_last_error(const cpprob_hip_ctx* ctx); /* ctx may be NULL: global message */
/* The context's hipStream_t, for event timing / interop by the caller. */
void* cpprob_hip_stream(cpprob_hip_ctx* ctx);
int cpprob_hip_sync(cpprob_hip_ctx* ctx);

/* ---- cpprob::inference ------------------------------------------------------------------
 * begin : State::set + StateInfer::start_infer + config_file (cpprob.hpp:184-192): validates,
 *         (re)allocates the particle store in HBM and uploads the observes.
 * run   : the whole `for (i < n)` loop of cpprob.hpp:194-201 for all particles at once, plus
 *         (SMC) weight normalisation, ESS test and resampling between observes, plus the
 *         posterior read-out (stats_printer.hpp:88-120 / empirical_distribution.hpp:30-81).
 *         Asynchronous; may be called repeatedly (each call is one full run; pass run_index to
 *         decorrelate runs: the Philox key used is seed + run_index).
 * summary / stats : finish_infer + StatsPrinter: synchronise and copy results to the host.
 *         h_stats receives n_predict * stats_per_predict doubles, row t = predict hit t:
 *         real -> {mean, variance} (raw_moment(2) - mean^2, empirical_distribution.hpp:78-81);
 *         int  -> {P(x_t = 0), ..., P(x_t = k-1)} (distribution(), :30-40).
 */
/* The three read-backs below in ONE call and behind ONE stream synchronisation (pinned staging): what cpprob::inference reads when a
 * run is over.  h_stats ([n_predict * stats_per_predict], n_doubles its capacity), h_ess and h_resampled ([n_predict]) may be NULL.
 * Same values, same errors as cpprob_hip_infer_summary / _stats / _step_trace. */
int cpprob_hip_infer_results(cpprob_hip_ctx* ctx, cpprob_hip_summary* out, double* h_stats, size_t n_doubles, double* h_ess, int32_t* h_resampled);
/* The table of CPPROB_HIP_MODEL_HMM_TABLE: h_means[k], h_transition[k * k] (row s = the weights discrete_distribution{T[s]} of
 * models.hpp:135 would be built from).  Kept by the context until the next call. */
int cpprob_hip_set_hmm(cpprob_hip_ctx* ctx, int32_t k, const double* h_means, const double* h_transition);
int cpprob_hip_infer_begin(cpprob_hip_ctx* ctx, const cpprob_hip_config* cfg, const double* h_observes, size_t n_observes);
int cpprob_hip_infer_run(cpprob_hip_ctx* ctx, uint64_t run_index);
int cpprob_hip_infer_summary(cpprob_hip_ctx* ctx, cpprob_hip_summary* out);
int cpprob_hip_infer_stats(cpprob_hip_ctx* ctx, double* h_stats, size_t n_doubles);
/* The same results left ON THE DEVICE, stream-ordered, no host synchronisation: d_out (device, caller-owned, at least
 * 4 + n_predict * stats_per_predict doubles) receives {log_evidence, ess_final, log_norm, max_logw, stats...} of the
 * run that was just enqueued.  For callers that feed them to a collective (multi-GPU island combine) or batch many
 * runs before looking at any. */
int cpprob_hip_infer_results_device(cpprob_hip_ctx* ctx, double* d_out, size_t n_doubles);
/* Per-step diagnostics of the last run: ESS after weighting step t and whether resampling
 * followed it.  Arrays of n_predict entries (either may be NULL). */
int cpprob_hip_infer_step_trace(cpprob_hip_ctx* ctx, double* h_ess, int32_t* h_resampled);

/* The particle store of the last run (the in-HBM form of the reference's posterior files,
 * src/cpprob/state.cpp:193-202,262-267).  Copies to host, synchronising.
 *   values    : [n_predict][n_particles], fp64 for real models, int32 for int models:
 *               value of predict hit t in slot i of generation t
 *   ancestors : [n_predict][n_particles] int32: slot of generation t-1 that slot i of
 *               generation t extends (row 0 and non-resampled steps: identity); SMC only
 *   logw      : [n_particles] final log-weights
 *   paths     : [n_predict][n_particles]: values along the ancestral line of final particle i,
 *               i.e. the full trace the reference would have dumped for particle i */
int cpprob_hip_copy_values(cpprob_hip_ctx* ctx, void* h_values, size_t n_bytes);
int cpprob_hip_copy_ancestors(cpprob_hip_ctx* ctx, int32_t* h_anc, size_t n_bytes);
int cpprob_hip_copy_logw(cpprob_hip_ctx* ctx, double* h_logw, size_t n_bytes);
int cpprob_hip_copy_paths(cpprob_hip_ctx* ctx, void* h_paths, size_t n_bytes);

/* ---- one joint population sharded over several contexts (one per GPU; the caller runs the collective) --
 * cfg.resample_scope = GLOBAL with n_global > n_particles.  Per step t = 0..T-1 (SIS: t = T-1 only):
 *   step_begin(t) propagates and weighs the local shard and writes this shard's
 *                 {max logw, sum exp(logw-max), sum exp(2(logw-max))} to d_local_totals (3 doubles, device);
 *   the caller all-gathers those over ranks (RCCL) into d_all_totals[3*world];
 *   step_end(t)   combines them ON DEVICE into the joint normaliser, ESS, evidence and resampling decision.
 * Resampling itself is local to the shard -- particles never migrate; right after a resampling step the
 * shard's particles carry log(shard mean weight / population mean weight), i.e. the shard's share of the
 * mass (distributed resampling with non-proportional allocation).  With world = 1 the protocol is
 * bit-identical to cpprob_hip_infer_run.  finish() runs the posterior read-out; cpprob_hip_infer_stats then
 * returns UN-NORMALISED weighted sums relative to exp(max_logw) -- real: {sum w x, sum w x^2}, int:
 * {sum w [x = s]} -- which the caller all-reduces and divides by W = exp(log_norm - max_logw)
 * (cpprob_amd/distributed.py; DESIGN.md section "Multi-GPU").  Everything is stream-ordered: no host sync. */
int cpprob_hip_smc_step_begin(cpprob_hip_ctx* ctx, int32_t t, uint64_t run_index, double* d_local_totals);
int cpprob_hip_smc_step_end(cpprob_hip_ctx* ctx, int32_t t, const double* d_all_totals, int32_t world, int32_t rank);
int cpprob_hip_smc_finish(cpprob_hip_ctx* ctx);
/* Fixed-point form (integer masses against a reference known before the generation exists): a generation whose heaviest particle
 * sat more than 6 nats below its reference lost too many of its 32 bits.  A single context repairs it inside cpprob_hip_infer_run
 * (cpprob_hip_summary::n_requantised); the shards of a joint population do the same through the step protocol, in integers, so
 * that the sharded run stays equal to the one-GPU run bit for bit (cpprob_hip_group_* drives this itself):
 *   first_bad_generation  after cpprob_hip_smc_finish: the first offending generation g, -1 when every generation kept its bits
 *                         (every rank reads the same g: the books come from the all-gathered totals); synchronises the stream;
 *   repair_begin(g)       generation g's log-weights recomputed from the particle store; d_local3 = {key of this shard's exact
 *                         maximum, 0, 0} (three 64-bit words, device): the caller all-gathers them as it does a step's totals;
 *   repair_end(g)         d_all3 = the all-gathered words; masses of generation g against the POPULATION's exact maximum, the
 *                         books rewound to where they stood before g; d_local3 = this shard's totals of the requantised
 *                         generation.  The caller all-gathers them and goes on as if step g had just run: cpprob_hip_smc_step_end(g),
 *                         the exchange behind step g, step_begin(g + 1) ... cpprob_hip_smc_finish; repeated while a later
 *                         generation trips (each round starts later).
 * There is no reference interface here: the reference has no SMC. */
int cpprob_hip_smc_first_bad_generation(cpprob_hip_ctx* ctx, int32_t* h_generation, double* h_gap);
int cpprob_hip_smc_repair_begin(cpprob_hip_ctx* ctx, int32_t g, double* d_local3);
int cpprob_hip_smc_repair_end(cpprob_hip_ctx* ctx, int32_t g, const double* d_all3, int32_t world, int32_t rank, double* d_local3);
/* Filtering-only shards (keep_history = 0) after finish: *joint_already = 1 -- cpprob_hip_infer_stats holds the JOINT population's
 * numbers (prefix-count form: they come from the all-gathered totals), nothing to combine; 0 -- it holds this shard's raw sums of
 * weight x f(x_t) per predict hit t and *d_masses (device, n_predict doubles) this shard's mass of generation t: the caller adds both
 * over ranks and divides (real predicts: mean = sum / mass, variance = second sum / mass - mean^2). */
int cpprob_hip_filter_masses(cpprob_hip_ctx* ctx, double** d_masses, int32_t* joint_already);

/* Exchange scope (cfg.resample_scope = CPPROB_HIP_SCOPE_EXCHANGE, systematic resampling): the sharded run draws the
 * SAME ancestors a single GPU holding all n_global particles would (one shared offset u makes every
 * rank's offspring range [o_r, o_{r+1}) a function of the all-gathered rank totals).  Output j lives on the rank whose
 * shard contains j; outputs whose ancestor sits on another rank arrive as lineage records -- the ancestor's trace
 * x_0 .. x_t, (t + 1) values of the model's value type -- and become extra columns of the receiving shard's
 * particle store, so every later kernel treats them as ordinary particles.  Between step_end(t) and
 * step_begin(t + 1), for every t < T - 1:
 *   plan(t)    host-synchronising; h_shard_begin[world + 1] = first global particle id of each rank's shard.
 *              Returns the resampling decision and, per peer, how many records this rank sends / receives
 *              (both zero when the step does not resample or when the shards' masses happen to match);
 *   pack(t)    writes sum(h_send_counts) records into d_send (device, caller-owned), grouped by destination rank
 *              in rank order;
 *   the caller moves them (RCCL all-to-all-v with the counts as split sizes, times (t + 1) elements);
 *   commit(t)  takes sum(h_recv_counts) records from d_recv, grouped by source rank in rank order.
 * pack/commit may be skipped data-wise (NULL pointers) when the respective count is zero, but must be called. */
int cpprob_hip_exchange_plan(cpprob_hip_ctx* ctx, int32_t t, int32_t world, int32_t rank, const uint64_t* h_shard_begin,
                             uint64_t* h_send_counts, uint64_t* h_recv_counts, int32_t* h_do_resample);
int cpprob_hip_exchange_pack(cpprob_hip_ctx* ctx, int32_t t, void* d_send);
int cpprob_hip_exchange_commit(cpprob_hip_ctx* ctx, int32_t t, const void* d_recv);

/* The same exchange WITHOUT any host synchronisation inside a run (what the multi-GPU drivers use: RCCL send / receive counts are
 * host constants, so the transport moves fixed-capacity segments and the plan lives on the device).
 *   setup      once after cpprob_hip_infer_begin: the shards' layout, the peer set -- all_peers = 0: the two neighbouring ranks (the
 *              offspring interval of a rank's sources leaves its shard by O(sqrt(n)) outputs: neighbours are all a well-mixed run
 *              needs), all_peers = 1: every rank (2, world = 1 only: the rank itself, a diagnostic that lets one GPU run the
 *              transport) -- and records_per_peer, the capacity of one peer segment;
 *   transport  the context-owned buffers: peer slot s (peer rank h_peers[s]) owns records_per_peer * (t + 1) values of
 *              bytes_per_value bytes at byte offset s * records_per_peer * (t + 1) * bytes_per_value of d_send / d_recv
 *              during the exchange that follows step t;
 *   between step_end(t) and step_begin(t + 1), t < T - 1:
 *     pack_async(t)    plans on the device and fills d_send;
 *     the caller moves slot s of d_send to rank h_peers[s], into the slot that rank keeps for this one, stream-ordered;
 *     commit_async(t)  turns what arrived in d_recv into annex columns;
 *   status     after the run (synchronises): overflow = 0 fine; otherwise a set of bits: 1 a peer segment was too small, 2 a rank
 *              outside the peer set was needed, 4 the immigrant annex was too small.  A non-zero value invalidates the run on
 *              EVERY rank of the group (ranks must agree on it -- all-reduce the bits): repeat it with what overflowed enlarged
 *              (records_per_peer / all_peers = 1 / cpprob_hip_config::annex_kcols).  Results do not depend on the transport parameters. */
int cpprob_hip_exchange_setup(cpprob_hip_ctx* ctx, int32_t world, int32_t rank, const uint64_t* h_shard_begin, int32_t all_peers, uint64_t records_per_peer);
int cpprob_hip_exchange_transport(cpprob_hip_ctx* ctx, void** d_send, void** d_recv, int32_t* n_peers, int32_t* h_peers, uint64_t* records_per_peer,
                                  uint64_t* bytes_per_value);
/*   direct     (optional, after setup) h_peer_recv[world]: for every peer rank r, rank r's d_recv (cpprob_hip_exchange_transport) as
 *              THIS context's device addresses it -- the same pointer when both contexts share a device, a peer-access pointer
 *              inside one process, a hipIpcOpenMemHandle mapping across processes; NULL elsewhere.  pack_async then stores every
 *              record straight into the receiver's slot for this rank and d_send is not used: the caller moves nothing, it only
 *              orders the receiver's commit_async(t) behind the senders' pack_async(t) (any collective every rank enters after its
 *              pack_async will do).  NULL switches back.  setup() resets it (the buffers may move).
 *   traffic    after the run (synchronises): lineage records this rank sent after each step (h_sent_per_step[n_predict], may be NULL),
 *              their total and their bytes -- records x (t + 1) x bytes_per_value: what the direct transport puts on the links. */
/*   remote     (optional, after direct) REMOTE LINEAGES: with every rank's particle store addressable from this device, a migrating
 *              particle takes only its current state and the slot it leaves along -- value + 8 bytes per record (+ 4 where the particles carry
 *              trace words: short discrete traces) -- and pack_async
 *              stores both straight into the RECEIVING rank's annex column and origin table: every rank keeps every rank's annex
 *              fill (a function of the all-gathered totals), so there is no receive buffer, no segment capacity, no peer set and
 *              commit_async launches nothing (what remains of the overflow bits is 4: an annex too small).  The particle's history
 *              stays where it is, and whoever walks the lineage later -- the read-out, cpprob_hip_copy_paths -- continues in that
 *              rank's store.  store() fills this context's own entry (and sizes its origin table); remote() takes h_stores[world],
 *              entry r = rank r's store as THIS device addresses it (peer access / hipIpcOpenMemHandle of the three arrays: they
 *              are written AND read through the mapping).  Every rank of the group must be in the same mode, and the caller still
 *              orders step_begin(t + 1) of every rank behind pack_async(t) of every rank.  No lineage is extracted, shipped or
 *              committed any more: the exchange of a step costs what its migrants' states cost. */
typedef struct cpprob_hip_store {
    const void* d_values; const void* d_ancestors; const void* d_origin;     /* [T][row_stride] values, [T][row_stride] int32, [annex] int64 */
    uint64_t row_stride, n_local_columns;
    const void* d_trace[2];    /* short discrete traces (hmm<T <= 16>): the particles' trace words, [row_stride] uint32 each, by the step's
                                  parity -- a migrant's word is stored into the receiving rank's with its state; NULL where not in use */
} cpprob_hip_store;
int cpprob_hip_exchange_store(cpprob_hip_ctx* ctx, cpprob_hip_store* out);
int cpprob_hip_exchange_remote(cpprob_hip_ctx* ctx, const cpprob_hip_store* h_stores);
int cpprob_hip_exchange_direct(cpprob_hip_ctx* ctx, void* const* h_peer_recv);
int cpprob_hip_exchange_traffic(cpprob_hip_ctx* ctx, int64_t* h_sent_per_step, size_t n_steps, uint64_t* h_records, uint64_t* h_bytes);
int cpprob_hip_exchange_pack_async(cpprob_hip_ctx* ctx, int32_t t);
int cpprob_hip_exchange_commit_async(cpprob_hip_ctx* ctx, int32_t t);
int cpprob_hip_exchange_status(cpprob_hip_ctx* ctx, int32_t* h_overflow, uint64_t* h_annex_used);

/* ---- one joint population over several GPUs, driven from the host side of this library ------------------------------------
 * A group = one context per GPU + collectives + a transport; cpprob_hip_group_run enqueues a WHOLE exchange-scope run on every local
 * rank -- per step: propagate / weigh, all-gather of 3 doubles per rank, device-side plan, the migrating lineages, commit -- with no
 * host synchronisation inside (cpprob_amd/csrc/group.hpp).  Replaces what a caller of the reference would have to build around
 * cpprob::inference to use more than one device (the reference is single-process, src/cpprob/state.cpp:20-21).
 *   create   world == n_local: every rank in this process.  Distinct devices: RCCL over xGMI, one communicator and one host
 *            thread per GPU.  All devices equal: "loopback" -- every rank's context on that one device and one stream, program
 *            order instead of collectives (how a one-GPU machine exercises the protocol; RCCL refuses duplicate devices).
 *            world > n_local: one rank (n_local = 1) of a group spread over processes; unique_id = the 128 bytes rank 0 got from
 *            cpprob_hip_group_unique_id and handed to every rank (the launcher's job: torchrun, MPI, a file).
 *   create_external   one rank of a group whose collectives are the CALLER's (MPI, gloo, ...): allgather(user, h_in, h_out,
 *            bytes_per_rank) gathers host bytes of every rank in rank order and returns 0; it is called from cpprob_hip_group_begin /
 *            _run / _results, which synchronise the stream around it.  The lineages move by direct stores (below); nothing else is
 *            asked of the caller.
 *   transport (before begin) records_per_peer = capacity of one peer segment (0: the default, 8 sqrt(N) + 4096); all_peers = 1: every
 *            rank is a peer, 0: the two neighbouring ranks, < 0: the default (neighbours); flags = CPPROB_HIP_GROUP_*.  Results never
 *            depend on any of them; a run they prove too small for is repeated with larger ones (results).
 *            How the lineages move: DIRECT -- the sending rank's packing kernel stores every record into the receiving rank's buffer
 *            (peer access inside a process, hipIpc mappings between processes), ordered by a one-double all-gather per step, so that
 *            what crosses xGMI is records x (t + 1) x value size and nothing on a step that does not resample -- wherever every rank
 *            can map its peers' buffers; otherwise SENDRECV -- ncclSend / ncclRecv of the fixed-capacity segments.
 *            The per-step collectives (24 bytes of totals per rank; the ordering of the direct stores): MAILBOXES wherever the ranks
 *            can map each other's memory -- a rank stores its words and the step's sequence number into every peer's mailbox and spins
 *            on its own, one short launch, no library call and no host inside a run (csrc/device_collectives.hpp); proven by a round
 *            trip at the first begin, and a wait that times out (5 s) makes results() repeat the run on the library's collectives.
 *            Otherwise RCCL calls / the caller's all-gather.  The collectives flags are read at the first begin (LIBRARY_COLLECTIVES at
 *            any later begin still switches the mailboxes off).
 *   begin    cfg as for cpprob_hip_infer_begin with n_particles = the WHOLE population (particle_offset / n_global / scope are
 *            set per rank by the group); shards are contiguous and equal unless h_shard_sizes[world] names them.  Systematic SMC
 *            runs in the exchange scope (exact global resampling); other resamplers and SIS in the global scope.  COLLECTIVE.
 *   run      asynchronous (RCCL / loopback).
 *   results  COLLECTIVE: every rank of the group calls it.  Synchronises, all-reduces and normalises: h_stats as
 *            cpprob_hip_infer_stats of ONE GPU holding the whole population would return.  If a transport segment or the annex
 *            overflowed -- every rank sees the same all-reduced flags -- it first repeats the LAST run with a larger transport
 *            (h_reruns counts those since begin): the numbers never depend on the transport.  Runs enqueued before the last one are
 *            not repeated: a caller that pipelines runs should settle the transport with one run + results first.
 *   traffic  of the run results last collected.
 *   context  the rank's context, e.g. for cpprob_hip_copy_paths of its shard. */
typedef struct cpprob_hip_group cpprob_hip_group;
typedef struct cpprob_hip_collectives {
    void* user;
    int (*allgather)(void* user, const void* h_in, void* h_out, size_t bytes_per_rank);
} cpprob_hip_collectives;
#define CPPROB_HIP_GROUP_SENDRECV 1u            /* never map peers' buffers: ncclSend / ncclRecv of fixed-capacity segments */
#define CPPROB_HIP_GROUP_WORLD1_COLLECTIVES 2u  /* diagnostic, world = 1: issue every collective of the multi-GPU path anyway and exchange
                                                   (zero records) with the rank itself, so that one GPU runs all of the transport's calls */
#define CPPROB_HIP_GROUP_SHIP_LINEAGES 4u      /* direct transport: migrants still take their whole lineage along (records of t + 1 values)
                                                   instead of leaving it on the rank they come from (remote lineages, the default where
                                                   every rank can address every rank's particle store) */
#define CPPROB_HIP_GROUP_LIBRARY_COLLECTIVES 8u /* the per-step collectives through RCCL / the caller's all-gather even where the ranks can map
                                                   each other's memory (default there: stores into the peers' mailboxes, no call inside a run) */
#define CPPROB_HIP_GROUP_MAILBOX_COLLECTIVES 16u /* loopback groups too (they gather for every rank in one launch by default) */
#define CPPROB_HIP_TRANSPORT_NONE 0
#define CPPROB_HIP_TRANSPORT_DIRECT 1
#define CPPROB_HIP_TRANSPORT_SENDRECV 2
typedef struct cpprob_hip_traffic {
    uint64_t records;          /* lineage records that changed rank, all ranks, all steps of the run                           */
    uint64_t payload_bytes;    /* sum over steps of records x (t + 1) x value size                                             */
    uint64_t wire_bytes;       /* what the transport put on the links for them: = payload_bytes (direct), capacity (send/recv)  */
    uint64_t collective_bytes; /* the small collectives: all-gathers of 3 doubles (+ 1 ordering the direct stores) per step, the final all-reduce */
    int32_t transport;         /* CPPROB_HIP_TRANSPORT_*                                                                       */
    int32_t remote_lineages;   /* 1: migrants left their history on the rank they came from (records of value + 8 bytes)           */
    int32_t mailbox_collectives; /* 1: the per-step collectives were stores into the peers' mailboxes, not library calls            */
    int32_t reserved;
} cpprob_hip_traffic;
int cpprob_hip_group_unique_id(void* out128, size_t n_bytes);
int cpprob_hip_group_create(const int32_t* devices, int32_t n_local, int32_t world, int32_t first_rank, const void* unique_id, cpprob_hip_group** out);
int cpprob_hip_group_create_external(int32_t device, int32_t world, int32_t rank, const cpprob_hip_collectives* collectives, cpprob_hip_group** out);
void cpprob_hip_group_destroy(cpprob_hip_group* group);
const char* cpprob_hip_group_last_error(const cpprob_hip_group* group);
int cpprob_hip_group_transport(cpprob_hip_group* group, uint64_t records_per_peer, int32_t all_peers, uint32_t flags);
int cpprob_hip_group_begin(cpprob_hip_group* group, const cpprob_hip_config* cfg, const double* h_observes, size_t n_observes, const uint64_t* h_shard_sizes);
int cpprob_hip_group_run(cpprob_hip_group* group, uint64_t run_index);
int cpprob_hip_group_sync(cpprob_hip_group* group);
int cpprob_hip_group_size(const cpprob_hip_group* group, int32_t* world, int32_t* n_local, int32_t* first_rank);
cpprob_hip_ctx* cpprob_hip_group_context(cpprob_hip_group* group, int32_t local_index);
int cpprob_hip_group_results(cpprob_hip_group* group, cpprob_hip_summary* out, double* h_stats, size_t n_doubles, int32_t* h_reruns);
int cpprob_hip_group_traffic(cpprob_hip_group* group, cpprob_hip_traffic* out);
/* Where a rank-step's time goes (what the first run over real links has to explain).  profile(on): HIP events between the launches of
 * every step of the following runs, on every local rank's stream.  profile_read: of the last run, microseconds per rank-step (mean over
 * the steps that run every phase, the slowest local rank; a loopback group: the SUM over its ranks, which share one stream):
 * [0] step kernel + shard totals (+ the mailbox all-gather where the totals launch carries it), [1] a separate all-gather (mailbox
 * launch or library call), [2] the hand-over of the gathered totals, [3] the packing launch, [4] the barrier that orders the commits
 * behind every rank's stores, [5] the commit, [6] of all that, the time mailbox waits spun (device wall clock), [7] steps timed. */
int cpprob_hip_group_profile(cpprob_hip_group* group, int32_t on);
int cpprob_hip_group_profile_read(cpprob_hip_group* group, double* h_out8);
/* Which collectives and which migrant transport the group settled on, and why (fall-back rungs included), as one line of text; valid
 * until the calling thread's next call. */
const char* cpprob_hip_group_note(const cpprob_hip_group* group);

/* ---- building blocks (also the unit-parity surface) --------------------------------------
 * All pointers are device pointers; n is the element count. */

/* Raw Philox4x32-10 blocks: d_out[4*i..4*i+3] = block(seed, group0 + i, draw)
 * (= rocrand_init(seed, subsequence = group, offset = 4*draw); rocrand4()). */
int cpprob_hip_philox_blocks(cpprob_hip_ctx* ctx, uint64_t seed, uint64_t group0, uint64_t draw, size_t n, uint32_t* d_out);
/* Variate generators standing in for boost::random::*::operator()(get_rng()) (cpprob.hpp:34):
 * element i is the draw of global particle pid0+i at statement ordinal `draw`.  Neighbouring
 * particles share Philox blocks: 32-bit variates (uniform_smallint, discrete) take word pid&3 of
 * block(group pid>>2); normal variates take component pid&1 of rocRAND's box_muller_double of
 * block(group pid>>1); 53-bit uniforms (uniform_real) take word pair pid&1 of block(pid>>1). */
int cpprob_hip_draw_normal(cpprob_hip_ctx* ctx, uint64_t seed, uint64_t pid0, uint64_t draw, double mean, double sigma, size_t n, double* d_out);
int cpprob_hip_draw_uniform_smallint(cpprob_hip_ctx* ctx, uint64_t seed, uint64_t pid0, uint64_t draw, int64_t a, int64_t b, size_t n, int32_t* d_out);
int cpprob_hip_draw_discrete(cpprob_hip_ctx* ctx, uint64_t seed, uint64_t pid0, uint64_t draw, const double* h_weights, int32_t k, size_t n, int32_t* d_out);
int cpprob_hip_draw_uniform_real(cpprob_hip_ctx* ctx, uint64_t seed, uint64_t pid0, uint64_t draw, double a, double b, size_t n, double* d_out);
/* poisson: inversion by sequential search on the particle's 53-bit uniform (exact law, cost O(mean)) */
int cpprob_hip_draw_poisson(cpprob_hip_ctx* ctx, uint64_t seed, uint64_t pid0, uint64_t draw, double mean, size_t n, int32_t* d_out);

/* logpdf functors (include/cpprob/distributions/utils_*.hpp), elementwise. */
int cpprob_hip_logpdf_normal(cpprob_hip_ctx* ctx, const double* d_x, const double* d_mean, const double* d_sigma, size_t n, double* d_out);
int cpprob_hip_logpdf_uniform_real(cpprob_hip_ctx* ctx, const double* d_x, const double* d_a, const double* d_b, size_t n, double* d_out);
int cpprob_hip_logpdf_poisson(cpprob_hip_ctx* ctx, const int32_t* d_x, const double* d_mean, size_t n, double* d_out);
int cpprob_hip_logpdf_uniform_smallint(cpprob_hip_ctx* ctx, const int32_t* d_x, int64_t a, int64_t b, size_t n, double* d_out);
int cpprob_hip_logpdf_discrete(cpprob_hip_ctx* ctx, const int32_t* d_x, const double* h_weights, int32_t k, size_t n, double* d_out);

/* The range-specific fp64 elementary functions the variate generators and the weight kernels use in place of the device library's
 * (cpprob_amd/include/cpprob/detail/fastmath.hpp; they stand where the reference calls std::log / std::exp through Boost.Random and
 * include/cpprob/distributions/utils_normal_distribution.hpp:38-41), elementwise: which = 0 log01 on [2^-53, 1], 1 sincospi02 on
 * (0, 2] (d_out0 = sin(pi x), d_out1 = cos(pi x)), 2 exp_nonpos on [-745, 0]; 3: the fixed-point weight min(rint(exp(x) 2^32), 2^32 - 1)
 * of a log-weight x <= 0 against the reference 0 (fixed_mass.hpp: fix_weight), as a double.  Unit-parity surface: tests sweep the domain edges. */
int cpprob_hip_fastmath(cpprob_hip_ctx* ctx, int32_t which, const double* d_x, size_t n, double* d_out0, double* d_out1);

/* EmpiricalDistribution (include/cpprob/postprocess/empirical_distribution.hpp):
 * h_out[0] = max, [1] = logsumexp (:125-143), [2] = ESS = (sum W^2)^-1. */
int cpprob_hip_logsumexp_ess(cpprob_hip_ctx* ctx, const double* d_logw, size_t n, double* h_out3);
/* h_out[0] = mean (:68-71), [1] = variance (:78-81), [2] = logsumexp, [3] = ESS. */
int cpprob_hip_weighted_moments(cpprob_hip_ctx* ctx, const double* d_x, const double* d_logw, size_t n, double* h_out4);
/* h_out[s] = sum_i W_i [x_i == s], s < k <= 8  (distribution(), :30-40). */
int cpprob_hip_weighted_hist(cpprob_hip_ctx* ctx, const int32_t* d_x, const double* d_logw, size_t n, int32_t k, double* h_out);
/* ... and of n_cols columns (column j at d_x + j * col_stride, col_stride >= n) against ONE log-weight array: one normalisation, one
 * read-out launch, one synchronisation for all of them.  h_out4: [n_cols][4] as cpprob_hip_weighted_moments; h_out: [n_cols][k].
 * Columns whose stride is a whole number of 1024-particle tiles (col_stride % 1024 == 0, col_stride >= n rounded up to a tile) are read
 * WHERE THEY LIE: their slots [n, col_stride) must be readable and hold finite numbers (they weigh nothing); any other stride is
 * copied into tile-padded scratch first.  A read-back hung on the context (cpprob_hip_readback_with_next_result) rides these calls too. */
int cpprob_hip_weighted_moments_columns(cpprob_hip_ctx* ctx, const double* d_x, size_t n_cols, size_t col_stride, const double* d_logw, size_t n, double* h_out4);
int cpprob_hip_weighted_hist_columns(cpprob_hip_ctx* ctx, const int32_t* d_x, size_t n_cols, size_t col_stride, const double* d_logw, size_t n, int32_t k, double* h_out,
                                     double* h_lse_ess /* may be NULL: {logsumexp of the weights, ESS} */);

/* Resampling: d_anc[jj] = ancestor (index into d_logw[0..n_in)) of output j0 + jj, for n_out
 * consecutive outputs of a population of n_total_out positions.  Uniforms come from
 * draw index (1<<40) + step: systematic one 53-bit uniform of group 0; stratified the 32-bit
 * uniform of id j; multinomial the 53-bit uniform of id j. */
int cpprob_hip_resample(cpprob_hip_ctx* ctx, int32_t kind, const double* d_logw, size_t n_in, uint64_t seed, uint64_t step,
                        uint64_t j0, size_t n_out, uint64_t n_total_out, int32_t* d_anc);
/* One step of SMC bookkeeping ON THE DEVICE, stream-ordered, no host synchronisation (what the generic model path runs
 * between two launches of the model body): normalises d_logw[0..n), stores the ESS in d_ess[step], decides
 * (ESS < ess_frac * n, never at the last step) into d_resampled[step], keeps the running log evidence in *d_log_z
 * (reset at step 0; + log mean weight whenever the step resamples or is the last), and fills d_anc[0..n) with the next
 * generation's ancestors -- the identity when the step does not resample.  Uniforms: draw index (1<<40) + step + 1,
 * as cpprob_hip_resample(seed, step + 1). */
int cpprob_hip_smc_bookkeep(cpprob_hip_ctx* ctx, int32_t kind, const double* d_logw, size_t n, uint64_t seed, int32_t step, int32_t last,
                            double ess_frac, double* d_ess, int32_t* d_resampled, double* d_log_z, int32_t* d_anc);
/* The same bookkeeping for SYSTEMATIC resampling on fixed-point weights (cpprob_amd/csrc/step_fixed.hpp, bookkeep_fixed.hpp): the
 * weights become integers q_i = min(rint(exp(logw_i - max logw) 2^32), 2^32 - 1), sums and the resampling comb run on exact 64-bit
 * masses -- three short launches, no floating-point CDF; uniforms as above.  d_anc may be NULL when last != 0.  n <= 2^28. */
int cpprob_hip_smc_bookkeep_fixed(cpprob_hip_ctx* ctx, const double* d_logw, size_t n, uint64_t seed, int32_t step, int32_t last, double ess_frac,
                                  double* d_ess, int32_t* d_resampled, double* d_log_z, int32_t* d_anc);
/* ... and for any resampler: kind = CPPROB_HIP_RESAMPLE_* -- stratified (output j at j + u_j against the same masses) and multinomial
 * (strata form: the counts of this resampling's thresholds per stratum in one or two short launches in front of the ancestors' launch),
 * the arithmetic of the built-in models' step kernels.  cpprob_hip_smc_bookkeep_fixed = kind SYSTEMATIC. */
int cpprob_hip_smc_bookkeep_fixed_rs(cpprob_hip_ctx* ctx, int32_t kind, const double* d_logw, size_t n, uint64_t seed, int32_t step, int32_t last, double ess_frac,
                                     double* d_ess, int32_t* d_resampled, double* d_log_z, int32_t* d_anc);
/* ---- SMC step of an UNCHANGED model with the resampling inside the model's own launch (cpprob_amd/include/cpprob/gpu.hpp:
 * model_step_kernel; replaces the loop body of reference include/cpprob/cpprob.hpp:194-201 for StateType::smc) ----------------------
 * The model translation unit owns the kernel (it is a template over the model function); the library owns what the kernel's
 * prologue and epilogue work on: three rotating copies of the 64-ary mass hierarchy (cpprob_amd/include/cpprob/detail/fixed_mass.hpp),
 * the integer weights of two generations and a small control block.  cpprob_hip_generic_begin sizes them for a population of n
 * particles (one hierarchy entry per 256-particle block = per workgroup of the step kernel; first call and growth allocate), clears
 * them on the context's stream and describes them in *out; step t of the run reads copy (t + 2) % 3 of the hierarchy, publishes
 * into copy t % 3 and clears the upper levels of copy (t + 1) % 3; integer weights: generation t in q[t & 1].  n <= 64^3 * 256. */
typedef struct cpprob_hip_generic_layout {
    uint64_t* hier;            /* [3][per_copy] 64-bit words */
    uint64_t per_copy;
    uint64_t lvl_off[3];       /* word offset of each level inside a copy */
    int32_t n_ent[3];          /* entries per level (n_ent[0] = blocks) */
    int32_t n_lev;
    uint64_t q0_off, m0_off;   /* the blocks' squares / maxima inside a copy */
    void* table;               /* the same layout in device memory (fixed_mass.hpp: HierTable) */
    uint32_t* q[2];            /* integer weights, by the step's parity: padded to whole 1024-particle tiles, and one tile more */
    void* ctrl;                /* device_trace.hpp: StepCtrl2 */
    int32_t blocks;
    int32_t block;             /* particles per hierarchy entry: 256 (cpprob_hip_generic_begin) or 1024 (cpprob_hip_generic_begin_tiles) */
} cpprob_hip_generic_layout;
int cpprob_hip_generic_begin(cpprob_hip_ctx* ctx, size_t n, cpprob_hip_generic_layout* out);
/* The same with one hierarchy entry per 1024-particle TILE: the layout of model_step_kernel_quad, whose workgroups run the model body
 * for four particles a lane behind ONE ancestor search (cpprob/gpu.hpp).  n <= 64^3 * 1024 (and 2^28).  The exact-reference passes
 * below work on 256-particle blocks only. */
int cpprob_hip_generic_begin_tiles(cpprob_hip_ctx* ctx, size_t n, cpprob_hip_generic_layout* out);
/* Exact-reference form (no host-known bound of a step's log-likelihood): the step's launch stored the log-weights only; two short
 * launches find the generation's exact maximum, quantise d_logw[0..n) against it into q[t & 1] and publish maxima and masses into
 * copy t % 3. */
int cpprob_hip_generic_quantize(cpprob_hip_ctx* ctx, int32_t t, const double* d_logw, size_t n);
/* The same in two halves, for one shard of a JOINT population whose host threads combine the ranks' numbers between the launches:
 * the maximum pass alone (into copy t % 3), and the masses against a reference the caller supplies (the population's exact maximum). */
int cpprob_hip_generic_max(cpprob_hip_ctx* ctx, int32_t t, const double* d_logw, size_t n);
int cpprob_hip_generic_quantize_ref(cpprob_hip_ctx* ctx, int32_t t, const double* d_logw, size_t n, double ref);
/* {mass, squares, order key of the maximum} of generation t as copy t % 3 holds them: 24 bytes into d_out3 (device memory; stream-ordered):
 * what a shard of a joint population contributes to the generation's totals. */
int cpprob_hip_generic_totals(cpprob_hip_ctx* ctx, int32_t t, size_t n, uint64_t* d_out3);
/* Bookkeeping of the run's LAST generation (T - 1: the copy (T - 1) % 3): d_ess[T-1], d_resampled[T-1] = 0, the final term of
 * *d_log_z; *d_flags gets 4 / 5 where the generation's heaviest particle sat above / more than gap_limit below its reference. */
int cpprob_hip_generic_finish(cpprob_hip_ctx* ctx, int32_t T, size_t n, double gap_limit, double* d_ess, int32_t* d_resampled, double* d_log_z, int32_t* d_flags);
/* The systematic offset in [0, 1) of the resampling that precedes step `step`: Philox draw (1 << 40) + step of group 0 --
 * a pure function of (seed, step), evaluated on the host so that launches carry it as an argument. */
double cpprob_hip_systematic_offset(uint64_t seed, uint64_t step);
/* Traces from per-step records: d_anc [T][n] (row t: slot of generation t-1 that slot i of generation t extends; a row counts only where
 * d_resampled[t-1] != 0), d_cols [H][n] with row h recorded in the slots of generation h_gen[h] (non-decreasing in h).  d_out[h][i] =
 * d_cols[h][slot of generation h_gen[h] on the ancestral line of FINAL particle i]: the value particle i's trace holds for that row.
 * is_int = 0: fp64 rows, 1: int32 rows.  Stream-ordered; what the unchanged-model SMC path reads its predicts out with. */
int cpprob_hip_lineage_gather(cpprob_hip_ctx* ctx, const int32_t* d_anc, const int32_t* d_resampled, int32_t T, size_t n, const void* d_cols,
                              int32_t is_int, const int32_t* h_gen, int32_t H, void* d_out);
/* The same walk with the rows' statistics taken on the way -- what cpprob_hip_lineage_gather followed by cpprob_hip_weighted_moments_columns /
 * _hist_columns against d_logw (the FINAL generation's log-weights, n of them) returns, bit for bit, without the traces' round trip through
 * memory (StatsPrinter's numbers, reference stats_printer.hpp:68-120, when nobody asked for the traces themselves).  h_out4: [H][4] =
 * {mean, variance, logsumexp, ess}; h_out: [H][k] probabilities, 1 <= k <= 8; h_lse_ess (may be NULL): {logsumexp, ess}.  Synchronises. */
/* Hangs ONE read-back of the caller's on the next cpprob_hip_lineage_moments / cpprob_hip_lineage_hist call of this context: bytes
 * [d_src, d_src + bytes) are stored into pinned host memory by the launch that stores the call's own result, and are in h_dst when
 * the call returns -- the call's own stream synchronisation is the only one (cpprob/gpu.hpp: the run's ESS / evidence / flag tail
 * rides the read-out's result instead of a host round trip of its own).  bytes = 0 cancels.  <= 1 MiB, a multiple of 4. */
int cpprob_hip_readback_with_next_result(cpprob_hip_ctx* ctx, const void* d_src, void* h_dst, size_t bytes);
/* Optional: uploads the records' generation table of the three calls above / below ahead of time (it is uploaded at their first use
 * otherwise, with a synchronisation; an unchanged table is never uploaded twice). */
int cpprob_hip_lineage_prepare(cpprob_hip_ctx* ctx, const int32_t* h_gen, int32_t H, int32_t T);
int cpprob_hip_lineage_moments(cpprob_hip_ctx* ctx, const int32_t* d_anc, const int32_t* d_resampled, int32_t T, size_t n, const double* d_cols,
                               const int32_t* h_gen, int32_t H, const double* d_logw, double* h_out4);
int cpprob_hip_lineage_hist(cpprob_hip_ctx* ctx, const int32_t* d_anc, const int32_t* d_resampled, int32_t T, size_t n, const int32_t* d_cols,
                            const int32_t* h_gen, int32_t H, const double* d_logw, int32_t k, double* h_out, double* h_lse_ess);
/* d_dst[i] = d_src[d_idx[i]] */
int cpprob_hip_gather_f64(cpprob_hip_ctx* ctx, const double* d_src, const int32_t* d_idx, size_t n, double* d_dst);
int cpprob_hip_gather_i32(cpprob_hip_ctx* ctx, const int32_t* d_src, const int32_t* d_idx, size_t n, int32_t* d_dst);

/* Kernel timing of the last cpprob_hip_infer_run, measured with HIP events on the context's
 * stream when enabled (adds two event records per kernel class; off by default).
 * h_ms[k] = accumulated milliseconds, h_calls[k] = launches of kernel class k since the last
 * reset; classes: 0 propagate/step, 1 scan-partials, 2 smoothing read-out, 3 finalize,
 * 4 sis, 5 resample-only.  Synchronises. */
#define CPPROB_HIP_N_KERNEL_CLASSES 6
int cpprob_hip_profile_enable(cpprob_hip_ctx* ctx, int32_t on);
int cpprob_hip_profile_read(cpprob_hip_ctx* ctx, double* h_ms, int64_t* h_calls, int32_t reset);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* CPPROB_HIP_H_ */
