// C ABI of the gfx950 SIS / SMC engine: context, particle store, launch sequencing.
// Declarations and the reference interfaces they replace: include/cpprob_hip.h.
#include "../../include/cpprob_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "step_counts.hpp"
#include "step_fixed.hpp"
#include "exchange.hpp"
#include "bookkeep_fixed.hpp"
#include "trace_words.hpp"
#include "device_collectives.hpp"

using namespace cph;

namespace {

thread_local std::string g_last_error;      // per thread: contexts (and the group driver's workers) live on threads of their own

struct EventPair { hipEvent_t a, b; int cls; int count; };

}  // namespace

struct cpprob_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    cpprob_hip_config cfg{};
    bool begun = false, ran = false;
    int T = 0;              // predict hits per trace
    int n_obs = 0;
    bool is_int = false;
    int K = 0;              // stats per predict
    int64_t n = 0, ld = 0;
    int64_t rs = 0;         // row stride of values[] / anc[]: ld + annex_cap
    bool grid_refs = false; // continuous-weight model: tile references on the grid {k ln 2} (kernels.hpp, grid_reference)
    size_t ssz = 8;         // bytes per element of values[] (Model::store_t; may be narrower than the model's value type)
    int nb = 0;             // tiles
    int smooth_grid = 0;
    int walk_cap = 0, walk_grid = 0, n_cu = 0;    // lineage walk: partials allocated for walk_cap workgroups; the grid of this configuration's walk (0: not chosen yet)
    ModelParams mp{};
    uint64_t run_seed = 0;
    uint64_t pop_n = 0;     // size of the population this shard is resampled with (n_global, or n for islands)

    // device buffers
    double* d_obs = nullptr;
    double* d_logw[2] = {nullptr, nullptr};
    double* d_wrel[2] = {nullptr, nullptr};
    double* d_bf = nullptr;
    bool final_from_counts = false, final_bookkeep_pending = false; int final_copy = 0;
    bool scan2_deferred = false;         // exchange scope, floating-point form: step_end left the cross-rank combine to the plan launch
    bool keep = true, cap_keep = true;   // keep_history: per-step values + ancestors (false: two rows, no ancestors, filtering statistics)
    double* d_filter_w = nullptr;        // filtering-only shards: [T] this shard's mass of generation t (the joint normaliser is their sum over ranks)
    double* d_fpart = nullptr;           // filtering-only runs, floating-point form: [T][K + 2][smooth_grid] (filter_partials_kernel)   // prefix-count form, single shard: the read-out works from the final generation's counts
    double* d_ll_tab = nullptr;     // hmm: [T][3] emission log-densities
    void* d_values = nullptr;
    int32_t* d_anc = nullptr;
    void* d_paths = nullptr;
    Partial* d_part[2] = {nullptr, nullptr};   // tile partials, ping-pong between generations
    int cur_part = 0;                          // which one holds the latest generation
    double* d_e_tab = nullptr;                 // hmm: [T][4]
    double* d_gpart = nullptr;                 // slab partials of the two-level normalisation (large populations)
    double* d_stile = nullptr;                 // SIS fused read-out: per-tile weighted sums [kMaxReadoutCols][nb]
    double* d_gstat = nullptr;                 //                     per-slab weighted sums [kMaxReadoutCols][kMaxSlabs]
    double* d_bc = nullptr;
    StepCtrl* d_ctrl = nullptr;
    double* d_ess = nullptr;
    int32_t* d_resampled = nullptr;
    double* d_stats_part = nullptr;
    double* d_stats = nullptr;
    uint32_t* d_bbf_strata = nullptr; uint32_t* d_bbf_strata_top = nullptr; size_t bbf_strata_cap = 0;     // ... of the bookkeeping protocol (cpprob_hip_smc_bookkeep_fixed_rs)
    uint32_t* d_strata = nullptr;   // multinomial, strata form: [T - 1][2^k + 1] first outputs of the strata, every step of the run
    uint32_t* d_strata_top = nullptr; int strata_phase = 0; bool strata_pending = false;   // ... [2][T][64] totals of the level-6 nodes (the set in use alternates run by run)
    size_t strata_cap_words = 0; int strata_cap_T = 0;
    // ... of one shard of a joint population (exchange scope, strata_cut.hpp): what the ranks' boundaries cut, rewritten by every exchange
    uint32_t* d_cut_tab = nullptr; CutHead* d_cut_head = nullptr; uint32_t* d_cut_srccnt = nullptr;
    double* d_cdf = nullptr;        // multinomial only
    int32_t* d_anc_pre = nullptr;   // multinomial only
    double* d_local_totals = nullptr;
    double* totals_out = nullptr;   // caller-provided {max, sum, sum of squares} of the shard (step protocol)
    bool sharded = false;           // the last run went through the step protocol: stats stay un-normalised
    bool step_protocol = false;     // a step-protocol run is in progress (the step kernel must not normalise on its own)
    int step_t = -1;                // step of the last cpprob_hip_smc_step_begin
    int cur = 0;                    // logw buffer holding the latest generation
    // exchange scope: exact global resampling, offspring of remote sources migrate in as annex columns
    bool exchange = false;
    int64_t annex_cap = 0, annex_used = 0;
    std::vector<int64_t> annex_used_before;   // (synchronising exchange calls) annex columns in use before the commit that follows step t: what a repair rewinds to
    double* d_obound = nullptr;     // [world + 2]: offspring-interval bounds per rank, then the resampling decision
    double* h_obound = nullptr;     // pinned host copy (the one host read-back per step of the exchange scope)
    // read-backs of small results go through pinned memory (a copy into pageable memory blocks the host for its own round trip: three of
    // them closed an unchanged-model run at ~25 us each) and ride ONE stream synchronisation; the caller may hang a copy of its own on it
    char* h_pin = nullptr; size_t h_pin_cap = 0; uint64_t pin_seq = 0;
    const void* ride_src = nullptr; void* ride_dst = nullptr; size_t ride_bytes = 0;
    int32_t* d_send_src = nullptr; size_t send_src_cap = 0;
    // device-resident exchange plan (exchange.hpp) and the transport geometry
    ExchangePlan* d_xplan = nullptr;
    int64_t* d_shard_begin = nullptr; int32_t* d_slot_of_rank = nullptr;
    int x_world = 0, x_rank = 0;                  // as of the last step_end
    const double* x_all_totals = nullptr;
    bool x_fixed = false; int64_t x_cap = 0; int x_mode = 0; std::vector<int> x_peers;   // fixed-capacity transport (cpprob_hip_exchange_setup)
    void* d_xsend = nullptr; void* d_xrecv = nullptr; size_t x_buf_bytes = 0;
    // direct transport: the packing kernel stores records straight into the receivers' buffers (cpprob_hip_exchange_direct)
    bool x_direct = false; void** d_peer_recv = nullptr; int32_t* d_peer_slot = nullptr;
    int64_t* d_sent = nullptr; int sent_cap = 0;  // [T] records sent after each step of the last run (traffic accounting)
    // remote lineages (cpprob_hip_exchange_remote): migrants leave their history where it is; d_origin[annex column] = (rank << 32) | slot
    bool x_remote = false; int64_t* d_origin = nullptr; int64_t origin_cap = 0; RemoteStores* d_remote = nullptr; int64_t* d_annex_all = nullptr; int annex_all_T = 0;
    std::vector<uint64_t> x_shard_begin;
    int x_plan_t = -1;                            // step whose plan sits in d_xplan
    struct { int t = -1; bool resample = false; std::vector<uint64_t> send_lo, send_cnt; uint64_t n_send = 0, n_recv = 0; int64_t l0 = 0, l1 = 0; } plan;
    size_t cap_particles = 0; int cap_T = 0; bool cap_int = false; bool cap_multinomial = false;
    // prefix-count form of the step (table-weight models on an every-step schedule; step_counts.hpp)
    uint64_t* d_hier = nullptr; size_t hier_entries = 0;   // three copies of the 64-ary count hierarchy
    HierTable hier{};                                      // host copy of the hierarchy's layout
    HierTable* d_hier_table = nullptr;                     // the same in device memory
    size_t hier_per_copy = 0;                              // 64-bit words per copy
    int hier_phase = 0;                                    // copy that step 0 of the next run reads: chosen so that the copy it adds into is clean
    int hier_phase_run = 0;                                // ... of the run in flight
    bool hier_run_open = false;                            // a run's steps are in flight (the rotation's state is known only at run boundaries)
    int64_t* d_annex_base = nullptr;                       // [T + 1] exchange scope: annex columns in use before each step's immigrants
    int32_t* d_skip = nullptr;                             // [T / 8 + 1][rs] exchange scope, long traces: skip rows (exchange.hpp: skip_rows_kernel)
    bool counts_mode = false;                              // this run's steps use smc_step_counts_kernel
    bool fixed_mode = false;                               // ... or smc_step_fixed_kernel (fixed-point weights, step_fixed.hpp)
    bool final_from_fixed = false;                         // the read-out takes its weights from the final generation's integer weights
    // The fixed-point weights are taken against a reference fixed before the generation exists (step_fixed.hpp); where some
    // generation's heaviest particle sat far below it (an observation many standard deviations from every particle) they lose bits.
    // The first call that reads a run's results checks the run's largest gap and repeats the run in the floating-point form.
    bool force_fp = false, fixed_check_pending = false, last_was_infer_run = false; uint64_t last_run_index = 0;
    double* d_lz_trace = nullptr; int n_requantised = 0;    // fixed-point form: the evidence before each generation's books (what a repair rewinds to); generations repaired in the last run
    uint32_t* d_q[2] = {nullptr, nullptr};                 // [ld] integer weights of the fixed-point form, ping-pong; the count form's trace words
    // trace words (trace_words.hpp): a single population's short discrete traces ride with the particles; the read-out streams them
    bool trace_mode = false; uint32_t* d_trace_cnt = nullptr; unsigned long long* d_trace_arrive = nullptr;
    // ... and of one shard of a joint population (remote lineages): the words by the step's parity, [rs] each -- annex columns included,
    // a migrant's word arrives with its state; trace_shard: every rank of the group has them
    uint32_t* d_tr[2] = {nullptr, nullptr}; size_t tr_cap = 0; bool trace_shard = false, trace_shard_run = false;
    std::vector<double> h_bound;                           // [T] upper bound of each step's incremental log-weight (host-evaluated)
    int hk = 0; std::vector<double> hk_mean, hk_trans;     // cpprob_hip_set_hmm: the table of CPPROB_HIP_MODEL_HMM_TABLE
    uint64_t* d_hk_thr = nullptr; double* d_hk_ll = nullptr;
    size_t hier_q0_off = 0, hier_m0_off = 0;               // the tiles' Q / M arrays inside one copy of the hierarchy
    std::vector<double> h_ll_tab, h_e_tab;                 // host copies of the table-weight model's per-step tables ([T][3], [T][4])
    double* d_wpart = nullptr;                             // bounded SIS: per-workgroup partial rows
    bool sis_bounded_ok = false;                           // this run's model / observes admit the bounded-weight SIS kernel

    // scratch for building blocks
    Partial* d_bb_part = nullptr; double* d_bb_bc = nullptr; double* d_bb_bf = nullptr; double* d_bb_wrel = nullptr; void* d_bb_col = nullptr;
    StepCtrl* d_bb_ctrl = nullptr; size_t bb_cap_nb = 0;
    double* d_bb_stats_part = nullptr; double* d_bb_stats = nullptr; double* d_bb_cdf = nullptr; size_t bb_cdf_cap = 0;
    void* d_bb_cols = nullptr; double* d_bb_cols_part = nullptr; double* d_bb_cols_stat = nullptr; size_t bb_cols_bytes = 0, bb_cols_part = 0, bb_cols_stat = 0;   // several columns at once
    int32_t* d_bb_first = nullptr; size_t bb_first_cap = 0;
    std::vector<int32_t> bb_first_host;
    // exchange scope, mailbox collectives (group.hpp): the step's shard-totals launch carries the all-gather (device_collectives.hpp)
    bool x_gather_on = false, x_gather_done = false; TotalsGather x_gather{}; unsigned long long x_gather_serial = 0;
    int x_gather_tshift = 0;                 // (a run resumed behind a repaired generation numbers its collectives from there: group.hpp)                              // what d_bb_first holds (an unchanged table is not uploaded again)

    // cpprob_hip_smc_bookkeep_fixed: two alternating copies of a mass hierarchy + the integer weights
    uint64_t* d_bbf_hier = nullptr; HierTable* d_bbf_table = nullptr; HierTable bbf_table{}; uint32_t* d_bbf_q = nullptr;
    size_t bbf_per_copy = 0, bbf_q0_off = 0, bbf_m0_off = 0; int bbf_nb = 0, bbf_phase = 0; size_t bbf_cap_nb = 0;
    // cpprob_hip_generic_*: three rotating copies of a mass hierarchy, two generations of integer weights, a control block
    uint64_t* d_gen_hier = nullptr; HierTable* d_gen_table = nullptr; HierTable gen_table{}; uint32_t* d_gen_q[2] = {nullptr, nullptr}; void* d_gen_ctrl = nullptr;
    size_t gen_per_copy = 0, gen_q0_off = 0, gen_m0_off = 0, gen_off[3] = {0, 0, 0}; int gen_nb = 0, gen_block = 0; size_t gen_cap_hier = 0, gen_cap_q = 0;

    // optional per-kernel-class timing
    bool profile = false;
    bool profile_suspended = false;   // an enclosing group scope is timing these launches
    std::vector<EventPair> ev_used, ev_free;
    double prof_ms[CPPROB_HIP_N_KERNEL_CLASSES] = {0};
    int64_t prof_calls[CPPROB_HIP_N_KERNEL_CLASSES] = {0};
};

namespace {

int fail(cpprob_hip_ctx* ctx, int code, const std::string& msg)
{
    if (ctx) ctx->err = msg;
    g_last_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess)                                                                              \
            return fail(ctx, CPPROB_HIP_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e__));       \
    } while (0)

template <class T>
void dfree(T*& p)
{
    if (p) { (void)hipFree(p); p = nullptr; }
}

struct ProfScope {
    cpprob_hip_ctx* c; int cls; EventPair ep{}; bool on;
    // count = launches bracketed by this pair (back-to-back launches of one class are timed as a group, so the
    // event records do not sit between them)
    ProfScope(cpprob_hip_ctx* ctx, int k, int count = 1) : c(ctx), cls(k), on(ctx->profile && !ctx->profile_suspended)
    {
        if (!on) return;
        if (!c->ev_free.empty()) { ep = c->ev_free.back(); c->ev_free.pop_back(); }
        else { (void)hipEventCreate(&ep.a); (void)hipEventCreate(&ep.b); }
        ep.cls = cls; ep.count = count;
        (void)hipEventRecord(ep.a, c->stream);
    }
    ~ProfScope()
    {
        if (!on) return;
        (void)hipEventRecord(ep.b, c->stream);
        c->ev_used.push_back(ep);
    }
};

void host_model_params(ModelParams& mp, int model)
{
    std::memset(&mp, 0, sizeof mp);
    const double pi = 3.14159265358979323846;
    if (model == CPPROB_HIP_MODEL_GAUSSIAN_UNKNOWN_MEAN) {          // models.hpp:26-30
        mp.mu0 = 1; mp.sigma0 = std::sqrt(5.0); mp.sigma = std::sqrt(2.0);
    } else if (model == CPPROB_HIP_MODEL_GAUSSIAN_README) {         // gaussian.cpp:8
        mp.mu0 = 1; mp.sigma0 = 1.5; mp.sigma = 2;
    } else if (model == CPPROB_HIP_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN) { // models.hpp:42,44
        mp.nd_mean[0] = 1; mp.nd_mean[1] = 2; mp.nd_sigma[0] = std::sqrt(5.0); mp.nd_sigma[1] = std::sqrt(3.0); mp.sigma = std::sqrt(2.0);
    } else { mp.mu0 = 0; mp.sigma0 = 1; mp.sigma = 1; }
    mp.log_norm_lik = std::log(2 * pi * mp.sigma * mp.sigma);       // utils_normal_distribution.hpp:40
    mp.inv_sigma = 1.0 / mp.sigma;
    mp.log_norm_unit = std::log(2 * pi * 1.0 * 1.0);
    const double mean[3] = {-1, 0, 1};                               // models.hpp:122
    const double T[3][3] = {{0.1, 0.5, 0.4}, {0.2, 0.2, 0.6}, {0.15, 0.15, 0.7}};  // models.hpp:123-125
    for (int s = 0; s < 3; ++s) {
        mp.hmm_mean[s] = mean[s];
        double tot = 0.0;
        for (int j = 0; j < 3; ++j) tot += T[s][j];
        double acc = 0.0;
        for (int j = 0; j < 2; ++j) { acc += T[s][j]; mp.hmm_thr[s][j] = (uint64_t)std::ceil((acc / tot) * 4294967296.0); }   // u >= c  <=>  word >= ceil(c * 2^32)
    }
}

int model_T(int model, size_t n_obs)
{
    return (model == CPPROB_HIP_MODEL_GAUSSIAN_UNKNOWN_MEAN || model == CPPROB_HIP_MODEL_GAUSSIAN_README) ? 1 : (int)n_obs;
}

int ensure_bb(cpprob_hip_ctx* ctx, size_t n)
{
    const size_t nb = (n + kTile - 1) / kTile;
    if (nb > ctx->bb_cap_nb || !ctx->d_bb_ctrl) {
        dfree(ctx->d_bb_part); dfree(ctx->d_bb_bc); dfree(ctx->d_bb_bf); dfree(ctx->d_bb_wrel); dfree(ctx->d_bb_col); dfree(ctx->d_bb_stats_part);
        const size_t cap = std::max<size_t>(nb, 1024);
        HIP_TRY(ctx, hipMalloc(&ctx->d_bb_part, (size_t)part_stride((int)cap) * 3 * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_bb_bc, (cap + 1) * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_bb_bf, cap * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_bb_wrel, cap * kTile * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_bb_col, cap * kTile * sizeof(double)));
        HIP_TRY(ctx, hipMalloc(&ctx->d_bb_stats_part, 2048 * 8 * sizeof(double)));
        if (!ctx->d_bb_ctrl) HIP_TRY(ctx, hipMalloc(&ctx->d_bb_ctrl, sizeof(StepCtrl)));
        if (!ctx->d_bb_stats) HIP_TRY(ctx, hipMalloc(&ctx->d_bb_stats, 16 * sizeof(double)));
        ctx->bb_cap_nb = cap;
    }
    return 0;
}

// partials + linear weights + scan of an arbitrary logw array into the building-block scratch
int bb_normalise(cpprob_hip_ctx* ctx, const double* d_logw, size_t n, double n_total)
{
    if (int rc = ensure_bb(ctx, n)) return rc;
    const int nb = (int)((n + kTile - 1) / kTile);
    hipLaunchKernelGGL(weights_partials_kernel, dim3(nb), dim3(kThreads), 0, ctx->stream, d_logw, (int64_t)n, ctx->d_bb_part, ctx->d_bb_wrel);
    ScanArgs sa{};
    sa.part = ctx->d_bb_part; sa.nb = nb; sa.bc = ctx->d_bb_bc; sa.bf = ctx->d_bb_bf; sa.ctrl = ctx->d_bb_ctrl; sa.t = 0; sa.T = 1;
    sa.n_pop = n_total; sa.n_local = (double)n; sa.ess_frac = 0.0; sa.force_no_resample = 1; sa.phase = 0; sa.seed = 0;
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(kScanThreads), 0, ctx->stream, sa);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// the SIS read-out can ride the normalisation (no read-back of the particle store) when its columns are few
template <class Model>
bool sis_readout_fused(const cpprob_hip_ctx* c)
{
    return !(c->cfg.flags & CPPROB_HIP_FLAG_SIS_SEPARATE_READOUT) && c->T <= kMaxReadoutT && c->T * Model::kStats <= kMaxReadoutCols;
}

template <class Model>
void launch_sis(cpprob_hip_ctx* c, bool readout)
{
    SisArgs<Model> a{};
    a.mp = c->mp; a.obs = c->d_obs; a.T = c->T; a.n = c->n; a.ld = c->ld; a.rs = c->rs; a.seed = c->run_seed; a.pid0 = c->cfg.particle_offset;
    a.values = static_cast<typename Model::store_t*>(c->d_values); a.logw = c->d_logw[0]; a.wrel = c->d_wrel[0]; a.part = c->d_part[0];
    a.stile = c->d_stile;
    c->cur_part = 0;
    ProfScope ps(c, 4);
    if (readout) hipLaunchKernelGGL((sis_kernel<Model, true>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
    else hipLaunchKernelGGL((sis_kernel<Model, false>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
}

// normalisation of the final SIS weights + the weighted moments in the same two launches (slab form at every size)
template <class Model>
void launch_sis_readout(cpprob_hip_ctx* c)
{
    ScanArgs sa{};
    sa.part = c->d_part[c->cur_part]; sa.nb = c->nb; sa.bc = c->d_bc; sa.bf = c->d_bf; sa.ctrl = c->d_ctrl; sa.t = c->T - 1; sa.T = c->T;
    sa.n_pop = (double)c->pop_n; sa.n_local = (double)c->n; sa.ess_frac = c->cfg.ess_threshold; sa.seed = c->run_seed;
    sa.ess_trace = c->d_ess; sa.resampled = c->d_resampled; sa.force_no_resample = 1; sa.grid_refs = c->grid_refs ? 1 : 0; sa.phase = 0;
    const int G = (c->nb + kSlabTiles - 1) / kSlabTiles;
    const int n_col = c->T * Model::kStats;
    ProfScope ps(c, 1);
    hipLaunchKernelGGL(scan_slab_partials_kernel, dim3(G), dim3(kThreads), 0, c->stream, sa.part, c->nb, c->d_gpart, sa.grid_refs,
                       (const double*)c->d_stile, n_col, c->d_gstat);
    hipLaunchKernelGGL(scan_slab_finish_kernel, dim3(G), dim3(kThreads), 0, c->stream, sa, (const double*)c->d_gpart, G,
                       (const double*)c->d_gstat, n_col, (int)Model::kStats, Model::kIsInt ? 1 : 0, c->d_stats);
}

// Bounded-weight SIS (kernels.hpp: sis_bounded_kernel): the whole run in two launches.  CPPROB_HIP_FLAG_SIS_PER_TILE keeps the per-tile form.
template <class Model>
bool launch_sis_bounded(cpprob_hip_ctx* c)
{
    if constexpr (Model::kBounded) {
        if ((c->cfg.flags & CPPROB_HIP_FLAG_SIS_PER_TILE) || !c->sis_bounded_ok) return false;
        SisBoundedArgs<Model> a{};
        a.mp = c->mp; a.T = c->T; a.n = c->n; a.ld = c->ld; a.rs = c->rs; a.nb = c->nb; a.seed = c->run_seed; a.pid0 = c->cfg.particle_offset;
        a.values = static_cast<typename Model::store_t*>(c->d_values); a.logw = c->d_logw[0]; a.wpart = c->d_wpart;
        const int G = std::min(c->nb, (int)kSisBoundedMaxGrid);
        c->cur = 0; c->cur_part = 0;
        {
            ProfScope ps(c, 4);
            if (c->T == 1) hipLaunchKernelGGL((sis_bounded_kernel<Model, 1>), dim3(G), dim3(kThreads), 0, c->stream, a);
            else hipLaunchKernelGGL((sis_bounded_kernel<Model, 2>), dim3(G), dim3(kThreads), 0, c->stream, a);
        }
        ProfScope ps(c, 1);
        hipLaunchKernelGGL(sis_bounded_finish_kernel, dim3(1), dim3(kThreads), 0, c->stream, (const double*)c->d_wpart, G, c->T, (int)Model::kStats, c->mp.lw_ref,
                           (double)c->pop_n, c->d_ctrl, c->d_stats, c->d_ess, c->d_resampled, 1);
        return true;
    }
    return false;
}

template <class Model, int FUSED, bool COUNTS>
void launch_step_kernels(cpprob_hip_ctx* c, StepArgs<Model>& a)
{
    const size_t shm = FUSED ? (size_t)(2 * c->nb + 1) * sizeof(double) : 0;
    switch (c->cfg.resampler) {
    case CPPROB_HIP_RESAMPLE_SYSTEMATIC:
        hipLaunchKernelGGL((smc_step_kernel<Model, RS_SYSTEMATIC, FUSED, COUNTS>), dim3(c->nb), dim3(kThreads), shm, c->stream, a); break;
    case CPPROB_HIP_RESAMPLE_STRATIFIED:
        hipLaunchKernelGGL((smc_step_kernel<Model, RS_STRATIFIED, FUSED, COUNTS>), dim3(c->nb), dim3(kThreads), shm, c->stream, a); break;
    default:
        hipLaunchKernelGGL((smc_step_kernel<Model, RS_PRECOMPUTED, 0, false>), dim3(c->nb), dim3(kThreads), 0, c->stream, a); break;
    }
}

template <class Model, int FUSED>
void launch_step_impl(cpprob_hip_ctx* c, StepArgs<Model>& a)
{
    // packed-count partials exist for table-weight models in the fused form only (a.part_counts is set accordingly)
    if (Model::kWeightTable > 0 && FUSED != 0 && a.part_counts) launch_step_kernels<Model, FUSED, (Model::kWeightTable > 0 && FUSED != 0)>(c, a);
    else launch_step_kernels<Model, FUSED, false>(c, a);
}

// fused = the step kernel normalises the previous generation itself (no scan_partials launch between steps)
bool step_is_fused(const cpprob_hip_ctx* c)
{
    const int max_tiles = c->cfg.fuse_max_tiles > 0 ? std::min((int)c->cfg.fuse_max_tiles, (int)kFuseMaxTiles) : (int)kFuseMaxTiles;
    return c->nb <= max_tiles && c->cfg.resampler != CPPROB_HIP_RESAMPLE_MULTINOMIAL && !c->step_protocol;
}

template <class Model>
void launch_step(cpprob_hip_ctx* c, int t)
{
    StepArgs<Model> a{};
    a.mp = c->mp; a.obs = c->d_obs; a.t = t; a.T = c->T; a.n = c->n; a.ld = c->ld; a.seed = c->run_seed;
    a.pid0 = c->cfg.particle_offset;
    a.values = static_cast<typename Model::store_t*>(c->d_values); a.anc = c->d_anc;
    a.logw_prev = c->d_logw[c->cur]; a.logw_next = c->d_logw[c->cur ^ 1];
    a.wrel_prev = c->d_wrel[c->cur]; a.wrel_next = c->d_wrel[c->cur ^ 1];
    a.part_prev = c->d_part[c->cur_part]; a.part = c->d_part[t == 0 ? c->cur_part : c->cur_part ^ 1];
    a.bc = c->d_bc; a.bf = c->d_bf; a.nb = c->nb; a.ctrl = c->d_ctrl; a.anc_pre = c->d_anc_pre;
    a.n_pop = (double)c->pop_n; a.ess_frac = c->cfg.ess_threshold; a.ess_trace = c->d_ess; a.resampled = c->d_resampled;
    a.store_logw = (c->cfg.ess_threshold > 1.0 && c->keep) ? 0 : 1;      // ESS <= N always: threshold > 1 resamples after every step (a filtering-only run reads them after every step)
    a.rs = c->rs;
    a.row_w = c->keep ? t : (t & 1); a.row_r = t > 0 ? (c->keep ? t - 1 : ((t - 1) & 1)) : 0;
    if (!c->keep) a.anc = nullptr;
    {
        // table-weight model + every step resamples + systematic + fused: the step kernel reads states instead of wrel
        const bool enabled = !(c->cfg.flags & CPPROB_HIP_FLAG_WREL_STORED);
        // (pays from ~2.6e5 particles: below, the extra selects cost more than the halved traffic saves -- profiles/r01_ab_notes.md)
        const int min_tiles = 256;
        a.wrel_from_state = (enabled && Model::kWeightTable > 0 && c->cfg.ess_threshold > 1.0 && c->cfg.resampler == CPPROB_HIP_RESAMPLE_SYSTEMATIC &&
                             !c->step_protocol && c->nb >= min_tiles) ? 1 : 0;
        const bool counts = !(c->cfg.flags & CPPROB_HIP_FLAG_FP_TILE_PARTIALS);
        a.part_counts = (counts && a.wrel_from_state && step_is_fused(c)) ? 1 : 0;
    }
    a.exchange = (c->exchange && c->step_protocol) ? 1 : 0; a.imm_l01 = c->d_xplan ? &c->d_xplan->l0 : nullptr; a.annex_base = c->d_annex_base;
#ifdef CPPROB_STAMPS
    static unsigned long long* d_st = nullptr;
    if (!d_st) { (void)hipMalloc(&d_st, (size_t)131072 * 16 * 8); (void)hipMemset(d_st, 0, (size_t)131072 * 16 * 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)); }
#endif
    if (c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && t > 0) {
        // literal thesis Alg. 1: materialise the CDF, draw N independent positions.  Runs
        // unconditionally; the step kernel ignores the result when ctrl says "no resampling".
        ProfScope ps(c, 5);
        hipLaunchKernelGGL(cdf_kernel, dim3(c->nb), dim3(kThreads), 0, c->stream, c->d_wrel[c->cur], c->d_bc, c->d_bf, c->d_ctrl, c->d_cdf);
        hipLaunchKernelGGL(multinomial_kernel, dim3((unsigned)((c->ld + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, c->d_cdf, c->n,
                           c->d_ctrl, c->run_seed, (uint64_t)t, c->cfg.particle_offset, c->n, c->ld, c->d_anc_pre, 0);
    }
    {
        ProfScope ps(c, 0);
        if (!step_is_fused(c)) launch_step_impl<Model, 0>(c, a);
        else if (c->nb <= 2 * kThreads) launch_step_impl<Model, 2>(c, a);
        else if (c->nb <= 4 * kThreads) launch_step_impl<Model, 4>(c, a);
        else launch_step_impl<Model, 8>(c, a);
    }
#ifdef CPPROB_STAMPS
    if (t == 8 && getenv("CPPROB_STAMP_DUMP")) {
        (void)hipStreamSynchronize(c->stream);
        std::vector<unsigned long long> h((size_t)c->nb * 16);
        (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull; for (int b2 = 0; b2 < c->nb; ++b2) t0 = std::min(t0, h[(size_t)b2 * 16]);
        fprintf(stderr, "STAMPS t=8 n=%lld (us since first workgroup start) mean/max:", (long long)c->n);
        for (int k = 0; k < 12; ++k) {
            double acc = 0, mx = 0;
            for (int b2 = 0; b2 < c->nb; ++b2) { const double v = (double)(h[(size_t)b2 * 16 + k] - t0) * 0.01; acc += v; mx = std::max(mx, v); }
            fprintf(stderr, " [%d] %.2f/%.2f", k, acc / c->nb, mx);
        }
        fprintf(stderr, "\n");
    }
#endif
    c->cur ^= 1;
    if (t > 0) c->cur_part ^= 1;
    if (!c->keep) {
        // filtering only: predict hit t's sums under the weights this step just left
        hipLaunchKernelGGL(filter_partials_kernel<Model>, dim3(c->smooth_grid), dim3(kThreads), 0, c->stream,
                           static_cast<const typename Model::store_t*>(c->d_values) + (int64_t)a.row_w * c->rs, (const double*)c->d_logw[c->cur], c->n,
                           c->d_fpart + (size_t)t * (Model::kStats + 2) * c->smooth_grid);
    }
}

// The prefix-count form serves table-weight models (three values) whose every step resamples systematically: ancestors are
// then a function of integer counts (bit-exact against the oracle at any size), no step needs a normalisation launch, and the
// per-step prologue is one wavefront reduction.  CPPROB_HIP_FLAG_FLOATING_POINT_STEP keeps the floating-point form (A/B runs).
template <class Model>
bool counts_eligible(const cpprob_hip_ctx* c)
{
    return !(c->cfg.flags & CPPROB_HIP_FLAG_FLOATING_POINT_STEP) && c->nb <= kCountsMaxTiles && Model::kWeightTable == 3 && sizeof(typename Model::store_t) == 1 && c->cfg.algorithm == CPPROB_HIP_ALG_SMC &&
           c->cfg.ess_threshold > 1.0 &&
           (c->cfg.resampler == CPPROB_HIP_RESAMPLE_SYSTEMATIC
                ? (c->cfg.resample_scope == CPPROB_HIP_SCOPE_EXCHANGE || c->cfg.n_global == c->cfg.n_particles || c->cfg.resample_scope == CPPROB_HIP_SCOPE_ISLAND)
                // (stratified resampling: the same walk with the outputs' own uniforms)
                : c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED
                ? (c->cfg.resample_scope == CPPROB_HIP_SCOPE_EXCHANGE || c->cfg.n_global == c->cfg.n_particles || c->cfg.resample_scope == CPPROB_HIP_SCOPE_ISLAND)
                // (multinomial, strata form: a population of its own or one shard of the exchange scope; the literal form runs on fixed-point masses)
                : (!(c->cfg.flags & CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL) &&
                   (c->cfg.resample_scope == CPPROB_HIP_SCOPE_EXCHANGE || c->cfg.n_global == c->cfg.n_particles || c->cfg.resample_scope == CPPROB_HIP_SCOPE_ISLAND)));
}

// Philox4x32-10 on the host (cpprob/detail/rng.hpp's draw_block): the systematic offset of a resampling step is a pure function of
// (seed, step), so the launch carries it as a kernel argument instead of a device-side hand-over between consecutive kernels.
double host_resample_u0(uint64_t seed, uint64_t step)
{
    const uint64_t draw = (1ull << 40) + step, group = 0;
    uint32_t c0 = (uint32_t)draw, c1 = (uint32_t)(draw >> 32), c2 = (uint32_t)group, c3 = (uint32_t)(group >> 32);
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const uint64_t bits = (uint64_t)c0 | ((uint64_t)(c1 >> 11) << 32);
    return (double)bits * 1.1102230246251565e-16;
}

// The launch-side view of one copy of the count hierarchy (levels not in use alias level 0: see step_counts.hpp).
static void hier_view(const cpprob_hip_ctx* c, int copy, Hier& h)
{
    for (int l = 0; l < kHierMaxLevels; ++l) {
        h.lvl[l] = l < c->hier.n_lev ? c->hier.lvl[copy][l] : c->hier.lvl[copy][0];
        h.n_ent[l] = c->hier.n_ent[l];
    }
    h.n_lev = c->hier.n_lev; h.table = c->d_hier_table; h.copy = copy;
    const int top = c->hier.n_lev - 1;
    h.top = c->hier.lvl[copy][top]; h.top_n = c->hier.n_ent[top]; h.top_stride = top == 0 ? 1 : kHierStride;
}

static void hier_rotation(cpprob_hip_ctx* c, int t, int& kp, int& kn, int& kc);
static void launch_strata(cpprob_hip_ctx* c);
// multinomial resampling, strata form: the outputs the strata are drawn for -- the population's in the exchange scope, else this context's
static uint64_t strata_outputs(const cpprob_hip_ctx* c) { return c->exchange ? c->pop_n : (uint64_t)c->n; }
static int strata_k_of(const cpprob_hip_ctx* c) { return strata_levels((int64_t)((strata_outputs(c) + kTile - 1) / kTile)); }
static CutView cut_view(const cpprob_hip_ctx* c) { return CutView{c->d_cut_tab, c->d_cut_head, c->d_cut_srccnt}; }

template <class Model>
void launch_step_counts(cpprob_hip_ctx* c, int t, const double* all_totals, int world, int rank)
{
    if constexpr (Model::kWeightTable == 3 && sizeof(typename Model::store_t) == 1) {
        StepCountsArgs<Model> a{};
        a.mp = c->mp; a.t = t; a.T = c->T; a.n = c->n; a.ld = c->ld; a.rs = c->rs; a.seed = c->run_seed; a.pid0 = c->cfg.particle_offset;
        a.values = static_cast<typename Model::store_t*>(c->d_values); a.anc = c->d_anc;
        {
            int kp, kn, kc;
            hier_rotation(c, t, kp, kn, kc);
            hier_view(c, kp, a.h);
            a.h.to_next = (int64_t)(kn - kp) * (int64_t)c->hier_per_copy; a.h.to_clear = (int64_t)(kc - kp) * (int64_t)c->hier_per_copy;
            if (t == 0) c->final_from_counts = false;
            if (t + 1 == c->T) {
                // The last step is a step like any other (the read-out works from the counts it leaves): the copy it read and the one
                // it wrote are dirty, the one it cleared is where the next run's step 0 writes -- the next run starts its rotation
                // there, without any clearing launch.
                c->hier_phase = kn; c->final_from_counts = true; c->final_copy = kn; c->final_bookkeep_pending = true;
                c->hier_run_open = false;
            }
        }
        a.ctrl = c->d_ctrl; a.n_pop = (double)c->pop_n; a.ess_trace = c->d_ess; a.resampled = c->d_resampled;
        a.row_w = c->keep ? t : (t & 1); a.row_r = t > 0 ? (c->keep ? t - 1 : ((t - 1) & 1)) : 0;
        if (!c->keep) { a.anc = nullptr; a.filter_stats = c->d_stats; }
        a.all_totals = all_totals; a.world = world; a.rank = rank; a.annex_base = c->d_annex_base;
        a.src_shift = (c->exchange && c->x_fixed && !c->x_peers.empty() && c->d_xplan) ? &c->d_xplan->src_shift : nullptr;
        for (int k = 0; k < 4; ++k) a.e_prev[k] = t > 0 ? c->h_e_tab[(size_t)(t - 1) * 4 + k] : 0.0;
        a.u0 = t > 0 ? host_resample_u0(c->run_seed, (uint64_t)t) : 0.0;
        if (c->trace_mode && !all_totals) { a.trace_prev = c->d_q[(t + 1) & 1]; a.trace_next = c->d_q[t & 1]; }
        if (c->step_protocol && c->trace_shard_run) { a.trace_prev = c->d_tr[(t + 1) & 1]; a.trace_next = c->d_tr[t & 1]; }
        ProfScope ps(c, 0);
        if (c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL) {
            a.strata_k = strata_k_of(c);
            if (t == 0) launch_strata(c);
            a.strata_offs = t > 0 ? c->d_strata + (size_t)(t - 1) * (((size_t)1 << a.strata_k) + 1) : nullptr;
            a.cut = cut_view(c);
            if (all_totals) hipLaunchKernelGGL((smc_step_counts_kernel<Model, true, kFixMultinomial>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else hipLaunchKernelGGL((smc_step_counts_kernel<Model, false, kFixMultinomial>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
        }
        else if (c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED && all_totals) hipLaunchKernelGGL((smc_step_counts_kernel<Model, true, kFixStratified>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
        else if (c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED) hipLaunchKernelGGL((smc_step_counts_kernel<Model, false, kFixStratified>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
        else if (all_totals) hipLaunchKernelGGL((smc_step_counts_kernel<Model, true>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
        else hipLaunchKernelGGL((smc_step_counts_kernel<Model, false>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
#ifdef CPPROB_STAMPS
        {
            // diagnostic build: the phase stamps of step 8 (kernels.hpp: CPH_STAMP; 8 .. 11 = inside the multinomial walk), CPPROB_STAMP_DUMP=1
            static unsigned long long* d_st = nullptr;
            if (!d_st) { (void)hipMalloc(&d_st, (size_t)131072 * 16 * 8); (void)hipMemset(d_st, 0, (size_t)131072 * 16 * 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)); }
            if (t == 8 && getenv("CPPROB_STAMP_DUMP")) {
                (void)hipStreamSynchronize(c->stream);
                std::vector<unsigned long long> h((size_t)c->nb * 16);
                (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
                unsigned long long t0 = ~0ull; for (int b2 = 0; b2 < c->nb; ++b2) t0 = std::min(t0, h[(size_t)b2 * 16]);
                fprintf(stderr, "STAMPS counts t=8 n=%lld rs=%d (us since first workgroup start) mean/max:", (long long)c->n, (int)c->cfg.resampler);
                for (int k : {0, 1, 2, 8, 9, 10, 11, 3, 4, 5}) {
                    double acc = 0, mx = 0;
                    for (int b2 = 0; b2 < c->nb; ++b2) { const double v = (double)(h[(size_t)b2 * 16 + k] - t0) * 0.01; acc += v; mx = std::max(mx, v); }
                    fprintf(stderr, " [%d] %.2f/%.2f", k, acc / c->nb, mx);
                }
                fprintf(stderr, "\n");
                for (int k : {0, 2, 3, 5}) {
                    std::vector<double> v((size_t)c->nb);
                    for (int b2 = 0; b2 < c->nb; ++b2) v[(size_t)b2] = (double)(h[(size_t)b2 * 16 + k] - t0) * 0.01;
                    std::sort(v.begin(), v.end());
                    fprintf(stderr, "   stamp %d percentiles 10/50/75/90/95/99/100: %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n", k, v[v.size() / 10], v[v.size() / 2], v[v.size() * 3 / 4],
                            v[v.size() * 9 / 10], v[v.size() * 95 / 100], v[v.size() * 99 / 100], v.back());
                }
                {
                    std::vector<int> order((size_t)c->nb);
                    for (int b2 = 0; b2 < c->nb; ++b2) order[(size_t)b2] = b2;
                    std::sort(order.begin(), order.end(), [&](int x, int y) { return h[(size_t)x * 16 + 5] > h[(size_t)y * 16 + 5]; });
                    for (int r = 0; r < 8 && r < c->nb; ++r) {
                        const int b2 = order[(size_t)r];
                        fprintf(stderr, "   slow workgroup %d:", b2);
                        for (int k : {0, 1, 2, 3, 4, 5}) fprintf(stderr, " [%d] %.2f", k, (double)(h[(size_t)b2 * 16 + k] - t0) * 0.01);
                        fprintf(stderr, "\n");
                    }
                }
            }
        }
#endif
        if (t + 1 == c->T) { c->cur = 0; c->cur_part = 0; }
    }
}

// The fixed-point form (step_fixed.hpp) serves what the prefix-count form does not: continuous weights and ESS-triggered schedules
// under systematic resampling, one population per context or one shard of an exchange-scope population.  Same hierarchy, same
// rotation; ancestors, decisions and evidence are integer-exact across tilings and shardings.
template <class Model>
bool fixed_eligible(const cpprob_hip_ctx* c)
{
    constexpr bool model_ok = std::is_same<Model, ModelLinearGaussian1D>::value || std::is_same<Model, ModelHmm3>::value || std::is_same<Model, ModelHmmK>::value;
    // (stratified and multinomial resampling run on the same integer masses; multinomial: one population per context)
    const bool own = c->cfg.n_global == c->cfg.n_particles || c->cfg.resample_scope == CPPROB_HIP_SCOPE_ISLAND;
    // (one interval of outputs per rank -- the strata form of multinomial resampling: one interval + the strata the ranks' boundaries cut)
    const bool systematic = c->cfg.resampler == CPPROB_HIP_RESAMPLE_SYSTEMATIC || c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED ||
                            (c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && !(c->cfg.flags & CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL));
    return model_ok && !c->force_fp && !(c->cfg.flags & CPPROB_HIP_FLAG_FLOATING_POINT_STEP) && c->nb <= kCountsMaxTiles && c->pop_n <= (1ull << 28) &&
           c->cfg.algorithm == CPPROB_HIP_ALG_SMC && !counts_eligible<Model>(c) &&
           (systematic ? (c->cfg.resample_scope == CPPROB_HIP_SCOPE_EXCHANGE || own) : (own && c->cfg.resample_scope != CPPROB_HIP_SCOPE_EXCHANGE));
}

static void fhier_view(const cpprob_hip_ctx* c, int copy, FHier& f)
{
    hier_view(c, copy, f.h);
    f.q0 = c->d_hier + (size_t)copy * c->hier_per_copy + c->hier_q0_off;
    f.m0 = c->d_hier + (size_t)copy * c->hier_per_copy + c->hier_m0_off;
}

// the copies step t reads / writes / clears (shared by both integer forms)
static void hier_rotation(cpprob_hip_ctx* c, int t, int& kp, int& kn, int& kc)
{
    if (t == 0) {
        // a run that was abandoned half-way leaves the rotation in an unknown state: start over from clean copies
        if (c->hier_run_open) { (void)hipMemsetAsync(c->d_hier, 0, c->hier_entries * sizeof(uint64_t), c->stream); c->hier_phase = 0; }
        c->hier_run_open = true; c->hier_phase_run = c->hier_phase;
    }
    kp = (t + c->hier_phase_run) % 3; kn = (kp + 1) % 3; kc = (kp + 2) % 3;
}

// Multinomial resampling, strata form (step_fixed.hpp): how many thresholds fall into each stratum does not depend on the weights --
// every step's counts in ONE launch (two above 64 tiles) in front of the run's first step.
static void launch_strata(cpprob_hip_ctx* c)
{
    if (!c->strata_pending) return;
    c->strata_pending = false;
    if (c->T < 2 || !c->d_strata) return;
    ProfScope ps(c, 5);
    StrataArgs sa{};
    // (a shard of a joint population draws the POPULATION's strata: every rank holds the whole table, a function of the seed alone)
    sa.seed = c->run_seed; sa.t0 = 1; sa.k = strata_k_of(c); sa.n_out = (uint32_t)strata_outputs(c); sa.offs = c->d_strata;
    if (sa.k <= kStrataTop) hipLaunchKernelGGL(multinomial_strata_kernel, dim3(1, c->T - 1), dim3(kThreads), 0, c->stream, sa);
    else {
        // (two sets of the steps' level-6 totals: the bottom launch clears the set the NEXT run adds into)
        sa.top = c->d_strata_top + (size_t)c->strata_phase * (size_t)c->T * 64; sa.top_clear = c->d_strata_top + (size_t)(c->strata_phase ^ 1) * (size_t)c->T * 64;
        c->strata_phase ^= 1;
        hipLaunchKernelGGL(multinomial_strata_top_kernel, dim3(strata_groups(sa.k), c->T - 1), dim3(kThreads), 0, c->stream, sa);
        hipLaunchKernelGGL(multinomial_strata_bottom_kernel, dim3(64, c->T - 1), dim3(kThreads), 0, c->stream, sa);
    }
}

template <class Model>
void launch_step_fixed(cpprob_hip_ctx* c, int t, const double* all_totals, int world, int rank)
{
    if constexpr (std::is_same<Model, ModelLinearGaussian1D>::value || std::is_same<Model, ModelHmm3>::value || std::is_same<Model, ModelHmmK>::value) {
        StepFixedArgs<Model> a{};
        a.mp = c->mp; a.obs = c->d_obs; a.t = t; a.T = c->T; a.n = c->n; a.ld = c->ld; a.rs = c->rs; a.seed = c->run_seed; a.pid0 = c->cfg.particle_offset;
        a.values = static_cast<typename Model::store_t*>(c->d_values); a.anc = c->d_anc;
        int kp, kn, kc;
        hier_rotation(c, t, kp, kn, kc);
        fhier_view(c, kp, a.f);
        a.f.h.to_next = (int64_t)(kn - kp) * (int64_t)c->hier_per_copy; a.f.h.to_clear = (int64_t)(kc - kp) * (int64_t)c->hier_per_copy;
        if (t == 0) { c->final_from_fixed = false; c->cur = 0; }
        if (t + 1 == c->T) { c->hier_phase = kn; c->final_from_fixed = true; c->final_copy = kn; c->final_bookkeep_pending = true; c->hier_run_open = false; }
        a.q_prev = c->d_q[c->cur]; a.q_next = c->d_q[c->cur ^ 1];
        a.logw_prev = c->d_logw[c->cur]; a.logw_next = c->d_logw[c->cur ^ 1];
        a.u0 = t > 0 ? host_resample_u0(c->run_seed, (uint64_t)t) : 0.0;
        a.bound = c->h_bound[(size_t)t]; a.bound_prev = t > 0 ? c->h_bound[(size_t)t - 1] : 0.0;
        a.ess_frac = c->cfg.ess_threshold; a.may_carry = c->cfg.ess_threshold > 1.0 ? 0 : 1;
        a.prefetch = a.may_carry ? 0 : 1;                        // (a schedule on which steps may not resample: the fetch is wasted on those, and costs registers on all)
        const int rs = c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED ? kFixStratified
                     : (c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL ? ((c->cfg.flags & CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL) ? kFixMultinomialLiteral : kFixMultinomial) : kFixSystematic);
        if (rs == kFixMultinomial) {
            a.prefetch = 0;
            a.strata_k = strata_k_of(c);
            if (t == 0) launch_strata(c);
            a.strata_offs = t > 0 ? c->d_strata + (size_t)(t - 1) * (((size_t)1 << a.strata_k) + 1) : nullptr;
            a.cut = cut_view(c);
        }
        if (rs == kFixMultinomialLiteral) {
            // the lanes' in-tile prefixes travel with the weights (ping-pong, in the halves of the floating-point form's CDF array);
            // the tiles' prefix masses of generation t-1 come from one short launch in front of the step
            uint64_t* lp = reinterpret_cast<uint64_t*>(c->d_cdf);
            a.lane_prefix_prev = lp + (size_t)c->cur * (size_t)(c->ld / kPPT); a.lane_prefix_next = lp + (size_t)(c->cur ^ 1) * (size_t)(c->ld / kPPT);
            a.tile_prefix = reinterpret_cast<const uint64_t*>(c->d_bc);
            a.prefetch = 0;
            if (t > 0) {
                ProfScope ps(c, 5);
                hipLaunchKernelGGL(fixed_tile_prefix_kernel, dim3(1), dim3(kTilePrefixThreads), 0, c->stream, a.f.h.lvl[0], c->nb, reinterpret_cast<uint64_t*>(c->d_bc));
            }
        }
        a.ctrl = c->d_ctrl; a.n_pop = (double)c->pop_n; a.ess_trace = c->d_ess; a.resampled = c->d_resampled;
        a.all_totals = reinterpret_cast<const uint64_t*>(all_totals); a.world = world; a.rank = rank; a.annex_base = c->d_annex_base;
        a.row_w = c->keep ? t : (t & 1); a.row_r = t > 0 ? (c->keep ? t - 1 : ((t - 1) & 1)) : 0;
        if (!c->keep) a.anc = nullptr;
        {
            ProfScope ps(c, 0);
            // A/B form (CPPROB_HIP_FLAG_PAIRED_STEP_LAUNCH): on schedules where a step may not resample, the step as two launches, each ending
            // at once when the step is the other's (step_fixed.hpp: smc_step_fixed_carry_body).  Measured at configs[4]'s shard: the carry
            // launch takes 70 us where the two-form kernel's non-resampling launches take 81, and the second launch costs what that saves.
            // (the A/B form has no sharded build of the stratified / multinomial resampling launch: those shards take the one-launch step)
            const bool paired = a.may_carry && t > 0 && (c->cfg.flags & CPPROB_HIP_FLAG_PAIRED_STEP_LAUNCH) && rs != kFixMultinomialLiteral &&
                                !(all_totals != nullptr && rs != kFixSystematic);
            if (paired) {
                const bool sh = all_totals != nullptr;
                if (sh) hipLaunchKernelGGL((smc_step_fixed_carry_kernel<Model, true>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
                else hipLaunchKernelGGL((smc_step_fixed_carry_kernel<Model, false>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
                if (rs == kFixMultinomial) hipLaunchKernelGGL((smc_step_fixed_resampling_kernel<Model, false, kFixMultinomial>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
                else if (rs == kFixStratified && !sh) hipLaunchKernelGGL((smc_step_fixed_resampling_kernel<Model, false, kFixStratified>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
                else if (sh) hipLaunchKernelGGL((smc_step_fixed_resampling_kernel<Model, true, kFixSystematic>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
                else hipLaunchKernelGGL((smc_step_fixed_resampling_kernel<Model, false, kFixSystematic>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            }
            else if (rs == kFixMultinomial && all_totals) hipLaunchKernelGGL((smc_step_fixed_sharded_kernel<Model, false, kFixMultinomial>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (rs == kFixMultinomial) hipLaunchKernelGGL((smc_step_fixed_kernel<Model, false, kFixMultinomial>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (rs == kFixMultinomialLiteral) hipLaunchKernelGGL((smc_step_fixed_kernel<Model, false, kFixMultinomialLiteral>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (rs == kFixStratified && all_totals && a.prefetch) hipLaunchKernelGGL((smc_step_fixed_sharded_kernel<Model, true, kFixStratified>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (rs == kFixStratified && all_totals) hipLaunchKernelGGL((smc_step_fixed_sharded_kernel<Model, false, kFixStratified>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (rs == kFixStratified && a.prefetch) hipLaunchKernelGGL((smc_step_fixed_kernel<Model, true, kFixStratified>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (rs == kFixStratified && !Model::kIsInt) hipLaunchKernelGGL((smc_step_fixed_five_kernel<Model, kFixStratified>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (rs == kFixStratified) hipLaunchKernelGGL((smc_step_fixed_kernel<Model, false, kFixStratified>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (all_totals && a.prefetch) hipLaunchKernelGGL((smc_step_fixed_sharded_kernel<Model, true>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (all_totals) hipLaunchKernelGGL((smc_step_fixed_sharded_kernel<Model, false>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (a.prefetch) hipLaunchKernelGGL((smc_step_fixed_kernel<Model, true>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else if (!Model::kIsInt) hipLaunchKernelGGL((smc_step_fixed_five_kernel<Model>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
            else hipLaunchKernelGGL((smc_step_fixed_kernel<Model, false>), dim3(c->nb), dim3(kThreads), 0, c->stream, a);
        }
#ifdef CPPROB_STAMPS
        {
            static unsigned long long* d_st = nullptr;
            if (!d_st) { (void)hipMalloc(&d_st, (size_t)131072 * 16 * 8); (void)hipMemset(d_st, 0, (size_t)131072 * 16 * 8); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st)); }
            if (t >= 40 && t < 48 && getenv("CPPROB_STAMP_DUMP")) {
                (void)hipStreamSynchronize(c->stream);
                std::vector<unsigned long long> h((size_t)c->nb * 16);
                (void)hipMemcpy(h.data(), d_st, h.size() * 8, hipMemcpyDeviceToHost);
                int32_t rs_prev = 0; (void)hipMemcpy(&rs_prev, c->d_resampled + (t - 1), 4, hipMemcpyDeviceToHost);
                unsigned long long t0 = ~0ull; for (int b2 = 0; b2 < c->nb; ++b2) t0 = std::min(t0, h[(size_t)b2 * 16]);
                fprintf(stderr, "STAMPS fixed t=%d resampled=%d n=%lld (us since first workgroup start) mean/max:", t, rs_prev, (long long)c->n);
                for (int k = 0; k < 8; ++k) {
                    double acc = 0, mx = 0;
                    for (int b2 = 0; b2 < c->nb; ++b2) { const double v = (double)(h[(size_t)b2 * 16 + k] - t0) * 0.01; acc += v; mx = std::max(mx, v); }
                    fprintf(stderr, " [%d] %.2f/%.2f", k, acc / c->nb, mx);
                }
                fprintf(stderr, "\n");
                for (int k : {0, 7}) {
                    std::vector<double> v((size_t)c->nb);
                    for (int b2 = 0; b2 < c->nb; ++b2) v[(size_t)b2] = (double)(h[(size_t)b2 * 16 + k] - t0) * 0.01;
                    std::sort(v.begin(), v.end());
                    fprintf(stderr, "   stamp %d percentiles 10/25/50/75/80/85/90/95/99/100: %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n", k, v[v.size() / 10], v[v.size() / 4], v[v.size() / 2],
                            v[v.size() * 3 / 4], v[v.size() * 8 / 10], v[v.size() * 85 / 100], v[v.size() * 9 / 10], v[v.size() * 95 / 100], v[v.size() * 99 / 100], v.back());
                }
            }
        }
#endif
        c->cur ^= 1;
        if (!c->keep) {
            // filtering only: predict hit t's sums under the integer weights this step just left
            hipLaunchKernelGGL(filter_partials_fixed_kernel<Model>, dim3(c->smooth_grid), dim3(kThreads), 0, c->stream,
                               static_cast<const typename Model::store_t*>(c->d_values) + (int64_t)a.row_w * c->rs, (const uint32_t*)c->d_q[c->cur], c->n,
                               c->d_fpart + (size_t)t * (Model::kStats + 2) * c->smooth_grid);
        }
    }
}

static void fixed_final_view(cpprob_hip_ctx* c, FixedFinal& ff, bool bookkeep)
{
    fhier_view(c, c->final_copy, ff.f);
    ff.n_pop = (double)c->pop_n; ff.ess_frac = c->cfg.ess_threshold; ff.T = c->T; ff.bookkeep = bookkeep ? 1 : 0;
    ff.ctrl = c->d_ctrl; ff.ess_trace = c->d_ess; ff.resampled = c->d_resampled;
}

// The final generation's counts, as the read-out and the joint bookkeeping see them.
static void counts_final_view(cpprob_hip_ctx* c, CountsFinal& f, bool bookkeep)
{
    hier_view(c, c->final_copy, f.h);
    for (int k = 0; k < 4; ++k) f.e[k] = c->h_e_tab[(size_t)(c->T - 1) * 4 + k];
    f.n_pop = (double)c->pop_n; f.T = c->T; f.bookkeep = bookkeep ? 1 : 0; f.ctrl = c->d_ctrl; f.ess_trace = c->d_ess; f.resampled = c->d_resampled;
    f.filter_stats = c->keep ? nullptr : c->d_stats;
}

ScanArgs make_scan_args(cpprob_hip_ctx* c, int t, int phase, const double* all_totals, int world, int rank)
{
    ScanArgs sa{};
    sa.part = c->d_part[c->cur_part]; sa.nb = c->nb; sa.bc = c->d_bc; sa.bf = c->d_bf; sa.ctrl = c->d_ctrl; sa.t = t; sa.T = c->T;
    sa.n_pop = (double)c->pop_n; sa.n_local = (double)c->n; sa.ess_frac = c->cfg.ess_threshold; sa.seed = c->run_seed;
    sa.ess_trace = c->d_ess; sa.resampled = c->d_resampled;
    sa.force_no_resample = c->cfg.algorithm == CPPROB_HIP_ALG_SIS ? 1 : 0;
    sa.grid_refs = c->grid_refs ? 1 : 0;
    sa.exchange = (c->exchange && phase == 2) ? 1 : 0; sa.obound = c->d_obound;
    sa.all_totals = all_totals; sa.world = world; sa.rank = rank; sa.local_totals = c->totals_out ? c->totals_out : c->d_local_totals; sa.phase = phase;
    return sa;
}

void launch_scan(cpprob_hip_ctx* c, int t, int phase, const double* all_totals, int world, int rank)
{
    ScanArgs sa = make_scan_args(c, t, phase, all_totals, world, rank);
    ProfScope ps(c, 1);
    if (phase != 2 && c->nb > kSlabThreshold) {
        // large population: two multi-workgroup launches instead of one single-CU pass
        const int G = (c->nb + kSlabTiles - 1) / kSlabTiles;
        hipLaunchKernelGGL(scan_slab_partials_kernel, dim3(G), dim3(kThreads), 0, c->stream, sa.part, c->nb, c->d_gpart, sa.grid_refs);
        hipLaunchKernelGGL(scan_slab_finish_kernel, dim3(G), dim3(kThreads), 0, c->stream, sa, (const double*)c->d_gpart, G);
    } else {
        hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(kScanThreads), 0, c->stream, sa);
    }
}


// Workgroups of the trace-word read-out: each adds its counts into the shared counters (~100 atomics), so few of them while the pass
// is short -- 256 up to 4e6 particles (10^6: 12.6 us; 1024 workgroups: 21) -- and one per sixteen tiles beyond, up to 2048
// (10^8 particles: 430 -> 244 us).
static int trace_readout_grid(int64_t n)
{
    const int64_t tiles = (n + kTile - 1) / kTile;
    return (int)std::min<int64_t>(tiles, std::max<int64_t>(256, std::min<int64_t>(2048, tiles / 16)));
}

template <class Model>
void launch_smooth(cpprob_hip_ctx* c, bool with_paths)
{
    SmoothArgs<Model> a{};
    a.values = static_cast<const typename Model::store_t*>(c->d_values); a.anc = c->d_anc; a.wrel = c->d_wrel[c->cur]; a.bf = c->d_bf; a.ctrl = c->d_ctrl;
    a.resampled = c->d_resampled; a.T = c->T; a.n = c->n; a.ld = c->ld; a.rs = c->rs;
    a.identity = c->cfg.algorithm == CPPROB_HIP_ALG_SIS ? 1 : 0;
    a.stats_part = c->d_stats_part;
    a.paths = with_paths ? static_cast<typename Model::value_t*>(c->d_paths) : nullptr;
    a.rem = (c->x_remote && c->sharded) ? c->d_remote : nullptr;
    const size_t shm = (size_t)kWaves * c->T * Model::kStats * sizeof(double);
    // The walk's grid.  Its workgroups loop over tiles (kSmoothTiles at a time for one-byte states), so what a launch costs is
    // passes x rounds: a grid that is not a multiple of what the chip holds at once leaves its last round part-empty (2048 workgroups
    // at six a CU: 1536 + 512), and one that divides the tiles badly gives some workgroups a pass more than others.  Chosen once per
    // configuration: k x residency, k = 1 .. 4, with the least padded capacity (ties: the larger grid -- shorter chains per workgroup).
    // hmm<128> at 1.25e7 particles: 2048 -> 6144 workgroups, 1.38 -> 1.20 ms; linear_gaussian_1d<100> at 1e7 keeps 2048 (eight a CU).
    auto choose_grid = [&](auto kernel, int tiles_at_a_time = kSmoothTiles) -> int {
        if (c->walk_grid > 0) return c->walk_grid;
        int grid = c->smooth_grid, occ = 0;
        if (c->n_cu == 0) { int v = 0; if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess) c->n_cu = v; }
        if (c->n_cu > 0 && hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, kThreads, shm) == hipSuccess && occ > 0) {
            const int64_t R = (int64_t)occ * c->n_cu, nt = c->nb;
            if (nt <= R) grid = (int)nt;
            else {
                double best = 1e300;
                for (int k = 1; k <= 4; ++k) {
                    const int64_t g = R * k;
                    if (g > c->walk_cap) break;
                    const int64_t per = (sizeof(typename Model::store_t) == 1 && tiles_at_a_time > 1 && nt > g) ? tiles_at_a_time : 1;
                    const int64_t passes = (nt + g * per - 1) / (g * per);
                    const double waste = (double)(passes * g * per) / (double)nt;
                    if (waste <= best * 1.01) { best = std::min(best, waste); grid = (int)std::min<int64_t>(g, nt); }
                }
            }
        }
        (void)hipGetLastError();
        c->walk_grid = grid;
        return grid;
    };
    int grid_used = c->smooth_grid;
    {
        ProfScope ps(c, 2);
        bool launched = false;
        if constexpr (Model::kWeightTable == 3 && sizeof(typename Model::store_t) == 1) {
            if (c->final_from_counts && c->sharded && c->trace_shard_run && !with_paths) {
                // one shard of a joint population: the same counting over this shard's words, un-normalised (the ranks' sums meet in the
                // run's final all-reduce); the final generation's bookkeeping came from the all-gathered totals
                TraceReadoutArgs ta{};
                ta.trace = c->d_tr[(c->T - 1) & 1]; ta.n = c->n; ta.T = c->T;
                counts_final_view(c, ta.f, false);
                ta.counters = c->d_trace_cnt; ta.arrive = c->d_trace_arrive; ta.stats = c->d_stats; ta.raw = 1; ta.n_local = (double)c->n;
                const int grid = trace_readout_grid(c->n);
                hipLaunchKernelGGL(trace_readout_kernel, dim3(grid), dim3(kThreads), 0, c->stream, ta);
                return;
            }
            if (c->final_from_counts && c->trace_mode && !with_paths) {
                // every particle carries its trace: one streaming pass, integer counts, statistics written by the last workgroup
                TraceReadoutArgs ta{};
                ta.trace = c->d_q[(c->T - 1) & 1]; ta.n = c->n; ta.T = c->T;
                counts_final_view(c, ta.f, c->final_bookkeep_pending);
                c->final_bookkeep_pending = false;
                ta.counters = c->d_trace_cnt; ta.arrive = c->d_trace_arrive; ta.stats = c->d_stats;
                const int grid = trace_readout_grid(c->n);
                hipLaunchKernelGGL(trace_readout_kernel, dim3(grid), dim3(kThreads), 0, c->stream, ta);
                return;
            }
            if (c->final_from_counts) {
                CountsFinal f{};
                counts_final_view(c, f, c->final_bookkeep_pending && !with_paths);
                if (!with_paths) c->final_bookkeep_pending = false;
                grid_used = choose_grid(smooth_counts_kernel<Model>);
                hipLaunchKernelGGL(smooth_counts_kernel<Model>, dim3(grid_used), dim3(kThreads), shm, c->stream, a, f);
                launched = true;
            }
        }
        if (!launched && c->final_from_fixed) {
            FixedFinal ff{};
            fixed_final_view(c, ff, c->final_bookkeep_pending && !with_paths);
            if (!with_paths) c->final_bookkeep_pending = false;
            bool int_walk = false;
            if constexpr (Model::kIsInt) {
                if (!a.rem) {                                            // every lineage on this device: the integer walk (step_fixed.hpp)
                    int_walk = true;
                    if (a.paths) {
                        grid_used = choose_grid(smooth_fixed_int_kernel<Model, true>, kSmoothTilesInt);
                        hipLaunchKernelGGL((smooth_fixed_int_kernel<Model, true>), dim3(grid_used), dim3(kThreads), shm, c->stream, a, ff, (const uint32_t*)c->d_q[c->cur]);
                    } else {
                        grid_used = choose_grid(smooth_fixed_int_kernel<Model, false>, kSmoothTilesInt);
                        hipLaunchKernelGGL((smooth_fixed_int_kernel<Model, false>), dim3(grid_used), dim3(kThreads), shm, c->stream, a, ff, (const uint32_t*)c->d_q[c->cur]);
                    }
                }
            }
            if (!int_walk) {
                grid_used = choose_grid(smooth_fixed_kernel<Model>);
                hipLaunchKernelGGL(smooth_fixed_kernel<Model>, dim3(grid_used), dim3(kThreads), shm, c->stream, a, ff, (const uint32_t*)c->d_q[c->cur]);
            }
            launched = true;
        }
        if (!launched) { grid_used = choose_grid(smooth_kernel<Model>); hipLaunchKernelGGL(smooth_kernel<Model>, dim3(grid_used), dim3(kThreads), shm, c->stream, a); }
    }
    if (with_paths && c->trace_mode && c->final_from_counts) return;     // (the statistics came from the trace words: the walk only materialises traces)
    ProfScope ps(c, 3);
    hipLaunchKernelGGL(finalize_kernel, dim3(c->T), dim3(kThreads), 0, c->stream,
                       c->d_stats_part, grid_used, c->T, Model::kStats, Model::kIsInt ? 1 : 0, c->d_ctrl, c->d_stats, c->sharded ? 0 : 1);
}

template <class F>
int dispatch_model(cpprob_hip_ctx* c, F&& f)
{
    switch (c->cfg.model) {
    case CPPROB_HIP_MODEL_GAUSSIAN_UNKNOWN_MEAN:
    case CPPROB_HIP_MODEL_GAUSSIAN_README: f(ModelGaussian{}); return 0;
    case CPPROB_HIP_MODEL_LINEAR_GAUSSIAN_1D: f(ModelLinearGaussian1D{}); return 0;
    case CPPROB_HIP_MODEL_HMM3: f(ModelHmm3{}); return 0;
    case CPPROB_HIP_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN: f(ModelGaussianND{}); return 0;
    case CPPROB_HIP_MODEL_HMM_TABLE: f(ModelHmmK{}); return 0;
    }
    return fail(c, CPPROB_HIP_EINVAL, "unknown model id");
}

void free_run_buffers(cpprob_hip_ctx* c)
{
    dfree(c->d_obs); dfree(c->d_logw[0]); dfree(c->d_logw[1]); dfree(c->d_wrel[0]); dfree(c->d_wrel[1]); dfree(c->d_bf); dfree(c->d_ll_tab); dfree(c->d_values); dfree(c->d_anc); dfree(c->d_paths);
    dfree(c->d_part[0]); dfree(c->d_part[1]); dfree(c->d_e_tab); dfree(c->d_gpart); dfree(c->d_stile); dfree(c->d_gstat); dfree(c->d_bc); dfree(c->d_ess); dfree(c->d_resampled); dfree(c->d_stats_part); dfree(c->d_stats);
    dfree(c->d_cdf); dfree(c->d_anc_pre); dfree(c->d_strata); dfree(c->d_strata_top); dfree(c->d_lz_trace); dfree(c->d_obound); dfree(c->d_hier); dfree(c->d_annex_base); dfree(c->d_fpart); dfree(c->d_filter_w); dfree(c->d_skip); dfree(c->d_q[0]); dfree(c->d_q[1]); dfree(c->d_trace_cnt); dfree(c->d_trace_arrive); dfree(c->d_tr[0]); dfree(c->d_tr[1]);
    c->cap_particles = 0; c->cap_T = 0; c->annex_cap = 0; c->tr_cap = 0;
}

}  // namespace

extern "C" {

int cpprob_hip_abi_version(void) { return CPPROB_HIP_ABI_VERSION; }

#ifndef CPPROB_BUILD_ID
#define CPPROB_BUILD_ID "unknown"
#endif
const char* cpprob_hip_build_id(void) { return CPPROB_BUILD_ID; }

int cpprob_hip_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e); return CPPROB_HIP_EDEVICE; }
    return n;
}

const char* cpprob_hip_last_error(const cpprob_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int cpprob_hip_create(int device, cpprob_hip_ctx** out)
{
    if (!out) return fail(nullptr, CPPROB_HIP_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, CPPROB_HIP_EDEVICE, std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                                                     "); this engine has no CPU fallback");
    if (device < 0 || device >= n) return fail(nullptr, CPPROB_HIP_EINVAL, "device index out of range");
    cpprob_hip_ctx* c = new cpprob_hip_ctx();
    c->device = device;
    const int rc = [&]() -> int {
        HIP_TRY(nullptr, hipSetDevice(device));
        HIP_TRY(nullptr, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        HIP_TRY(nullptr, hipMalloc(&c->d_ctrl, sizeof(StepCtrl)));
        HIP_TRY(nullptr, hipMalloc(&c->d_local_totals, 4 * sizeof(double)));
        HIP_TRY(nullptr, hipMemsetAsync(c->d_ctrl, 0, sizeof(StepCtrl), c->stream));
        return 0;
    }();
    if (rc) { cpprob_hip_destroy(c); return rc; }     // nothing leaks on a failed create
    *out = c;
    return 0;
}

void cpprob_hip_destroy(cpprob_hip_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_run_buffers(c);
    dfree(c->d_ctrl); dfree(c->d_local_totals); dfree(c->d_cut_tab); dfree(c->d_cut_head); dfree(c->d_cut_srccnt);
    dfree(c->d_send_src); dfree(c->d_hk_thr); dfree(c->d_hk_ll); dfree(c->d_hier_table); dfree(c->d_wpart); dfree(c->d_xplan); dfree(c->d_shard_begin); dfree(c->d_slot_of_rank); dfree(c->d_xsend); dfree(c->d_xrecv); dfree(c->d_peer_recv); dfree(c->d_peer_slot); dfree(c->d_sent); dfree(c->d_origin); dfree(c->d_remote); dfree(c->d_annex_all);
    if (c->h_obound) { (void)hipHostFree(c->h_obound); c->h_obound = nullptr; }
    if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = nullptr; c->h_pin_cap = 0; }
    dfree(c->d_bb_part); dfree(c->d_bb_bc); dfree(c->d_bb_bf); dfree(c->d_bb_wrel); dfree(c->d_bb_col); dfree(c->d_bb_ctrl); dfree(c->d_bb_stats_part); dfree(c->d_bb_stats); dfree(c->d_bb_cdf); dfree(c->d_bb_cols); dfree(c->d_bb_cols_part); dfree(c->d_bb_cols_stat); dfree(c->d_bb_first); dfree(c->d_bbf_hier); dfree(c->d_bbf_table); dfree(c->d_bbf_q); dfree(c->d_bbf_strata); dfree(c->d_bbf_strata_top);
    dfree(c->d_gen_hier); dfree(c->d_gen_table); dfree(c->d_gen_q[0]); dfree(c->d_gen_q[1]); dfree(c->d_gen_ctrl);
    for (auto& ep : c->ev_used) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (auto& ep : c->ev_free) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

void* cpprob_hip_stream(cpprob_hip_ctx* c) { return c ? (void*)c->stream : nullptr; }

int cpprob_hip_sync(cpprob_hip_ctx* c)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int cpprob_hip_set_hmm(cpprob_hip_ctx* c, int32_t k, const double* h_means, const double* h_transition)
{
    if (!c || !h_means || !h_transition) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (k < 2 || k > 8) return fail(c, CPPROB_HIP_EINVAL, "the table model holds 2 .. 8 states");
    for (int s2 = 0; s2 < k; ++s2) {
        double tot = 0.0;
        for (int j = 0; j < k; ++j) { const double w = h_transition[(size_t)s2 * k + j]; if (!(w >= 0.0) || !std::isfinite(w)) return fail(c, CPPROB_HIP_EINVAL, "transition weights must be finite and >= 0"); tot += w; }
        if (!(tot > 0.0)) return fail(c, CPPROB_HIP_EINVAL, "a transition row without mass");
        if (!std::isfinite(h_means[s2])) return fail(c, CPPROB_HIP_EINVAL, "state means must be finite");
    }
    c->hk = k; c->hk_mean.assign(h_means, h_means + k); c->hk_trans.assign(h_transition, h_transition + (size_t)k * k);
    c->begun = false;                                   // (a run in flight keeps the tables it was begun with; the next begin takes these)
    return 0;
}

int cpprob_hip_infer_begin(cpprob_hip_ctx* c, const cpprob_hip_config* cfg, const double* h_obs, size_t n_obs)
{
    if (!c || !cfg || !h_obs) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (cfg->algorithm != CPPROB_HIP_ALG_SIS && cfg->algorithm != CPPROB_HIP_ALG_SMC)
        return fail(c, CPPROB_HIP_EUNSUPPORTED, "algorithm must be sis or smc (compile/csis/dryrun are outside this engine)");
    if (cfg->model < 0 || cfg->model > CPPROB_HIP_MODEL_HMM_TABLE) return fail(c, CPPROB_HIP_EINVAL, "unknown model id");
    if (cfg->model == CPPROB_HIP_MODEL_HMM_TABLE && c->hk < 2) return fail(c, CPPROB_HIP_ESTATE, "CPPROB_HIP_MODEL_HMM_TABLE: call cpprob_hip_set_hmm first");
    if (n_obs == 0) return fail(c, CPPROB_HIP_EINVAL, "the model has to receive the observed values (cpprob.hpp:182)");
    const bool gauss = cfg->model == CPPROB_HIP_MODEL_GAUSSIAN_UNKNOWN_MEAN || cfg->model == CPPROB_HIP_MODEL_GAUSSIAN_README;
    if (gauss && n_obs != 2) return fail(c, CPPROB_HIP_EINVAL, "gaussian_unknown_mean takes exactly two observes");
    if (cfg->model == CPPROB_HIP_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN && n_obs != 2) return fail(c, CPPROB_HIP_EINVAL, "gaussian_2d_unk_mean observes one vector of two values");
    if (cfg->n_particles == 0) return fail(c, CPPROB_HIP_EINVAL, "n_particles must be > 0");
    if (cfg->n_particles > (uint64_t)INT32_MAX - kTile) return fail(c, CPPROB_HIP_EINVAL, "n_particles per context must fit int32 ancestor indices");
    if (cfg->n_particles > (uint64_t)kMaxSlabs * kSlabTiles * kTile) return fail(c, CPPROB_HIP_EINVAL, "n_particles per context exceeds the two-level normalisation's capacity");
    if (cfg->n_global < cfg->n_particles || cfg->particle_offset + cfg->n_particles > cfg->n_global)
        return fail(c, CPPROB_HIP_EINVAL, "shard [particle_offset, particle_offset + n_particles) must lie inside [0, n_global)");
    if (cfg->algorithm == CPPROB_HIP_ALG_SMC) {
        if (cfg->resampler < 0 || cfg->resampler > CPPROB_HIP_RESAMPLE_MULTINOMIAL) return fail(c, CPPROB_HIP_EINVAL, "unknown resampler");
        if (!(cfg->ess_threshold >= 0.0)) return fail(c, CPPROB_HIP_EINVAL, "ess_threshold must be >= 0");
    }
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));

    if (cfg->resample_scope != CPPROB_HIP_SCOPE_GLOBAL && cfg->resample_scope != CPPROB_HIP_SCOPE_ISLAND && cfg->resample_scope != CPPROB_HIP_SCOPE_EXCHANGE)
        return fail(c, CPPROB_HIP_EINVAL, "unknown resample_scope");
    const bool exchange = cfg->resample_scope == CPPROB_HIP_SCOPE_EXCHANGE && cfg->algorithm == CPPROB_HIP_ALG_SMC;
    if (exchange && cfg->resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && ((cfg->flags & (CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL | CPPROB_HIP_FLAG_FLOATING_POINT_STEP)) || cfg->keep_history == 0))
        return fail(c, CPPROB_HIP_EUNSUPPORTED, "multinomial resampling in the exchange scope is the strata form on integer weights with histories kept: its thresholds come stratum by "
                                                 "stratum, so a rank's sources own one interval of outputs plus its share of the strata the ranks' boundaries cut (not the literal form, "
                                                 "not the floating-point step, not keep_history = 0)");
    c->cfg = *cfg;
    // a model with ONE observe statement has nothing to resample between: smc is sis (the components of its vector-valued
    // statements are rows of the particle store, not resampling points)
    if (cfg->model == CPPROB_HIP_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN) c->cfg.algorithm = CPPROB_HIP_ALG_SIS;
    const bool island = cfg->resample_scope == CPPROB_HIP_SCOPE_ISLAND;
    c->pop_n = island ? cfg->n_particles : cfg->n_global;
    c->T = model_T(cfg->model, n_obs);
    c->n_obs = (int)n_obs;
    c->begun = false; c->step_protocol = false; c->step_t = -1; c->force_fp = false; c->fixed_check_pending = false;
    // the read-out keeps kWaves * T * kStats accumulators in LDS (smooth_kernel)
    c->is_int = cfg->model == CPPROB_HIP_MODEL_HMM3 || cfg->model == CPPROB_HIP_MODEL_HMM_TABLE;
    c->K = cfg->model == CPPROB_HIP_MODEL_HMM_TABLE ? 8 : (c->is_int ? 3 : 2);
    if ((size_t)kWaves * (size_t)c->T * (size_t)std::max(c->K, 3) * sizeof(double) > 64 * 1024)
        return fail(c, CPPROB_HIP_EUNSUPPORTED, "too many predict hits per trace for the read-out kernel's LDS accumulators (T x statistics per hit <= 2048)");
    c->n = (int64_t)cfg->n_particles;
    c->nb = (int)((c->n + kTile - 1) / kTile);
    c->ld = (int64_t)c->nb * kTile;                 // padded to the tile: no ragged tails in any kernel
    c->smooth_grid = std::min(c->nb, 2048);
    c->walk_cap = std::min(c->nb, 8192);
    c->walk_grid = 0;
    host_model_params(c->mp, cfg->model);
    const bool smc = c->cfg.algorithm == CPPROB_HIP_ALG_SMC;
    const bool multinomial = smc && cfg->resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL;
    // keep_history = 0: filtering only -- two rows of values, no ancestors, predict hit t's statistics under generation t's own
    // weights.  SIS traces are their own lines (nothing to drop); a joint population's migration moves lineages (nothing to move).
    c->keep = cfg->keep_history != 0 || !smc;
    // (a shard of a joint population keeps no history in the exchange scope only: its migrants are then their current state alone)
    if (!c->keep && !exchange && cfg->resample_scope != CPPROB_HIP_SCOPE_ISLAND && cfg->n_global != cfg->n_particles)
        return fail(c, CPPROB_HIP_EUNSUPPORTED, "keep_history = 0 (filtering only) serves one population per context or a shard in the exchange scope: not a locally resampled shard");

    c->exchange = exchange;
    // exchange scope: room for immigrant lineages next to every row (grown on demand by cpprob_hip_exchange_commit)
    // default: what a well-mixed run needs -- a rank's offspring interval leaves its shard by O(sqrt(N)) outputs per resampling, so
    // ~sqrt(n_global) immigrants per step and T steps of them (measured 0.5 sqrt(N) per rank-step on BASELINE configs[3]) -- and at
    // least a sixteenth of the shard; a run that needs more reports overflow and is repeated with four times as much
    // (multinomial resampling, strata form: the two strata a rank's boundaries cut hand it up to ~a tile of immigrants a step whatever N is)
    const int64_t annex_strata = (exchange && cfg->resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL) ? (int64_t)c->T * (kTile + kTile / 2) : 0;
    const int64_t annex_mixed = (int64_t)(std::sqrt((double)cfg->n_global) * (double)c->T) / kTile * kTile + annex_strata;
    const int64_t annex_want = cfg->annex_kcols > 0 ? (int64_t)cfg->annex_kcols * 1024 : std::max<int64_t>(std::max<int64_t>(4 * kTile, c->ld / 16 / kTile * kTile), annex_mixed);
    const int64_t annex0 = exchange ? std::max<int64_t>(c->annex_cap, annex_want) : 0;
    c->ssz = 8;
    dispatch_model(c, [&](auto m) { c->ssz = sizeof(typename decltype(m)::store_t); c->grid_refs = decltype(m)::kWeightTable == 0; });
    // long traces in the exchange scope: skip rows shorten the extraction of migrating lineages (exchange.hpp)
    const bool want_skip = exchange && c->keep && !(cfg->flags & CPPROB_HIP_FLAG_NO_SKIP_ROWS) && c->T >= 3 * kSkipEvery;
    const bool realloc = (size_t)c->ld > c->cap_particles || c->T > c->cap_T || c->is_int != c->cap_int || (multinomial && !c->cap_multinomial) || !c->d_values ||
                         annex0 != c->annex_cap || c->keep != c->cap_keep || want_skip != (c->d_skip != nullptr);
    if (realloc) {
        free_run_buffers(c);
        c->annex_cap = annex0;
        const size_t ld = (size_t)c->ld, T = (size_t)c->T, rs = ld + (size_t)annex0;
        const size_t vsz = c->ssz;
        HIP_TRY(c, hipMalloc(&c->d_logw[0], ld * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_logw[1], ld * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_wrel[0], ld * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_wrel[1], ld * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_bf, (size_t)c->nb * sizeof(double)));
        const size_t rows = c->keep ? T : 2;
        HIP_TRY(c, hipMalloc(&c->d_values, rows * rs * vsz));
        HIP_TRY(c, hipMemsetAsync(c->d_values, 0, rows * rs * vsz, c->stream));
        // (ancestor and skip rows start as zeros -- a valid slot: a run that overflowed its transport still walks its lineages once,
        //  through annex columns nobody committed, before the host sees the flag and repeats it)
        if (c->keep) { HIP_TRY(c, hipMalloc(&c->d_anc, T * rs * sizeof(int32_t))); HIP_TRY(c, hipMemsetAsync(c->d_anc, 0, T * rs * sizeof(int32_t), c->stream)); }
        else { HIP_TRY(c, hipMalloc(&c->d_fpart, T * (8 + 2) * (size_t)c->smooth_grid * sizeof(double))); HIP_TRY(c, hipMalloc(&c->d_filter_w, T * sizeof(double))); }
        if (want_skip) {
            HIP_TRY(c, hipMalloc(&c->d_skip, (T / kSkipEvery + 1) * rs * sizeof(int32_t)));
            HIP_TRY(c, hipMemsetAsync(c->d_skip, 0, (T / kSkipEvery + 1) * rs * sizeof(int32_t), c->stream));
        }
        HIP_TRY(c, hipMalloc(&c->d_obound, (1024 + 2) * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_part[0], (size_t)part_stride(c->nb) * 3 * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_part[1], (size_t)part_stride(c->nb) * 3 * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_bc, ((size_t)c->nb + 1) * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_gpart, (size_t)3 * kMaxSlabs * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_stile, (size_t)kMaxReadoutCols * c->nb * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_gstat, (size_t)kMaxReadoutCols * kMaxSlabs * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_ess, T * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_lz_trace, T * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_resampled, T * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(&c->d_stats_part, (size_t)c->walk_cap * T * 8 * sizeof(double)));
        HIP_TRY(c, hipMalloc(&c->d_stats, T * 8 * sizeof(double)));
        {
            // 64-ary hierarchy of per-tile state counts, three rotating copies (step_counts.hpp); levels >= 1 one line per entry
            size_t per_copy = 0; int lev = 0;
            for (size_t e = (size_t)c->nb;; e = (e + 63) / 64, ++lev) { per_copy += e * (lev == 0 ? 1 : kHierStride); if (e <= 64) break; }
            per_copy += 2 * (size_t)c->nb;                          // fixed-point form: the tiles' Q and M arrays
            c->hier_entries = 3 * per_copy;
            HIP_TRY(c, hipMalloc(&c->d_q[0], ld * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_q[1], ld * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_hier, c->hier_entries * sizeof(uint64_t)));
            HIP_TRY(c, hipMalloc(&c->d_annex_base, (T + 1) * sizeof(int64_t)));
            HIP_TRY(c, hipMemsetAsync(c->d_annex_base, 0, (T + 1) * sizeof(int64_t), c->stream));
        }
        if (multinomial) {
            HIP_TRY(c, hipMalloc(&c->d_cdf, ld * sizeof(double)));
            HIP_TRY(c, hipMalloc(&c->d_anc_pre, ld * sizeof(int32_t)));
        }
        c->cap_particles = ld; c->cap_T = c->T; c->cap_int = c->is_int; c->cap_multinomial = multinomial; c->cap_keep = c->keep;
    }
    {
        std::memset(&c->hier, 0, sizeof c->hier);
        size_t per_copy = 0; int nl = 0;
        size_t off[kHierMaxLevels] = {0, 0, 0};
        for (size_t e = (size_t)c->nb;; e = (e + 63) / 64) {
            if (nl >= kHierMaxLevels) break;                       // (more than 64^3 tiles: the prefix-count form is not eligible)
            off[nl] = per_copy; c->hier.n_ent[nl] = (int)e; per_copy += e * (nl == 0 ? 1 : kHierStride); ++nl;
            if (e <= 64) break;
        }
        c->hier_q0_off = per_copy; c->hier_m0_off = per_copy + (size_t)c->nb; per_copy += 2 * (size_t)c->nb;
        c->hier.n_lev = nl; c->hier_per_copy = per_copy;
        for (int k = 0; k < 3; ++k)
            for (int l = 0; l < nl; ++l) c->hier.lvl[k][l] = c->d_hier + (size_t)k * per_copy + off[l];
        if (3 * per_copy > c->hier_entries) return fail(c, CPPROB_HIP_EDEVICE, "count hierarchy exceeds its allocation");
        if (!c->d_hier_table) HIP_TRY(c, hipMalloc(&c->d_hier_table, sizeof(HierTable)));
        HIP_TRY(c, hipMemcpyAsync(c->d_hier_table, &c->hier, sizeof(HierTable), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_hier, 0, c->hier_entries * sizeof(uint64_t), c->stream));     // (the layout may have changed)
        c->hier_phase = 0; c->hier_run_open = false;
    }
    if (multinomial) {
        // the strata of every step of a run ([T - 1][2^k + 1]; a shard of a joint population holds the POPULATION's) and the level-6 totals
        const size_t words = (size_t)c->T * (((size_t)1 << strata_k_of(c)) + 1);
        if (words > c->strata_cap_words || c->T > c->strata_cap_T || !c->d_strata || !c->d_strata_top) {
            dfree(c->d_strata); dfree(c->d_strata_top);
            HIP_TRY(c, hipMalloc(&c->d_strata, words * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_strata_top, (size_t)2 * c->T * 64 * sizeof(uint32_t)));
            c->strata_cap_words = words; c->strata_cap_T = c->T;
        }
        if (exchange && !c->d_cut_tab) {
            HIP_TRY(c, hipMalloc(&c->d_cut_tab, (size_t)kCutSlots * kCutRow * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_cut_head, (size_t)kCutSlots * sizeof(CutHead)));
            HIP_TRY(c, hipMalloc(&c->d_cut_srccnt, (size_t)kCutSlots * kCutSlots * sizeof(uint32_t)));
            HIP_TRY(c, hipMemsetAsync(c->d_cut_tab, 0, (size_t)kCutSlots * kCutRow * sizeof(uint32_t), c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_cut_head, 0, (size_t)kCutSlots * sizeof(CutHead), c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_cut_srccnt, 0, (size_t)kCutSlots * kCutSlots * sizeof(uint32_t), c->stream));
        }
    }
    c->rs = c->ld + c->annex_cap;
    c->annex_used = 0; c->plan.t = -1; c->x_plan_t = -1; c->x_all_totals = nullptr;
    if (!c->d_xplan) { HIP_TRY(c, hipMalloc(&c->d_xplan, sizeof(ExchangePlan))); HIP_TRY(c, hipMemsetAsync(c->d_xplan, 0, sizeof(ExchangePlan), c->stream)); }
    dfree(c->d_obs);
    HIP_TRY(c, hipMalloc(&c->d_obs, n_obs * sizeof(double)));
    HIP_TRY(c, hipMemcpyAsync(c->d_obs, h_obs, n_obs * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_resampled, 0, (size_t)c->T * sizeof(int32_t), c->stream));
    // (the trace-word read-out's counters: allocated HERE, not by the first run that needs them -- a context made for one run paid two
    //  hipMalloc and two clears inside that run's clock: ~40 us of a 150-us run through cpprob::inference)
    if (cfg->algorithm == CPPROB_HIP_ALG_SMC && cfg->model == CPPROB_HIP_MODEL_HMM3 && c->T <= kTraceMaxT && !c->d_trace_cnt) {
        HIP_TRY(c, hipMalloc(&c->d_trace_cnt, kTraceCounterWords * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_trace_arrive, sizeof(unsigned long long)));
        HIP_TRY(c, hipMemsetAsync(c->d_trace_cnt, 0, kTraceCounterWords * sizeof(uint32_t), c->stream));
        HIP_TRY(c, hipMemsetAsync(c->d_trace_arrive, 0, sizeof(unsigned long long), c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(&c->d_ctrl->lz_trace, &c->d_lz_trace, sizeof(double*), hipMemcpyHostToDevice, c->stream));
    if (multinomial && c->d_strata_top) {                      // (its layout follows T; a run abandoned between its two launches leaves totals behind)
        HIP_TRY(c, hipMemsetAsync(c->d_strata_top, 0, (size_t)2 * c->T * 64 * sizeof(uint32_t), c->stream));
        c->strata_phase = 0;
    }
    {
        // Gaussian models: the log-weight as a quadratic in the prior's standard-normal variate (extended precision on the host),
        // and the grid point above its maximum
        c->sis_bounded_ok = false;
        const bool g1 = gauss, g2 = cfg->model == CPPROB_HIP_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN;
        bool finite = true;
        for (size_t i = 0; i < n_obs; ++i) finite = finite && std::isfinite(h_obs[i]);
        if ((g1 || g2) && finite) {
            long double lmax = 0.0L;
            const long double sg = c->mp.sigma, ln = c->mp.log_norm_lik;
            if (g1) {
                const long double b = (long double)c->mp.sigma0 / sg, a1 = ((long double)h_obs[0] - c->mp.mu0) / sg, a2 = ((long double)h_obs[1] - c->mp.mu0) / sg;
                const long double A = -b * b, B = b * (a1 + a2), C = -0.5L * (a1 * a1 + a2 * a2) - ln;
                c->mp.quad[0][0] = (double)A; c->mp.quad[0][1] = (double)B; c->mp.quad[0][2] = (double)C;
                lmax = C - B * B / (4.0L * A);
            } else {
                for (int d = 0; d < 2; ++d) {
                    const long double b = (long double)c->mp.nd_sigma[d] / sg, a = ((long double)h_obs[d] - c->mp.nd_mean[d]) / sg;
                    const long double A = -0.5L * b * b, B = a * b, C = -0.5L * a * a - 0.5L * ln;
                    c->mp.quad[d][0] = (double)A; c->mp.quad[d][1] = (double)B; c->mp.quad[d][2] = (double)C;
                    lmax += C - B * B / (4.0L * A);
                }
            }
            c->mp.lw_ref = std::ceil((double)lmax * kInvLn2) * kLn2;
            c->sis_bounded_ok = c->T <= kMaxReadoutT;
            if (!c->d_wpart) HIP_TRY(c, hipMalloc(&c->d_wpart, (size_t)(2 + kMaxReadoutT * 2) * kSisBoundedMaxGrid * sizeof(double)));
        }
    }
    dfree(c->d_ll_tab);
    c->mp.ll_tab = nullptr; c->mp.e_tab = nullptr;
    if (cfg->model == CPPROB_HIP_MODEL_HMM3) {
        // log N(y_t; state_mean[s], 1): three values per step, computed once with the same functor
        // the reference applies per particle (utils_normal_distribution.hpp:20-45)
        std::vector<double> tab((size_t)c->T * 3);
        for (int t = 0; t < c->T; ++t)
            for (int s2 = 0; s2 < 3; ++s2) tab[(size_t)t * 3 + s2] = normal_logpdf(h_obs[t], c->mp.hmm_mean[s2], 1.0);
        c->h_ll_tab = tab;
        HIP_TRY(c, hipMalloc(&c->d_ll_tab, tab.size() * sizeof(double)));
        HIP_TRY(c, hipMemcpy(c->d_ll_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        c->mp.ll_tab = c->d_ll_tab;
        // linear weights of the three values against their maximum: exp(ll - max), and the max itself
        std::vector<double> et((size_t)c->T * 4);
        for (int t = 0; t < c->T; ++t) {
            const double* l = &tab[(size_t)t * 3];
            const double mx = std::max(l[0], std::max(l[1], l[2]));
            for (int s2 = 0; s2 < 3; ++s2) et[(size_t)t * 4 + s2] = std::exp(l[s2] - mx);
            et[(size_t)t * 4 + 3] = mx;
        }
        c->h_e_tab = et;
        dfree(c->d_e_tab);
        HIP_TRY(c, hipMalloc(&c->d_e_tab, et.size() * sizeof(double)));
        HIP_TRY(c, hipMemcpy(c->d_e_tab, et.data(), et.size() * sizeof(double), hipMemcpyHostToDevice));
        c->mp.e_tab = c->d_e_tab;
    }
    c->mp.hk = 0; c->mp.hk_thr = nullptr; c->mp.hk_ll = nullptr;
    if (cfg->model == CPPROB_HIP_MODEL_HMM_TABLE) {
        // thresholds of the rows' inverse CDFs (u >= c  <=>  word >= ceil(c 2^32), as for the three-state model) and the
        // emission log-densities of every step
        const int k = c->hk;
        std::vector<uint64_t> thr((size_t)k * 8, ~0ull);
        for (int s2 = 0; s2 < k; ++s2) {
            double tot = 0.0, acc = 0.0;
            for (int j = 0; j < k; ++j) tot += c->hk_trans[(size_t)s2 * k + j];
            for (int j = 0; j + 1 < k; ++j) { acc += c->hk_trans[(size_t)s2 * k + j]; thr[(size_t)s2 * 8 + j] = (uint64_t)std::ceil((acc / tot) * 4294967296.0); }
        }
        std::vector<double> ll((size_t)c->T * 8, -INFINITY);
        for (int t = 0; t < c->T; ++t)
            for (int s2 = 0; s2 < k; ++s2) ll[(size_t)t * 8 + s2] = normal_logpdf(h_obs[t], c->hk_mean[(size_t)s2], 1.0);
        dfree(c->d_hk_thr); dfree(c->d_hk_ll);
        HIP_TRY(c, hipMalloc(&c->d_hk_thr, thr.size() * sizeof(uint64_t)));
        HIP_TRY(c, hipMalloc(&c->d_hk_ll, ll.size() * sizeof(double)));
        HIP_TRY(c, hipMemcpy(c->d_hk_thr, thr.data(), thr.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_hk_ll, ll.data(), ll.size() * sizeof(double), hipMemcpyHostToDevice));
        c->mp.hk = k; c->mp.hk_thr = c->d_hk_thr; c->mp.hk_ll = c->d_hk_ll;
        c->h_ll_tab = ll;                                   // (host copy: the bounds below)
    }
    {
        // fixed-point form: the upper bound of each step's incremental log-weight, with the very expressions the kernels evaluate
        c->h_bound.assign((size_t)c->T, 0.0);
        for (int t = 0; t < c->T; ++t) {
            if (cfg->model == CPPROB_HIP_MODEL_HMM3) c->h_bound[(size_t)t] = c->h_e_tab[(size_t)t * 4 + 3];                       // the largest table value
            else if (cfg->model == CPPROB_HIP_MODEL_LINEAR_GAUSSIAN_1D) c->h_bound[(size_t)t] = normal_logpdf_hoisted(h_obs[t], h_obs[t], 1.0, c->mp.log_norm_unit);   // the emission's density at its mode
            else if (cfg->model == CPPROB_HIP_MODEL_HMM_TABLE) { double b = -INFINITY; for (int s2 = 0; s2 < c->hk; ++s2) b = std::max(b, c->h_ll_tab[(size_t)t * 8 + s2]); c->h_bound[(size_t)t] = b; }
        }
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->begun = true; c->ran = false;
    return 0;
}

int cpprob_hip_infer_run(cpprob_hip_ctx* c, uint64_t run_index)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->begun) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_infer_begin has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->cfg.resample_scope != CPPROB_HIP_SCOPE_ISLAND && c->cfg.n_global != c->cfg.n_particles)
        return fail(c, CPPROB_HIP_ESTATE, "this context holds one shard of a joint population: drive it with cpprob_hip_smc_step_begin/_end/_finish");
    c->run_seed = c->cfg.seed + run_index;
    c->cur = 0; c->cur_part = 0;
    c->sharded = false;
    if (!c->force_fp) c->n_requantised = 0;
    c->final_from_counts = false; c->final_from_fixed = false;
    bool readout_done = false, sis_bounded = false;
    (void)sis_bounded;
    if (c->cfg.algorithm == CPPROB_HIP_ALG_SIS) {
        dispatch_model(c, [&](auto m) {
            using M = decltype(m);
            if (launch_sis_bounded<M>(c)) { readout_done = true; sis_bounded = true; return; }
            readout_done = sis_readout_fused<M>(c);
            launch_sis<M>(c, readout_done);
            if (readout_done) launch_sis_readout<M>(c);
        });
        if (!readout_done) launch_scan(c, c->T - 1, 0, nullptr, 1, 0);
    } else {
        c->step_protocol = false;
        c->counts_mode = false; c->fixed_mode = false;
        dispatch_model(c, [&](auto m) { c->counts_mode = counts_eligible<decltype(m)>(c); c->fixed_mode = fixed_eligible<decltype(m)>(c); });
        // a population of its own, short traces of a few states: the particles carry their traces (trace_words.hpp)
        c->trace_mode = c->counts_mode && c->keep && !c->exchange && c->T <= kTraceMaxT && c->cfg.model == CPPROB_HIP_MODEL_HMM3 &&
                        !(c->cfg.flags & CPPROB_HIP_FLAG_WALK_READOUT);
        if (c->trace_mode && !c->d_trace_cnt) {
            HIP_TRY(c, hipMalloc(&c->d_trace_cnt, kTraceCounterWords * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_trace_arrive, sizeof(unsigned long long)));
            HIP_TRY(c, hipMemsetAsync(c->d_trace_cnt, 0, kTraceCounterWords * sizeof(uint32_t), c->stream));
            HIP_TRY(c, hipMemsetAsync(c->d_trace_arrive, 0, sizeof(unsigned long long), c->stream));
        }
        const bool fused = step_is_fused(c);
        if (c->fixed_mode) {
            c->strata_pending = c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && !(c->cfg.flags & CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL);
            launch_strata(c);
            ProfScope group(c, 0, c->T);
            c->profile_suspended = true;
            for (int t = 0; t < c->T; ++t) dispatch_model(c, [&](auto m) { launch_step_fixed<decltype(m)>(c, t, nullptr, 1, 0); });
            c->profile_suspended = false;
            // (no normalisation launch at any size: the read-out works from the final generation's masses)
        } else if (c->counts_mode) {
            c->strata_pending = c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL;
            launch_strata(c);
            {
                ProfScope group(c, 0, c->T);
                c->profile_suspended = true;
                for (int t = 0; t < c->T; ++t) dispatch_model(c, [&](auto m) { launch_step_counts<decltype(m)>(c, t, nullptr, 1, 0); });
                c->profile_suspended = false;
            }
            // (no normalisation launch: the read-out works from the final generation's counts)
        } else if (fused) {
            // the T step kernels run back to back: one event pair around the group
            {
                ProfScope group(c, 0, c->T);
                c->profile_suspended = true;
                for (int t = 0; t < c->T; ++t) dispatch_model(c, [&](auto m) { launch_step<decltype(m)>(c, t); });
                c->profile_suspended = false;
            }
            launch_scan(c, c->T - 1, 0, nullptr, 1, 0);                          // only the final generation needs the standalone pass
        } else {
            for (int t = 0; t < c->T; ++t) {
                dispatch_model(c, [&](auto m) { launch_step<decltype(m)>(c, t); });
                launch_scan(c, t, 0, nullptr, 1, 0);
            }
        }
    }
    if (!c->keep) {
        // filtering only: every step left its own statistics (count form: from the generation's totals; floating-point form:
        // filter_partials_kernel); what remains is the final generation's bookkeeping and the normalisation
        if (c->final_from_counts) {
            CountsFinal f{};
            counts_final_view(c, f, true);
            c->final_bookkeep_pending = false;
            hipLaunchKernelGGL(counts_filter_final_kernel, dim3(1), dim3(kWave), 0, c->stream, f);
        } else {
            if (c->final_from_fixed) {
                FixedFinal ff{};
                fixed_final_view(c, ff, true);
                c->final_bookkeep_pending = false;
                hipLaunchKernelGGL(fixed_filter_final_kernel, dim3(1), dim3(kWave), 0, c->stream, ff);
            }
            hipLaunchKernelGGL(filter_finalize_kernel, dim3(c->T), dim3(kThreads), 0, c->stream, (const double*)c->d_fpart, c->smooth_grid, c->K, c->is_int ? 1 : 0, c->d_stats, 1, (double*)nullptr);
        }
    } else if (!readout_done) dispatch_model(c, [&](auto m) { launch_smooth<decltype(m)>(c, false); });
    HIP_TRY(c, hipGetLastError());
    c->ran = true;
    c->fixed_check_pending = c->fixed_mode && c->cfg.algorithm == CPPROB_HIP_ALG_SMC; c->last_was_infer_run = true; c->last_run_index = run_index;
    return 0;
}

}  // extern "C"
namespace {
// A generation whose heaviest particle sat more than kFixGapLimit nats below its reference kept too few of its 32 bits (an observation
// far from EVERY particle).  Repair, in integers: the offending generation g is weighed again against its EXACT maximum -- its
// log-weights recomputed from the particle store (fixed_relogw_kernel), maximum and masses by the two order-free passes of
// bookkeep_fixed.hpp -- the books are rewound to where they stood before g (StepCtrl::lz_trace), and the steps behind g run again;
// repeated while a later generation trips (each round starts later).  The CPU restatement states the same rule (orc_smc_impl), so a
// repaired run still equals it index for index, and no floating-point CDF enters.
template <class Model>
int repair_fixed_generation(cpprob_hip_ctx* c, int g)
{
    if constexpr (std::is_same<Model, ModelLinearGaussian1D>::value || std::is_same<Model, ModelHmm3>::value || std::is_same<Model, ModelHmmK>::value) {
        std::vector<int32_t> res((size_t)c->T);
        HIP_TRY(c, hipMemcpyAsync(res.data(), c->d_resampled, res.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        int s0 = 0;
        for (int s2 = g - 1; s2 >= 0; --s2) if (res[(size_t)s2]) { s0 = s2 + 1; break; }
        c->cur = (g + 1) & 1;                                        // (where step g left its weights: the buffers alternate step by step)
        double* lw = c->d_logw[c->cur];
        uint32_t* q = c->d_q[c->cur];
        hipLaunchKernelGGL(fixed_relogw_kernel<Model>, dim3((unsigned)((c->ld + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, c->mp, (const double*)c->d_obs,
                           static_cast<const typename Model::store_t*>(c->d_values), c->rs, s0, g, c->n, c->ld, lw);
        // the hierarchy: every copy clean, generation g's masses in the copy step g + 1 reads
        HIP_TRY(c, hipMemsetAsync(c->d_hier, 0, c->hier_entries * sizeof(uint64_t), c->stream));
        const int ka = (g + 1 + c->hier_phase_run) % 3, kb = (ka + 1) % 3;
        FHier f{};
        fhier_view(c, ka, f);
        f.h.to_next = 0; f.h.to_clear = (int64_t)(kb - ka) * (int64_t)c->hier_per_copy;
        hipLaunchKernelGGL(bbf_max_kernel, dim3(c->nb), dim3(kThreads), 0, c->stream, (const double*)lw, c->n, f);
        hipLaunchKernelGGL(bbf_quantize_kernel, dim3(c->nb), dim3(kThreads), 0, c->stream, (const double*)lw, c->n, f, q);
        hipLaunchKernelGGL(fixed_repair_ctrl_kernel, dim3(1), dim3(kWave), 0, c->stream, c->d_ctrl, f, g, (const int32_t*)c->d_resampled);
        for (int t = g + 1; t < c->T; ++t) launch_step_fixed<Model>(c, t, nullptr, 1, 0);
        if (g + 1 == c->T) { c->final_from_fixed = true; c->final_copy = ka; c->final_bookkeep_pending = true; }
        launch_smooth<Model>(c, false);
        HIP_TRY(c, hipGetLastError());
        return 0;
    }
    return fail(c, CPPROB_HIP_ESTATE, "the fixed-point form serves the state-space models");
}

// Before a run's results leave the library: did the fixed-point weights keep their bits?  If not the offending generations are
// repaired as above; where that is not possible (a shard of a joint population, a filtering-only run, a generation without mass) the
// run is repeated in the floating-point form (runs the caller drives step by step cannot be repeated here: CPPROB_HIP_EPRECISION).
// Small results reach the host WITHOUT copy commands and without waiting for the stream: one short launch stores them into pinned
// (device-visible, coherent) host memory, then a sequence number behind them; the host polls that word.  (Measured on the way here,
// per read-back of a few hundred bytes: a copy into pageable memory ~35 us of host round trip, an asynchronous copy into pinned memory
// ~20 us each -- three of them closed every run -- and hipStreamSynchronize's own wake-up behind the last kernel.)
constexpr size_t kPinHeader = 64;                                       // the block's first bytes: the sequence word
struct PinSeg { const uint32_t* src; uint32_t* dst; uint32_t words; uint32_t pad; };
struct PinSegs { PinSeg s[4]; int n; };
__global__ __launch_bounds__(kThreads) void pin_pack_kernel(PinSegs a, unsigned long long* flag, unsigned long long seq)
{
    for (int k = 0; k < a.n; ++k)
        for (uint32_t i = threadIdx.x; i < a.s[k].words; i += kThreads) a.s[k].dst[i] = a.s[k].src[i];
    __threadfence_system();                                             // (every thread's stores are on their way before the flag is)
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static int pin_reserve(cpprob_hip_ctx* c, size_t bytes)                 // bytes of data (behind the header)
{
    bytes += kPinHeader;
    if (bytes <= c->h_pin_cap) return 0;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_pin) { (void)hipHostFree(c->h_pin); c->h_pin = nullptr; c->h_pin_cap = 0; }
    const size_t want = std::max<size_t>(bytes + bytes / 2, (size_t)1 << 16);
    // (coherent, i.e. fine-grained: the host sees the sequence word while the stream is still busy; a runtime that refuses the flag
    //  combination still gets a correct block -- pin_wait falls back to the stream's completion)
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_pin), want, hipHostMallocCoherent | hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
        (void)hipGetLastError();
        c->h_pin = nullptr;
        HIP_TRY(c, hipHostMalloc(reinterpret_cast<void**>(&c->h_pin), want, hipHostMallocPortable | hipHostMallocMapped));
    }
    std::memset(c->h_pin, 0, kPinHeader);
    c->h_pin_cap = want; c->pin_seq = 0;
    return 0;
}
static char* pin_data(cpprob_hip_ctx* c) { return c->h_pin + kPinHeader; }
struct PinPack {
    PinSegs segs{};
    void add(const void* d_src, void* h_dst, size_t bytes)             // (device memory -> the context's pinned block; 4-byte words)
    {
        if (bytes == 0) return;
        PinSeg& g = segs.s[segs.n++];
        g.src = static_cast<const uint32_t*>(d_src); g.dst = static_cast<uint32_t*>(h_dst); g.words = (uint32_t)((bytes + 3) / 4); g.pad = 0;
    }
    // the launch; `wait` is what the host calls when it wants the bytes
    int launch(cpprob_hip_ctx* c)
    {
        c->pin_seq += 1;
        hipLaunchKernelGGL(pin_pack_kernel, dim3(1), dim3(kThreads), 0, c->stream, segs, reinterpret_cast<unsigned long long*>(c->h_pin), (unsigned long long)c->pin_seq);
        HIP_TRY(c, hipGetLastError());
        return 0;
    }
};
// the host's side: poll the sequence word (the launch is short and last in the stream); a look at the stream every few thousand polls
// catches a failed launch, and a stream that finished without the word showing falls back to its synchronisation
static int pin_wait(cpprob_hip_ctx* c)
{
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(c->h_pin);
    const unsigned long long want = (unsigned long long)c->pin_seq;
    for (unsigned k = 1;; ++k) {
        if (*flag == want) break;
        if ((k & 0x3fffu) == 0) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) { if (*flag != want) HIP_TRY(c, hipStreamSynchronize(c->stream)); break; }
            if (q != hipErrorNotReady) { HIP_TRY(c, q); }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return 0;
}

int settle_fixed(cpprob_hip_ctx* c)
{
    if (!c->fixed_check_pending) return 0;
    c->fixed_check_pending = false;
    StepCtrl h{};
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h.fix_gap <= kFixGapLimit) return 0;
    // (the literal multinomial form keeps lane prefixes beside the weights that a requantised generation would leave stale: it repeats the run)
    const bool literal = c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && (c->cfg.flags & CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL);
    if (c->last_was_infer_run && c->keep && !c->exchange && c->d_lz_trace && !literal && !(c->cfg.flags & CPPROB_HIP_FLAG_REPEAT_IN_FLOATING_POINT)) {
        int g_prev = -1;
        for (int round = 0; round <= c->T; ++round) {
            const int g = h.first_bad;
            if (g < 0 || g >= c->T || g <= g_prev) break;                // (no progress: a generation without any mass)
            int rc = 0;
            dispatch_model(c, [&](auto m) { rc = repair_fixed_generation<decltype(m)>(c, g); });
            if (rc) return rc;
            c->n_requantised += 1;
            g_prev = g;
            HIP_TRY(c, hipMemcpyAsync(&h, c->d_ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (h.fix_gap <= kFixGapLimit) return 0;
        }
    }
    if (!c->last_was_infer_run)
        return fail(c, CPPROB_HIP_EPRECISION, "some generation's heaviest particle sat more than 6 nats below the fixed-point reference (an observation far from every particle): "
                                               "repair it in the run (cpprob_hip_smc_repair_begin / _end on every rank, from the first offending generation: StepCtrl::first_bad "
                                               "through cpprob_hip_smc_first_bad_generation) or repeat the run with CPPROB_HIP_FLAG_FLOATING_POINT_STEP");
    c->force_fp = true;
    return cpprob_hip_infer_run(c, c->last_run_index);
}
}  // namespace
extern "C" {

static int ensure_shard_trace(cpprob_hip_ctx* c, bool& words);

int cpprob_hip_smc_step_begin(cpprob_hip_ctx* c, int32_t t, uint64_t run_index, double* d_local_totals)
{
    if (!c || !d_local_totals) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->begun) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_infer_begin has not been called");
    if (t < 0 || t >= c->T) return fail(c, CPPROB_HIP_EINVAL, "step out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    const bool sis = c->cfg.algorithm == CPPROB_HIP_ALG_SIS;
    if (sis && t != c->T - 1) return fail(c, CPPROB_HIP_EINVAL, "SIS shards run in one launch: call step_begin(T-1) only");
    if (t == 0 || sis) {
        c->strata_pending = !sis && c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && !(c->cfg.flags & CPPROB_HIP_FLAG_MULTINOMIAL_LITERAL);
        c->run_seed = c->cfg.seed + run_index; c->cur = 0; c->cur_part = 0; c->ran = false; c->annex_used = 0; c->plan.t = -1; c->x_plan_t = -1;
        c->n_requantised = 0; c->annex_used_before.clear();
        c->final_from_counts = false; c->final_from_fixed = false;
        c->counts_mode = false; c->fixed_mode = false;
        // the integer forms serve exact joint resampling (exchange scope) and a population held by this context alone; a shard that
        // resamples locally (global scope, several ranks) keeps the floating-point form and its mass-share bookkeeping
        if (!sis && (c->exchange || c->cfg.n_global == c->cfg.n_particles))
            dispatch_model(c, [&](auto m) { c->counts_mode = counts_eligible<decltype(m)>(c); c->fixed_mode = fixed_eligible<decltype(m)>(c); });
    }
    if ((t == 0 || sis) && c->exchange && !sis && c->cfg.resampler != CPPROB_HIP_RESAMPLE_SYSTEMATIC && !c->counts_mode && !c->fixed_mode)
        return fail(c, CPPROB_HIP_EUNSUPPORTED, "stratified / multinomial resampling in the exchange scope run on the integer forms of the step (the floating-point form plans the ranks' "
                                                 "offspring intervals from the ONE systematic offset): not with CPPROB_HIP_FLAG_FLOATING_POINT_STEP, beyond 2^28 particles, or for models "
                                                 "without a fixed-point form");
    if (c->exchange && t > 0 && c->x_plan_t != t - 1)
        return fail(c, CPPROB_HIP_ESTATE, "exchange scope: the exchange of the previous step (plan / pack / commit) must run before the next step_begin");
    c->step_protocol = true; c->step_t = t; c->trace_mode = false;
    if (t == 0 || sis) {
        c->trace_shard_run = c->trace_shard && c->x_remote && c->counts_mode && c->keep && !c->x_peers.empty();
        if (c->exchange && c->x_fixed && c->x_peers.empty() && c->counts_mode && c->keep) {
            // (a shard with nobody to exchange with -- a group of one -- needs nobody's consent)
            bool words = false;
            if (int rc = ensure_shard_trace(c, words)) return rc;
            c->trace_shard_run = words;
        }
    }
    c->totals_out = d_local_totals;
    c->x_gather_done = c->x_gather_on && (c->counts_mode || c->fixed_mode);
    if (c->counts_mode) {
        // prefix-count form: the step consumes the all-gathered counts of generation t-1 itself; what leaves is this shard's
        // {n_0, n_1, particles} (exact doubles), after the last step too: the read-out works from the final generation's counts
        if (t > 0 && !c->x_all_totals) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_smc_step_end(t-1) has not run");
        dispatch_model(c, [&](auto m) { launch_step_counts<decltype(m)>(c, t, t > 0 ? c->x_all_totals : nullptr, c->x_world, c->x_rank); });
        {
            Hier h{};
            const int kn = (t + 1 + c->hier_phase_run) % 3;
            hier_view(c, kn, h);
            if (c->x_gather_on) { TotalsGather d = c->x_gather; d.parity = (t - c->x_gather_tshift) & 1; d.seq = (c->x_gather_serial << 32) | (unsigned long long)(uint32_t)(t - c->x_gather_tshift + 1);
                                  hipLaunchKernelGGL(counts_totals_gather_kernel, dim3(1), dim3(kWave), 0, c->stream, h, (double)c->n, d_local_totals, d); }
            else hipLaunchKernelGGL(counts_totals_kernel, dim3(1), dim3(kWave), 0, c->stream, h, (double)c->n, d_local_totals);
        }
    } else if (c->fixed_mode) {
        // fixed-point form: likewise; what leaves is this shard's {mass, squares, key of the largest log-weight}: 24 bytes
        if (t > 0 && !c->x_all_totals) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_smc_step_end(t-1) has not run");
        dispatch_model(c, [&](auto m) { launch_step_fixed<decltype(m)>(c, t, t > 0 ? c->x_all_totals : nullptr, c->x_world, c->x_rank); });
        {
            FHier f{};
            const int kn = (t + 1 + c->hier_phase_run) % 3;
            fhier_view(c, kn, f);
            if (c->x_gather_on) { TotalsGather d = c->x_gather; d.parity = (t - c->x_gather_tshift) & 1; d.seq = (c->x_gather_serial << 32) | (unsigned long long)(uint32_t)(t - c->x_gather_tshift + 1);
                                  hipLaunchKernelGGL(fixed_totals_gather_kernel, dim3(1), dim3(kWave), 0, c->stream, f, reinterpret_cast<uint64_t*>(d_local_totals), d); }
            else hipLaunchKernelGGL(fixed_totals_kernel, dim3(1), dim3(kWave), 0, c->stream, f, reinterpret_cast<uint64_t*>(d_local_totals));
        }
    } else {
        if (sis) dispatch_model(c, [&](auto m) { launch_sis<decltype(m)>(c, false); });
        else dispatch_model(c, [&](auto m) { launch_step<decltype(m)>(c, t); });
        launch_scan(c, t, 1, nullptr, 1, 0);
    }
    if (c->d_skip && !c->x_remote && t >= kSkipEvery && (t % kSkipEvery) == 0 && !sis) {
        // every eighth step: where each local slot's lineage sat eight generations ago
        hipLaunchKernelGGL(skip_rows_kernel, dim3((unsigned)((c->n + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, (const int32_t*)c->d_anc, c->rs, c->n,
                           (const int32_t*)c->d_resampled, t, c->d_skip + (size_t)(t / kSkipEvery) * c->rs);
    }
    HIP_TRY(c, hipGetLastError());
    c->sharded = true;
    return 0;
}

int cpprob_hip_smc_step_end(cpprob_hip_ctx* c, int32_t t, const double* d_all_totals, int32_t world, int32_t rank)
{
    if (!c || !d_all_totals) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->begun || !c->step_protocol || t != c->step_t) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_smc_step_end(t) follows cpprob_hip_smc_step_begin(t)");
    if (world < 1 || world > 1024 || rank < 0 || rank >= world) return fail(c, CPPROB_HIP_EINVAL, "bad world/rank (1 <= world <= 1024)");
    if (c->exchange && world > kMaxWorld) return fail(c, CPPROB_HIP_EUNSUPPORTED, "the exchange scope plans on one wavefront (world + 1 bounds, one per lane): world <= 63");
    HIP_TRY(c, hipSetDevice(c->device));
    c->x_all_totals = d_all_totals; c->x_world = world; c->x_rank = rank;
    if (c->fixed_mode) {
        if (t + 1 == c->T) {
            if (world > kWave) return fail(c, CPPROB_HIP_EUNSUPPORTED, "the fixed-point form sums the ranks' totals on one wavefront: world <= 64");
            FixedFinal ff{};
            fixed_final_view(c, ff, true);
            hipLaunchKernelGGL(fixed_final_ctrl_kernel, dim3(1), dim3(kWave), 0, c->stream, ff, reinterpret_cast<const uint64_t*>(d_all_totals), world);
            c->final_bookkeep_pending = false;
        }
    } else if (!c->counts_mode) {
        // floating-point form: the ranks' {max, sum, sum of squares} into ctrl.  In the exchange scope the step's plan launch follows
        // and does this on its way in (one launch less per step); the last step has no exchange
        c->scan2_deferred = c->exchange && t + 1 < c->T;
        if (!c->scan2_deferred) launch_scan(c, t, 2, d_all_totals, world, rank);
    }
    else if (t + 1 == c->T) {
        // the final generation's bookkeeping from the population's totals (the same sums a single GPU's hierarchy would hold)
        if (world > kWave) return fail(c, CPPROB_HIP_EUNSUPPORTED, "the prefix-count form sums the ranks' totals on one wavefront: world <= 64");
        CountsFinal f{};
        counts_final_view(c, f, true);
        hipLaunchKernelGGL(counts_final_ctrl_kernel, dim3(1), dim3(kWave), 0, c->stream, f, d_all_totals, world);
        c->final_bookkeep_pending = false;
    }
    HIP_TRY(c, hipGetLastError());
    return 0;
}

// ---- a shard's generation that lost its bits: repaired in the run, as a single context's is (settle_fixed) ----------------------
// Every rank finds the same first offending generation g in its books (they come from the all-gathered totals).  The repair is the
// single context's, with the POPULATION's exact maximum in place of the shard's:
//   repair_begin(g)   generation g's log-weights again from the particle store, this shard's exact maximum -> d_local3 = {key(M), 0, 0}
//   (the caller all-gathers the 24 bytes, as it does a step's totals)
//   repair_end(g)     masses against the largest of the ranks' maxima, the books rewound to where they stood before g,
//                     this shard's {S, Q, key(M)} of the requantised generation -> d_local3
//   (all-gather; cpprob_hip_smc_step_end(g); the exchange behind step g; the steps g + 1 .. T - 1 as ever; cpprob_hip_smc_finish)
}  // extern "C"
template <class Model>
static int shard_repair_begin(cpprob_hip_ctx* c, int g, double* d_local3)
{
    if constexpr (std::is_same<Model, ModelLinearGaussian1D>::value || std::is_same<Model, ModelHmm3>::value || std::is_same<Model, ModelHmmK>::value) {
        std::vector<int32_t> res((size_t)c->T);
        HIP_TRY(c, hipMemcpyAsync(res.data(), c->d_resampled, res.size() * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        int s0 = 0;
        for (int s2 = g - 1; s2 >= 0; --s2) if (res[(size_t)s2]) { s0 = s2 + 1; break; }
        c->cur = (g + 1) & 1;
        hipLaunchKernelGGL(fixed_relogw_kernel<Model>, dim3((unsigned)((c->ld + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream, c->mp, (const double*)c->d_obs,
                           static_cast<const typename Model::store_t*>(c->d_values), c->rs, s0, g, c->n, c->ld, c->d_logw[c->cur]);
        HIP_TRY(c, hipMemsetAsync(c->d_hier, 0, c->hier_entries * sizeof(uint64_t), c->stream));
        const int ka = (g + 1 + c->hier_phase_run) % 3, kb = (ka + 1) % 3;
        FHier f{};
        fhier_view(c, ka, f);
        f.h.to_next = 0; f.h.to_clear = (int64_t)(kb - ka) * (int64_t)c->hier_per_copy;
        hipLaunchKernelGGL(bbf_max_kernel, dim3(c->nb), dim3(kThreads), 0, c->stream, (const double*)c->d_logw[c->cur], c->n, f);
        hipLaunchKernelGGL(fixed_repair_max_kernel, dim3(1), dim3(kWave), 0, c->stream, f, reinterpret_cast<uint64_t*>(d_local3));
        HIP_TRY(c, hipGetLastError());
        return 0;
    }
    return fail(c, CPPROB_HIP_ESTATE, "the fixed-point form serves the state-space models");
}

extern "C" {

int cpprob_hip_smc_first_bad_generation(cpprob_hip_ctx* c, int32_t* h_generation, double* h_gap)
{
    if (!c || !h_generation) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->begun) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_infer_begin has not been called");
    HIP_TRY(c, hipSetDevice(c->device));
    StepCtrl h{};
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *h_generation = (c->fixed_mode && !(h.fix_gap <= kFixGapLimit)) ? h.first_bad : -1;
    if (h_gap) *h_gap = c->fixed_mode ? h.fix_gap : 0.0;
    return 0;
}

int cpprob_hip_smc_repair_begin(cpprob_hip_ctx* c, int32_t g, double* d_local3)
{
    if (!c || !d_local3) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->begun || !c->step_protocol || !c->fixed_mode || !c->keep || !c->d_lz_trace) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_smc_repair_begin: a history-keeping step-protocol run on fixed-point weights");
    if (g < 0 || g >= c->T) return fail(c, CPPROB_HIP_EINVAL, "generation out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = 0;
    dispatch_model(c, [&](auto m) { rc = shard_repair_begin<decltype(m)>(c, g, d_local3); });
    return rc;
}

int cpprob_hip_smc_repair_end(cpprob_hip_ctx* c, int32_t g, const double* d_all3, int32_t world, int32_t rank, double* d_local3)
{
    if (!c || !d_all3 || !d_local3) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->begun || !c->step_protocol || !c->fixed_mode) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_smc_repair_end follows cpprob_hip_smc_repair_begin");
    if (g < 0 || g >= c->T || world < 1 || world > kWave || rank < 0 || rank >= world) return fail(c, CPPROB_HIP_EINVAL, "bad generation / world / rank");
    HIP_TRY(c, hipSetDevice(c->device));
    const int ka = (g + 1 + c->hier_phase_run) % 3, kb = (ka + 1) % 3;
    FHier f{};
    fhier_view(c, ka, f);
    f.h.to_next = 0; f.h.to_clear = (int64_t)(kb - ka) * (int64_t)c->hier_per_copy;
    const uint64_t* keys = reinterpret_cast<const uint64_t*>(d_all3);
    hipLaunchKernelGGL(bbf_quantize_kernel, dim3(c->nb), dim3(kThreads), 0, c->stream, (const double*)c->d_logw[c->cur], c->n, f, c->d_q[c->cur], keys, (int)world);
    hipLaunchKernelGGL(fixed_repair_ctrl_kernel, dim3(1), dim3(kWave), 0, c->stream, c->d_ctrl, f, (int)g, (const int32_t*)c->d_resampled, keys, (int)world);
    hipLaunchKernelGGL(fixed_totals_kernel, dim3(1), dim3(kWave), 0, c->stream, f, reinterpret_cast<uint64_t*>(d_local3));
    HIP_TRY(c, hipGetLastError());
    // the protocol goes on as if step g had just run: its step_end, its exchange, then step g + 1
    c->step_t = g; c->x_plan_t = g - 1; c->plan.t = -1; c->x_all_totals = nullptr; c->ran = false;
    // (a caller of the synchronising exchange calls: the annex as it stood before the exchange that followed step g; a step whose
    //  exchange planned nothing left no mark -- the next marked one stands for it, the columns in use do not shrink)
    for (size_t k = (size_t)g; k < c->annex_used_before.size(); ++k) {
        if (c->annex_used_before[k] >= 0) { c->annex_used = c->annex_used_before[k]; break; }
    }
    for (size_t k = (size_t)g; k < c->annex_used_before.size(); ++k) c->annex_used_before[k] = -1;
    if (g + 1 == c->T) { c->final_from_fixed = true; c->final_copy = ka; c->final_bookkeep_pending = true; }
    c->n_requantised += 1;
    c->fixed_check_pending = false;
    return 0;
}

int cpprob_hip_smc_finish(cpprob_hip_ctx* c)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->begun || !c->step_protocol || c->step_t != c->T - 1) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_smc_finish follows the last step's cpprob_hip_smc_step_end");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->keep) {
        // filtering-only shard: every step left its statistics -- the count form the JOINT population's probabilities (from the
        // all-gathered totals: nothing to combine), the fixed-point form this shard's raw sums and masses (the caller adds them
        // over ranks and divides: cpprob_hip_filter_masses)
        if (c->fixed_mode) hipLaunchKernelGGL(filter_finalize_kernel, dim3(c->T), dim3(kThreads), 0, c->stream, (const double*)c->d_fpart, c->smooth_grid, c->K, c->is_int ? 1 : 0, c->d_stats, 0, c->d_filter_w);
    } else dispatch_model(c, [&](auto m) { launch_smooth<decltype(m)>(c, false); });
    HIP_TRY(c, hipGetLastError());
    c->ran = true;
    c->fixed_check_pending = c->fixed_mode; c->last_was_infer_run = false;
    return 0;
}

int cpprob_hip_filter_masses(cpprob_hip_ctx* c, double** d_masses, int32_t* joint_already)
{
    if (!c || !d_masses || !joint_already) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->begun || c->keep) return fail(c, CPPROB_HIP_ESTATE, "not a filtering-only run (keep_history = 0)");
    *d_masses = c->d_filter_w; *joint_already = c->counts_mode ? 1 : 0;
    return 0;
}

// ---- exchange scope -----------------------------------------------------------------------------
#define GRID1(n) dim3((unsigned)(((n) + 255) / 256)), dim3(256), 0, c->stream

}  // extern "C"

namespace {

int upload_geometry(cpprob_hip_ctx* c, int world, int rank, const uint64_t* h_shard_begin)
{
    if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world) return fail(c, CPPROB_HIP_EINVAL, "bad world/rank (1 <= world <= 63)");
    if (h_shard_begin[rank] != c->cfg.particle_offset || h_shard_begin[rank + 1] - h_shard_begin[rank] != (uint64_t)c->n || h_shard_begin[world] != c->cfg.n_global)
        return fail(c, CPPROB_HIP_EINVAL, "h_shard_begin does not describe this context's shard");
    std::vector<uint64_t> sb(h_shard_begin, h_shard_begin + world + 1);
    if (sb == c->x_shard_begin && c->d_shard_begin) return 0;
    if (!c->d_shard_begin) HIP_TRY(c, hipMalloc(&c->d_shard_begin, (kWorldSlots + 1) * sizeof(int64_t)));
    std::vector<int64_t> h(sb.begin(), sb.end());
    HIP_TRY(c, hipMemcpyAsync(c->d_shard_begin, h.data(), h.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));           // (h is a local)
    c->x_shard_begin = sb;
    return 0;
}

// the plan of the exchange that follows step t, on the device (no host synchronisation)
int launch_plan(cpprob_hip_ctx* c, int t, bool fixed_layout)
{
    ExchangeGeom g{};
    g.world = c->x_world; g.rank = c->x_rank; g.n = c->n; g.shard_begin = c->d_shard_begin;
    g.slot_of_rank = fixed_layout ? c->d_slot_of_rank : nullptr; g.cap = fixed_layout ? c->x_cap : (int64_t)1 << 40;
    g.annex_cap = fixed_layout ? c->annex_cap : (int64_t)1 << 40;      // (callers that synchronise grow the annex themselves)
    g.bytes_per_value = (int)(fixed_layout ? c->ssz : (c->is_int ? sizeof(int32_t) : sizeof(double))); g.sent_per_step = c->d_sent;
    g.no_history = c->keep ? 0 : 1; g.remote = (c->x_remote && c->keep && fixed_layout) ? 1 : 0;
    if (g.remote) { g.rem = c->d_remote; g.annex_all = c->d_annex_all; g.slot_of_rank = nullptr; g.cap = (int64_t)1 << 40; g.trace_words = c->trace_shard_run ? 1 : 0; }
    PlanCountsIn pc{};
    pc.all_totals = c->x_all_totals; pc.n_pop = (double)c->pop_n;
    if (c->fixed_mode) {
        PlanFixedIn pf{};
        pf.all_totals = reinterpret_cast<const uint64_t*>(c->x_all_totals); pf.u0 = host_resample_u0(c->run_seed, (uint64_t)t + 1); pf.n_pop = (double)c->pop_n;
        pf.ess_frac = c->cfg.ess_threshold;
        pf.rs = c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED ? kFixStratified : kFixSystematic; pf.seed = c->run_seed; pf.draw = kResampleDrawBase + (uint64_t)t + 1;
        hipLaunchKernelGGL(exchange_plan_fixed_kernel, dim3(1), dim3(kWave), 0, c->stream, g, pf, t, c->d_annex_base, c->d_xplan);
    } else if (c->counts_mode) {
        pc.e0 = c->h_e_tab[(size_t)t * 4]; pc.e1 = c->h_e_tab[(size_t)t * 4 + 1]; pc.e2 = c->h_e_tab[(size_t)t * 4 + 2];
        pc.u0 = host_resample_u0(c->run_seed, (uint64_t)t + 1);
        pc.rs = c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED ? kFixStratified : kFixSystematic; pc.seed = c->run_seed; pc.draw = kResampleDrawBase + (uint64_t)t + 1;
        hipLaunchKernelGGL((exchange_plan_kernel<true, false>), dim3(1), dim3(kWave), 0, c->stream, g, pc, (const double*)c->d_obound, t, c->d_annex_base, c->d_xplan, ScanArgs{});
    } else if (c->scan2_deferred) {
        const ScanArgs sa = make_scan_args(c, t, 2, c->x_all_totals, c->x_world, c->x_rank);
        hipLaunchKernelGGL((exchange_plan_kernel<false, true>), dim3(1), dim3(kWave), 0, c->stream, g, pc, (const double*)c->d_obound, t, c->d_annex_base, c->d_xplan, sa);
        c->scan2_deferred = false;
    } else {
        hipLaunchKernelGGL((exchange_plan_kernel<false, false>), dim3(1), dim3(kWave), 0, c->stream, g, pc, (const double*)c->d_obound, t, c->d_annex_base, c->d_xplan, ScanArgs{});
    }
    c->x_plan_t = t;
    return 0;
}

template <class Model, class R>
void launch_pack(cpprob_hip_ctx* c, int t, R* d_send, int grid, bool plan_inside = false)
{
    PackArgs<Model, R> a{};
    if (c->x_direct && c->x_fixed) { a.peer_recv = c->d_peer_recv; a.peer_slot = c->d_peer_slot; a.cap = c->x_cap; }
    a.geom.no_history = c->keep ? 0 : 1; a.geom.remote = (c->x_remote && c->keep && c->x_fixed) ? 1 : 0;
    if (a.geom.remote) { a.geom.rem = c->d_remote; a.geom.annex_all = c->d_annex_all; a.peer_recv = nullptr; }
    if (a.geom.remote && c->trace_shard_run) { a.trace_cur = c->d_tr[t & 1]; a.trace_par = t & 1; a.geom.trace_words = 1; }
    if (plan_inside) {
        a.geom.world = c->x_world; a.geom.rank = c->x_rank; a.geom.n = c->n; a.geom.shard_begin = c->d_shard_begin; a.geom.slot_of_rank = c->d_slot_of_rank;
        a.geom.cap = c->x_cap; a.geom.annex_cap = c->annex_cap; a.geom.bytes_per_value = (int)c->ssz; a.geom.sent_per_step = c->d_sent;
        a.annex_base = c->d_annex_base; a.plan_out = c->d_xplan;
        if (a.geom.remote) { a.geom.slot_of_rank = nullptr; a.geom.cap = (int64_t)1 << 40; }     // (no segments: straight into the receivers' annexes)
    }
    a.values = static_cast<const typename Model::store_t*>(c->d_values); a.anc = c->d_anc; a.rs = c->rs; a.n = c->n; a.nb = c->nb;
    a.resampled = c->d_resampled; a.t = t; a.plan = c->d_xplan; a.world = c->x_world; a.rank = c->x_rank; a.send = d_send;
    a.skip = (c->keep && !a.geom.remote && c->d_skip && t >= 2 * kSkipEvery) ? c->d_skip : nullptr;
    const dim3 pgrid((unsigned)grid, a.skip ? (unsigned)(t / kSkipEvery + 1) : 1u);
    a.pc.all_totals = c->x_all_totals; a.pc.n_pop = (double)c->pop_n;
    a.wrel = c->d_wrel[c->cur]; a.bc = c->d_bc; a.bf = c->d_bf; a.ctrl = c->d_ctrl; a.seed = c->run_seed; a.pid0 = c->cfg.particle_offset;
    if constexpr (Model::kWeightTable == 3 && sizeof(typename Model::store_t) == 1) {
        if (c->counts_mode) {
            const int kn = (t + 1 + c->hier_phase_run) % 3;            // the copy step t wrote = the one step t+1 reads
            hier_view(c, kn, a.h);
            a.pc.e0 = c->h_e_tab[(size_t)t * 4]; a.pc.e1 = c->h_e_tab[(size_t)t * 4 + 1]; a.pc.e2 = c->h_e_tab[(size_t)t * 4 + 2];
            a.pc.u0 = host_resample_u0(c->run_seed, (uint64_t)t + 1);
            a.pc.rs = c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED ? kFixStratified : kFixSystematic; a.pc.seed = c->run_seed; a.pc.draw = kResampleDrawBase + (uint64_t)t + 1;
            if (plan_inside) hipLaunchKernelGGL((exchange_pack_kernel<Model, R, kPackCounts, true>), pgrid, dim3(kThreads), 0, c->stream, a);
            else hipLaunchKernelGGL((exchange_pack_kernel<Model, R, kPackCounts, false>), pgrid, dim3(kThreads), 0, c->stream, a);
            return;
        }
    }
    if (c->fixed_mode) {
        const int kn = (t + 1 + c->hier_phase_run) % 3;
        fhier_view(c, kn, a.f);
        a.pf.all_totals = reinterpret_cast<const uint64_t*>(c->x_all_totals); a.pf.u0 = host_resample_u0(c->run_seed, (uint64_t)t + 1); a.pf.n_pop = (double)c->pop_n;
        a.pf.ess_frac = c->cfg.ess_threshold;
        a.pf.rs = c->cfg.resampler == CPPROB_HIP_RESAMPLE_STRATIFIED ? kFixStratified : kFixSystematic; a.pf.seed = c->run_seed; a.pf.draw = kResampleDrawBase + (uint64_t)t + 1;
        a.q_prev = c->d_q[c->cur];                                     // the weights of the generation step t just produced
        if (plan_inside) hipLaunchKernelGGL((exchange_pack_kernel<Model, R, kPackFixed, true>), pgrid, dim3(kThreads), 0, c->stream, a);
        else hipLaunchKernelGGL((exchange_pack_kernel<Model, R, kPackFixed, false>), pgrid, dim3(kThreads), 0, c->stream, a);
        return;
    }
    hipLaunchKernelGGL((exchange_pack_kernel<Model, R, kPackFloat, false>), pgrid, dim3(kThreads), 0, c->stream, a);
}

// Multinomial resampling (strata form) in the exchange scope, remote lineages: what the ranks' boundaries cut (one workgroup per
// boundary), then the packing launch that derives the plan from it and stores every migrant into its column of the receiver's annex.
static bool strata_exchange(const cpprob_hip_ctx* c)
{
    return c->exchange && c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL && (c->counts_mode || c->fixed_mode);
}
template <class Model>
void launch_pack_strata(cpprob_hip_ctx* c, int t, int grid)
{
    const int k = strata_k_of(c);
    const uint32_t* offs = c->d_strata + (size_t)t * (((size_t)1 << k) + 1);      // the strata of the resampling in front of step t + 1
    if (c->x_world > 1) {
        CutArgs ca{};
        ca.world = c->x_world; ca.shard_begin = c->d_shard_begin; ca.offs = offs; ca.k = k; ca.seed = c->run_seed; ca.draw2 = kResampleDrawBase2 + (uint64_t)t + 1;
        ca.totals_u = reinterpret_cast<const uint64_t*>(c->x_all_totals); ca.n_pop = (double)c->pop_n; ca.ess_frac = c->cfg.ess_threshold;
        ca.totals_d = c->x_all_totals;
        if (c->counts_mode) { ca.e0 = c->h_e_tab[(size_t)t * 4]; ca.e1 = c->h_e_tab[(size_t)t * 4 + 1]; ca.e2 = c->h_e_tab[(size_t)t * 4 + 2]; }
        ca.tab = c->d_cut_tab; ca.head = c->d_cut_head; ca.srccnt = c->d_cut_srccnt;
        if (c->counts_mode) hipLaunchKernelGGL((exchange_cut_kernel<true>), dim3((unsigned)(c->x_world - 1)), dim3(kThreads), 0, c->stream, ca);
        else hipLaunchKernelGGL((exchange_cut_kernel<false>), dim3((unsigned)(c->x_world - 1)), dim3(kThreads), 0, c->stream, ca);
    }
    PackStrataArgs<Model> a{};
    a.values = static_cast<const typename Model::store_t*>(c->d_values); a.rs = c->rs; a.n = c->n; a.nb = c->nb; a.t = t;
    a.world = c->x_world; a.rank = c->x_rank; a.pid0 = c->cfg.particle_offset; a.n_pop = (double)c->pop_n;
    a.geom.world = c->x_world; a.geom.rank = c->x_rank; a.geom.n = c->n; a.geom.shard_begin = c->d_shard_begin; a.geom.slot_of_rank = nullptr;
    a.geom.cap = (int64_t)1 << 40; a.geom.annex_cap = c->annex_cap; a.geom.bytes_per_value = (int)c->ssz; a.geom.sent_per_step = c->d_sent;
    a.geom.no_history = 0; a.geom.remote = 1; a.geom.rem = c->d_remote; a.geom.annex_all = c->d_annex_all;
    a.annex_base = c->d_annex_base; a.plan_out = c->d_xplan;
    a.cut = cut_view(c); a.offs = offs; a.k = k; a.seed = c->run_seed; a.draw2 = kResampleDrawBase2 + (uint64_t)t + 1;
    if (c->trace_shard_run) { a.trace_cur = c->d_tr[t & 1]; a.trace_par = t & 1; a.geom.trace_words = 1; }
    const int kn = (t + 1 + c->hier_phase_run) % 3;                    // the copy step t wrote = the one step t + 1 reads
    if constexpr (Model::kWeightTable == 3 && sizeof(typename Model::store_t) == 1) {
        if (c->counts_mode) {
            hier_view(c, kn, a.h);
            a.totals_d = c->x_all_totals; a.e0 = c->h_e_tab[(size_t)t * 4]; a.e1 = c->h_e_tab[(size_t)t * 4 + 1]; a.e2 = c->h_e_tab[(size_t)t * 4 + 2];
            hipLaunchKernelGGL((exchange_pack_strata_kernel<Model, kPackCounts>), dim3((unsigned)grid), dim3(kThreads), 0, c->stream, a);
            return;
        }
    }
    if constexpr (std::is_same<Model, ModelLinearGaussian1D>::value || std::is_same<Model, ModelHmm3>::value || std::is_same<Model, ModelHmmK>::value) {
        fhier_view(c, kn, a.f);
        a.totals_u = reinterpret_cast<const uint64_t*>(c->x_all_totals); a.ess_frac = c->cfg.ess_threshold; a.q_prev = c->d_q[c->cur];
        hipLaunchKernelGGL((exchange_pack_strata_kernel<Model, kPackFixed>), dim3((unsigned)grid), dim3(kThreads), 0, c->stream, a);
    }
}

template <class Model, class R>
void launch_commit(cpprob_hip_ctx* c, int t, const R* d_recv, int grid)
{
    using S = typename Model::store_t;
    hipLaunchKernelGGL((exchange_commit_kernel<S, R>), dim3(grid), dim3(kThreads), 0, c->stream, (const ExchangePlan*)c->d_xplan, c->x_world, d_recv, t,
                       (const int64_t*)c->d_annex_base, static_cast<S*>(c->d_values), c->d_anc, c->rs, c->ld, c->d_skip);
}

// more annex columns: re-stride values[] / anc[] (callers that synchronise per step only)
int grow_annex(cpprob_hip_ctx* c, int64_t need)
{
    int64_t cap = std::max<int64_t>(2 * c->annex_cap, (need + kTile - 1) / kTile * kTile);
    const size_t T = (size_t)c->cap_T, vsz = c->ssz;
    const size_t ldc = c->cap_particles, rs_new = (size_t)c->ld + (size_t)cap, rs_old = (size_t)c->rs;
    void* nv = nullptr; int32_t* na = nullptr;
    HIP_TRY(c, hipMalloc(&nv, T * (ldc + (size_t)cap) * vsz));
    HIP_TRY(c, hipMalloc(&na, T * (ldc + (size_t)cap) * sizeof(int32_t)));
    HIP_TRY(c, hipMemsetAsync(nv, 0, T * (ldc + (size_t)cap) * vsz, c->stream));
    HIP_TRY(c, hipMemsetAsync(na, 0, T * (ldc + (size_t)cap) * sizeof(int32_t), c->stream));
    const size_t used = (size_t)(c->ld + c->annex_used);
    HIP_TRY(c, hipMemcpy2DAsync(nv, rs_new * vsz, c->d_values, rs_old * vsz, used * vsz, (size_t)c->T, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(na, rs_new * 4, c->d_anc, rs_old * 4, used * 4, (size_t)c->T, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_skip) {
        int32_t* ns = nullptr;
        const size_t M = T / kSkipEvery + 1;
        HIP_TRY(c, hipMalloc(&ns, M * (ldc + (size_t)cap) * sizeof(int32_t)));
        HIP_TRY(c, hipMemsetAsync(ns, 0, M * (ldc + (size_t)cap) * sizeof(int32_t), c->stream));
        HIP_TRY(c, hipMemcpy2DAsync(ns, rs_new * 4, c->d_skip, rs_old * 4, used * 4, (size_t)c->T / kSkipEvery + 1, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        dfree(c->d_skip); c->d_skip = ns;
    }
    dfree(c->d_values); dfree(c->d_anc);
    c->d_values = nv; c->d_anc = na; c->annex_cap = cap; c->rs = c->ld + cap;
    return 0;
}

}  // namespace

extern "C" {

// ---- exchange, synchronising form: the caller sizes its buffers from host-visible counts ----
int cpprob_hip_exchange_plan(cpprob_hip_ctx* c, int32_t t, int32_t world, int32_t rank, const uint64_t* h_shard_begin, uint64_t* h_send_counts,
                             uint64_t* h_recv_counts, int32_t* h_do_resample)
{
    if (!c || !h_shard_begin || !h_send_counts || !h_recv_counts) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->exchange) return fail(c, CPPROB_HIP_ESTATE, "the context was not begun with resample_scope = CPPROB_HIP_SCOPE_EXCHANGE");
    if (c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL)
        return fail(c, CPPROB_HIP_EUNSUPPORTED, "multinomial resampling in the exchange scope moves its migrants by remote lineages (cpprob_hip_exchange_setup / _direct / _remote, "
                                                 "then cpprob_hip_exchange_pack_async): the synchronising plan / pack / commit calls serve systematic and stratified resampling");
    if (t < 0 || t >= c->T) return fail(c, CPPROB_HIP_EINVAL, "step out of range");
    if (!c->step_protocol || c->step_t != t || !c->x_all_totals || world != c->x_world || rank != c->x_rank)
        return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_exchange_plan(t) follows cpprob_hip_smc_step_end(t) with the same world / rank");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = upload_geometry(c, world, rank, h_shard_begin)) return rc;
    if (int rc = launch_plan(c, t, false)) return rc;
    HIP_TRY(c, hipGetLastError());
    ExchangePlan h;
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_xplan, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    auto& p = c->plan;
    p.t = t; p.resample = h.resample != 0; p.n_send = (uint64_t)h.n_send; p.n_recv = (uint64_t)h.n_recv; p.l0 = h.l0; p.l1 = h.l1;
    for (int r = 0; r < world; ++r) { h_send_counts[r] = (uint64_t)h.send_cnt[r]; h_recv_counts[r] = (uint64_t)h.recv_cnt[r]; }
    if (h_do_resample) *h_do_resample = p.resample ? 1 : 0;
    if (p.resample && (int64_t)p.n_recv != c->n - (p.l1 - p.l0)) return fail(c, CPPROB_HIP_EDEVICE, "exchange plan does not cover the shard");
    return 0;
}

int cpprob_hip_exchange_pack(cpprob_hip_ctx* c, int32_t t, void* d_send)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    auto& p = c->plan;
    if (!c->exchange || p.t != t) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_exchange_plan(t) has not run");
    if (!p.resample || p.n_send == 0) return 0;
    if (!d_send) return fail(c, CPPROB_HIP_EINVAL, "d_send is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    const int grid = (int)std::min<uint64_t>((p.n_send + kTile - 1) / kTile + 1, 1024);
    dispatch_model(c, [&](auto m) { using M = decltype(m); launch_pack<M, typename M::value_t>(c, t, static_cast<typename M::value_t*>(d_send), grid); });
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_exchange_commit(cpprob_hip_ctx* c, int32_t t, const void* d_recv)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    auto& p = c->plan;
    if (!c->exchange || p.t != t) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_exchange_plan(t) has not run");
    if (!p.resample || p.n_recv == 0) return 0;
    if (!d_recv) return fail(c, CPPROB_HIP_EINVAL, "d_recv is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->ld + c->annex_used + (int64_t)p.n_recv + kTile > (int64_t)INT32_MAX) return fail(c, CPPROB_HIP_EINVAL, "immigrant columns exceed int32 ancestor indices");
    if (c->annex_used_before.size() < (size_t)c->T + 1) c->annex_used_before.assign((size_t)c->T + 1, -1);
    c->annex_used_before[(size_t)t] = c->annex_used;
    if (c->annex_used + (int64_t)p.n_recv > c->annex_cap) { if (int rc = grow_annex(c, c->annex_used + (int64_t)p.n_recv)) return rc; }
    const int grid = (int)std::min<uint64_t>((p.n_recv * (uint64_t)(t + 1) + kThreads - 1) / kThreads, 2048);
    dispatch_model(c, [&](auto m) { using M = decltype(m); launch_commit<M, typename M::value_t>(c, t, static_cast<const typename M::value_t*>(d_recv), grid); });
    HIP_TRY(c, hipGetLastError());
    c->annex_used += (int64_t)p.n_recv;
    return 0;
}

// ---- exchange, stream-ordered form: fixed-capacity transport segments, no host synchronisation inside a run ----
int cpprob_hip_exchange_setup(cpprob_hip_ctx* c, int32_t world, int32_t rank, const uint64_t* h_shard_begin, int32_t all_peers, uint64_t records_per_peer)
{
    if (!c || !h_shard_begin) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->begun || !c->exchange) return fail(c, CPPROB_HIP_ESTATE, "begin the context with resample_scope = CPPROB_HIP_SCOPE_EXCHANGE first");
    if (records_per_peer == 0) return fail(c, CPPROB_HIP_EINVAL, "records_per_peer must be > 0");
    HIP_TRY(c, hipSetDevice(c->device));
    if (int rc = upload_geometry(c, world, rank, h_shard_begin)) return rc;
    std::vector<int32_t> slot((size_t)kWorldSlots, -1);
    c->x_peers.clear();
    if (all_peers == 2 && world != 1) return fail(c, CPPROB_HIP_EINVAL, "all_peers = 2 (a rank that is its own peer) is the one-rank diagnostic");
    for (int r = 0; r < world; ++r) {
        if (r == rank && all_peers != 2) continue;
        if (all_peers || r == rank - 1 || r == rank + 1) { slot[(size_t)r] = (int32_t)c->x_peers.size(); c->x_peers.push_back(r); }
    }
    if (!c->d_slot_of_rank) HIP_TRY(c, hipMalloc(&c->d_slot_of_rank, kWorldSlots * sizeof(int32_t)));
    HIP_TRY(c, hipMemcpy(c->d_slot_of_rank, slot.data(), kWorldSlots * sizeof(int32_t), hipMemcpyHostToDevice));
    c->x_cap = (int64_t)records_per_peer; c->x_mode = all_peers ? 1 : 0; c->x_fixed = true;
    const size_t need = std::max<size_t>(1, c->x_peers.size()) * (size_t)records_per_peer * (size_t)c->T * c->ssz;
    if (need > c->x_buf_bytes) {
        dfree(c->d_xsend); dfree(c->d_xrecv);
        HIP_TRY(c, hipMalloc(&c->d_xsend, need));
        HIP_TRY(c, hipMalloc(&c->d_xrecv, need));
        c->x_buf_bytes = need;
    }
    c->x_world = world; c->x_rank = rank;
    c->x_direct = false; c->x_remote = false; c->trace_shard = false;   // (the peers' buffers may have moved: cpprob_hip_exchange_direct / _remote again)
    if (c->T > c->sent_cap) { dfree(c->d_sent); HIP_TRY(c, hipMalloc(&c->d_sent, (size_t)c->T * sizeof(int64_t))); c->sent_cap = c->T; }
    HIP_TRY(c, hipMemset(c->d_sent, 0, (size_t)c->sent_cap * sizeof(int64_t)));
    return 0;
}

int cpprob_hip_exchange_direct(cpprob_hip_ctx* c, void* const* h_peer_recv)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->x_fixed) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_exchange_setup has not run");
    if (!h_peer_recv) { c->x_direct = false; return 0; }
    HIP_TRY(c, hipSetDevice(c->device));
    // the slot each peer keeps for this rank: its peer list is the ranks around it (or every other rank) in rank order
    std::vector<void*> ptr((size_t)kWorldSlots, nullptr);
    std::vector<int32_t> slot((size_t)kWorldSlots, 0);
    for (int r : c->x_peers) {
        if (!h_peer_recv[r]) return fail(c, CPPROB_HIP_EINVAL, "cpprob_hip_exchange_direct: no receive buffer for a peer rank");
        ptr[(size_t)r] = h_peer_recv[r];
        if (r == c->x_rank) slot[(size_t)r] = 0;                               // (one-rank diagnostic: its own, only, slot)
        else if (c->x_mode) slot[(size_t)r] = c->x_rank < r ? c->x_rank : c->x_rank - 1;
        else slot[(size_t)r] = (c->x_rank == r - 1) ? 0 : (r - 1 >= 0 ? 1 : 0);
    }
    if (!c->d_peer_recv) HIP_TRY(c, hipMalloc(&c->d_peer_recv, kWorldSlots * sizeof(void*)));
    if (!c->d_peer_slot) HIP_TRY(c, hipMalloc(&c->d_peer_slot, kWorldSlots * sizeof(int32_t)));
    HIP_TRY(c, hipMemcpy(c->d_peer_recv, ptr.data(), kWorldSlots * sizeof(void*), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_peer_slot, slot.data(), kWorldSlots * sizeof(int32_t), hipMemcpyHostToDevice));
    c->x_direct = true;
    return 0;
}

// trace words of an exchange-scope shard (hmm<T <= 16> on the count form): the two word arrays over local + annex columns and the
// read-out's counters; `words` = this shard can carry them
static int ensure_shard_trace(cpprob_hip_ctx* c, bool& words)
{
    words = false;
    dispatch_model(c, [&](auto m) { words = counts_eligible<decltype(m)>(c); });
    words = words && c->keep && c->cfg.model == CPPROB_HIP_MODEL_HMM3 && c->T <= kTraceMaxT && !(c->cfg.flags & CPPROB_HIP_FLAG_WALK_READOUT);
    if (!words) return 0;
    if ((size_t)c->rs > c->tr_cap || !c->d_tr[0]) {
        dfree(c->d_tr[0]); dfree(c->d_tr[1]);
        HIP_TRY(c, hipMalloc(&c->d_tr[0], (size_t)c->rs * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_tr[1], (size_t)c->rs * sizeof(uint32_t)));
        HIP_TRY(c, hipMemset(c->d_tr[0], 0, (size_t)c->rs * sizeof(uint32_t)));
        HIP_TRY(c, hipMemset(c->d_tr[1], 0, (size_t)c->rs * sizeof(uint32_t)));
        c->tr_cap = (size_t)c->rs;
    }
    if (!c->d_trace_cnt) {
        HIP_TRY(c, hipMalloc(&c->d_trace_cnt, kTraceCounterWords * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(&c->d_trace_arrive, sizeof(unsigned long long)));
        HIP_TRY(c, hipMemset(c->d_trace_cnt, 0, kTraceCounterWords * sizeof(uint32_t)));
        HIP_TRY(c, hipMemset(c->d_trace_arrive, 0, sizeof(unsigned long long)));
    }
    return 0;
}

int cpprob_hip_exchange_store(cpprob_hip_ctx* c, cpprob_hip_store* out)
{
    if (!c || !out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->begun || !c->exchange || !c->keep) return fail(c, CPPROB_HIP_ESTATE, "no history-keeping exchange-scope run begun");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->annex_cap > c->origin_cap || !c->d_origin) {
        dfree(c->d_origin);
        HIP_TRY(c, hipMalloc(&c->d_origin, (size_t)std::max<int64_t>(c->annex_cap, 1) * sizeof(int64_t)));
        c->origin_cap = c->annex_cap;
    }
    out->d_values = c->d_values; out->d_ancestors = c->d_anc; out->d_origin = c->d_origin; out->row_stride = (uint64_t)c->rs; out->n_local_columns = (uint64_t)c->ld;
    // short discrete traces on the count form: the particles carry their traces across ranks too (trace_words.hpp)
    out->d_trace[0] = nullptr; out->d_trace[1] = nullptr;
    bool words = false;
    if (int rc = ensure_shard_trace(c, words)) return rc;
    if (words) { out->d_trace[0] = c->d_tr[0]; out->d_trace[1] = c->d_tr[1]; }
    return 0;
}

int cpprob_hip_exchange_remote(cpprob_hip_ctx* c, const cpprob_hip_store* h_stores)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->x_fixed || !c->x_direct) return fail(c, CPPROB_HIP_ESTATE, "remote lineages ride the direct transport: cpprob_hip_exchange_setup and _direct first");
    if (!h_stores) { c->x_remote = false; c->trace_shard = false; return 0; }
    if (!c->keep) return 0;                                 // (a filtering-only shard has no lineages to leave anywhere)
    HIP_TRY(c, hipSetDevice(c->device));
    RemoteStores rs{};
    rs.world = c->x_world; rs.rank = c->x_rank;
    for (int r = 0; r < c->x_world; ++r) {
        if (!h_stores[r].d_values || !h_stores[r].d_ancestors || !h_stores[r].d_origin) return fail(c, CPPROB_HIP_EINVAL, "cpprob_hip_exchange_remote: a rank's store is missing");
        rs.values[r] = h_stores[r].d_values; rs.anc[r] = static_cast<const int32_t*>(h_stores[r].d_ancestors); rs.origin[r] = static_cast<const int64_t*>(h_stores[r].d_origin);
        rs.rs[r] = (int64_t)h_stores[r].row_stride; rs.ld[r] = (int64_t)h_stores[r].n_local_columns;
    }
    bool words = true;
    for (int r = 0; r < c->x_world; ++r) words = words && h_stores[r].d_trace[0] && h_stores[r].d_trace[1];
    for (int r = 0; r < c->x_world; ++r)
        for (int k = 0; k < 2; ++k) rs.trace[k][r] = words ? static_cast<uint32_t*>(const_cast<void*>(h_stores[r].d_trace[k])) : nullptr;
    c->trace_shard = words && c->d_tr[0] != nullptr;
    if (!c->d_remote) HIP_TRY(c, hipMalloc(&c->d_remote, sizeof(RemoteStores)));
    if (c->T + 1 > c->annex_all_T || !c->d_annex_all) {
        dfree(c->d_annex_all);
        HIP_TRY(c, hipMalloc(&c->d_annex_all, (size_t)(c->T + 1) * kWorldSlots * sizeof(int64_t)));
        c->annex_all_T = c->T + 1;
    }
    HIP_TRY(c, hipMemcpy(c->d_remote, &rs, sizeof rs, hipMemcpyHostToDevice));
    c->x_remote = true;
    return 0;
}

int cpprob_hip_exchange_traffic(cpprob_hip_ctx* c, int64_t* h_sent_per_step, size_t n_steps, uint64_t* h_records, uint64_t* h_bytes)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->exchange || !c->d_xplan || !c->d_sent) return fail(c, CPPROB_HIP_ESTATE, "no exchange-scope run on a fixed transport");
    HIP_TRY(c, hipSetDevice(c->device));
    ExchangePlan h;
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_xplan, sizeof h, hipMemcpyDeviceToHost, c->stream));
    if (h_sent_per_step) {
        if (n_steps < (size_t)c->T) return fail(c, CPPROB_HIP_EINVAL, "h_sent_per_step too small");
        HIP_TRY(c, hipMemcpyAsync(h_sent_per_step, c->d_sent, (size_t)c->T * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h_records) *h_records = (uint64_t)h.run_records;
    if (h_bytes) *h_bytes = (uint64_t)h.run_bytes;
    return 0;
}

int cpprob_hip_exchange_transport(cpprob_hip_ctx* c, void** d_send, void** d_recv, int32_t* n_peers, int32_t* h_peers, uint64_t* records_per_peer,
                                  uint64_t* bytes_per_value)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->x_fixed) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_exchange_setup has not run");
    if (d_send) *d_send = c->d_xsend;
    if (d_recv) *d_recv = c->d_xrecv;
    if (n_peers) *n_peers = (int32_t)c->x_peers.size();
    if (h_peers) for (size_t i = 0; i < c->x_peers.size(); ++i) h_peers[i] = c->x_peers[i];
    if (records_per_peer) *records_per_peer = (uint64_t)c->x_cap;
    if (bytes_per_value) *bytes_per_value = (uint64_t)c->ssz;
    return 0;
}

int cpprob_hip_exchange_pack_async(cpprob_hip_ctx* c, int32_t t)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->exchange || !c->x_fixed) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_exchange_setup has not run");
    if (!c->step_protocol || c->step_t != t || !c->x_all_totals) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_exchange_pack_async(t) follows cpprob_hip_smc_step_end(t)");
    if (t < 0 || t + 1 >= c->T) return fail(c, CPPROB_HIP_EINVAL, "no exchange follows the last step");
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->keep && !c->counts_mode && !c->fixed_mode) return fail(c, CPPROB_HIP_EUNSUPPORTED, "filtering-only shards run on the integer forms of the step (systematic resampling)");
    // count form: the plan is a pure function of the all-gathered totals -- every packing workgroup derives it on its first wavefront
    // and workgroup 0 stores it (one launch less per step); floating-point form: the plan launch also combines the ranks' totals
    const bool plan_inside = c->counts_mode || c->fixed_mode;
    if (c->cfg.resampler == CPPROB_HIP_RESAMPLE_MULTINOMIAL) {
        if (!strata_exchange(c)) return fail(c, CPPROB_HIP_EUNSUPPORTED, "multinomial resampling in the exchange scope runs on the integer forms of the step");
        if (c->x_peers.empty()) { c->x_plan_t = t; c->plan.t = t; return 0; }     // (a group of one: every threshold lies in its own mass)
        if (!(c->x_remote && c->keep))
            return fail(c, CPPROB_HIP_EUNSUPPORTED, "multinomial resampling in the exchange scope needs remote lineages (every rank addresses every rank's particle store: "
                                                     "cpprob_hip_exchange_remote) -- its migrants are not one interval per peer, so the segment transports do not carry them");
        c->x_plan_t = t;
        const int grid = (int)std::min<int64_t>(std::max<int64_t>(1, (c->x_cap + kTile - 1) / kTile), 256);
        dispatch_model(c, [&](auto m) { launch_pack_strata<decltype(m)>(c, t, grid); });
        HIP_TRY(c, hipGetLastError());
        c->plan.t = t;
        return 0;
    }
    if (c->x_peers.empty()) {
        // nobody to exchange with (a group of one): the count form's step derives everything it needs from the totals; the
        // floating-point form's plan launch still combines the ranks' totals into ctrl
        if (!plan_inside) { if (int rc = launch_plan(c, t, true)) return rc; HIP_TRY(c, hipGetLastError()); }
        c->x_plan_t = t; c->plan.t = t;
        return 0;
    }
    if (!plan_inside) { if (int rc = launch_plan(c, t, true)) return rc; }
    else c->x_plan_t = t;
    // enough workgroups for a full segment per peer; those beyond the planned tiles leave at once
    const int grid = (int)std::min<int64_t>(std::max<int64_t>(1, (c->x_cap + kTile - 1) / kTile), 256);
    dispatch_model(c, [&](auto m) { using M = decltype(m); launch_pack<M, typename M::store_t>(c, t, static_cast<typename M::store_t*>(c->d_xsend), grid, plan_inside); });
    HIP_TRY(c, hipGetLastError());
    c->plan.t = t;
    return 0;
}

int cpprob_hip_exchange_commit_async(cpprob_hip_ctx* c, int32_t t)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->exchange || !c->x_fixed || c->x_plan_t != t) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_exchange_pack_async(t) has not run");
    if (c->x_peers.empty()) return 0;
    if (c->x_remote && c->keep) return 0;                  // (remote lineages: the senders stored into this rank's annex themselves)
    HIP_TRY(c, hipSetDevice(c->device));
    dispatch_model(c, [&](auto m) { using M = decltype(m); launch_commit<M, typename M::store_t>(c, t, static_cast<const typename M::store_t*>(c->d_xrecv), 256); });
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_exchange_status(cpprob_hip_ctx* c, int32_t* h_overflow, uint64_t* h_annex_used)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->exchange || !c->d_xplan) return fail(c, CPPROB_HIP_ESTATE, "no exchange-scope run");
    HIP_TRY(c, hipSetDevice(c->device));
    int32_t head[2] = {0, 0};
    std::vector<int64_t> ab((size_t)c->T + 1, 0);
    HIP_TRY(c, hipMemcpyAsync(head, c->d_xplan, sizeof head, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(ab.data(), c->d_annex_base, ab.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (h_overflow) *h_overflow = head[1];
    if (h_annex_used) *h_annex_used = (uint64_t)(c->T >= 2 ? ab[(size_t)c->T - 1] : 0);
    return 0;
}

int cpprob_hip_infer_summary(cpprob_hip_ctx* c, cpprob_hip_summary* out)
{
    if (!c || !out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    if (int rc = settle_fixed(c)) return rc;
    StepCtrl h{};
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    out->log_evidence = h.log_z;
    out->ess_final = h.ess;
    out->log_norm = h.M + std::log(h.W);
    out->max_logw = h.M;
    out->n_predict = c->T;
    out->stats_per_predict = c->K;
    out->is_int = c->is_int ? 1 : 0;
    out->n_resampled = h.n_resampled;
    out->step_form = c->cfg.algorithm != CPPROB_HIP_ALG_SMC ? CPPROB_HIP_FORM_FLOAT : (c->fixed_mode ? CPPROB_HIP_FORM_FIXED : (c->counts_mode ? CPPROB_HIP_FORM_COUNTS : CPPROB_HIP_FORM_FLOAT));
    out->n_requantised = c->fixed_mode ? c->n_requantised : 0;
    return 0;
}

// Everything cpprob::inference reads when a run is over -- the summary, the per-predict statistics, the step trace -- behind ONE stream
// synchronisation, through pinned memory (three calls of the single-purpose functions above cost three host round trips, each
// a blocking copy into pageable memory: ~40 us on a 150-us run).
static void fill_summary(const cpprob_hip_ctx* c, const StepCtrl& h, cpprob_hip_summary* out)
{
    out->log_evidence = h.log_z;
    out->ess_final = h.ess;
    out->log_norm = h.M + std::log(h.W);
    out->max_logw = h.M;
    out->n_predict = c->T;
    out->stats_per_predict = c->K;
    out->is_int = c->is_int ? 1 : 0;
    out->n_resampled = h.n_resampled;
    out->step_form = c->cfg.algorithm != CPPROB_HIP_ALG_SMC ? CPPROB_HIP_FORM_FLOAT : (c->fixed_mode ? CPPROB_HIP_FORM_FIXED : (c->counts_mode ? CPPROB_HIP_FORM_COUNTS : CPPROB_HIP_FORM_FLOAT));
    out->n_requantised = c->fixed_mode ? c->n_requantised : 0;
}
int cpprob_hip_infer_results(cpprob_hip_ctx* c, cpprob_hip_summary* out, double* h_stats, size_t n_doubles, double* h_ess, int32_t* h_resampled)
{
    if (!c || !out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t T = (size_t)c->T, need = T * (size_t)c->K;
    if (h_stats && n_doubles < need) return fail(c, CPPROB_HIP_EINVAL, "h_stats too small");
    const bool sis = c->cfg.algorithm == CPPROB_HIP_ALG_SIS;
    const size_t o_ctrl = 0, o_stats = (sizeof(StepCtrl) + 15) / 16 * 16, o_ess = o_stats + need * sizeof(double), o_res = o_ess + T * sizeof(double), total = o_res + T * sizeof(int32_t);
    if (int rc = pin_reserve(c, total)) return rc;
    for (int pass = 0; pass < 2; ++pass) {
        char* const pin = pin_data(c);
        PinPack pk;
        pk.add(c->d_ctrl, pin + o_ctrl, sizeof(StepCtrl));
        if (h_stats) pk.add(c->d_stats, pin + o_stats, need * sizeof(double));
        if (h_ess && sis) pk.add(c->d_ess + (T - 1), pin + o_ess + (T - 1) * sizeof(double), sizeof(double));
        else if (h_ess) pk.add(c->d_ess, pin + o_ess, T * sizeof(double));
        if (h_resampled && !sis) pk.add(c->d_resampled, pin + o_res, T * sizeof(int32_t));
        if (int rc = pk.launch(c)) return rc;
        if (int rc = pin_wait(c)) return rc;
        StepCtrl h;
        std::memcpy(&h, pin + o_ctrl, sizeof h);
        if (pass == 0 && c->fixed_check_pending) {
            // (a fixed-point run is settled on the same read-back: only a generation that lost its bits costs more round trips)
            if (h.fix_gap <= kFixGapLimit) c->fixed_check_pending = false;
            else { if (int rc = settle_fixed(c)) return rc; continue; }
        }
        fill_summary(c, h, out);
        if (h_stats) std::memcpy(h_stats, pin + o_stats, need * sizeof(double));
        if (h_ess) {
            std::memcpy(h_ess, pin + o_ess, T * sizeof(double));
            if (sis) for (size_t t = 0; t + 1 < T; ++t) h_ess[t] = 0.0;                  // (one weighting pass: only the final entry is defined)
        }
        if (h_resampled) {
            if (sis) for (size_t t = 0; t < T; ++t) h_resampled[t] = 0;
            else std::memcpy(h_resampled, pin + o_res, T * sizeof(int32_t));
        }
        return 0;
    }
    return fail(c, CPPROB_HIP_ESTATE, "the run did not settle");
}

int cpprob_hip_infer_stats(cpprob_hip_ctx* c, double* h_stats, size_t n_doubles)
{
    if (!c || !h_stats) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    if (int rc = settle_fixed(c)) return rc;
    const size_t need = (size_t)c->T * c->K;
    if (n_doubles < need) return fail(c, CPPROB_HIP_EINVAL, "h_stats too small");
    HIP_TRY(c, hipMemcpyAsync(h_stats, c->d_stats, need * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int cpprob_hip_infer_results_device(cpprob_hip_ctx* c, double* d_out, size_t n_doubles)
{
    if (!c || !d_out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    const size_t need = 4 + (size_t)c->T * c->K;
    if (n_doubles < need) return fail(c, CPPROB_HIP_EINVAL, "d_out too small");
    HIP_TRY(c, hipSetDevice(c->device));
    const int ns = c->T * c->K;
    hipLaunchKernelGGL(pack_results_kernel, dim3((unsigned)((ns + 255) / 256)), dim3(256), 0, c->stream, c->d_ctrl, c->d_stats, ns, d_out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_infer_step_trace(cpprob_hip_ctx* c, double* h_ess, int32_t* h_resampled)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    if (int rc = settle_fixed(c)) return rc;
    if (c->cfg.algorithm == CPPROB_HIP_ALG_SIS) {
        // one weighting pass: only the final entry is defined
        if (h_ess) { for (int t = 0; t < c->T; ++t) h_ess[t] = 0.0; }
        if (h_resampled) { for (int t = 0; t < c->T; ++t) h_resampled[t] = 0; }
        if (h_ess) HIP_TRY(c, hipMemcpyAsync(h_ess + (c->T - 1), c->d_ess + (c->T - 1), sizeof(double), hipMemcpyDeviceToHost, c->stream));
    } else {
        if (h_ess) HIP_TRY(c, hipMemcpyAsync(h_ess, c->d_ess, (size_t)c->T * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (h_resampled) HIP_TRY(c, hipMemcpyAsync(h_resampled, c->d_resampled, (size_t)c->T * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int copy_rows(cpprob_hip_ctx* c, void* h, const void* d, size_t elem, size_t rows, size_t n_bytes, size_t stride)
{
    const size_t need = rows * (size_t)c->n * elem;
    if (n_bytes < need) return fail(c, CPPROB_HIP_EINVAL, "host buffer too small");
    HIP_TRY(c, hipMemcpy2DAsync(h, (size_t)c->n * elem, d, stride * elem, (size_t)c->n * elem, rows, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
}

int cpprob_hip_copy_values(cpprob_hip_ctx* c, void* h, size_t n_bytes)
{
    if (!c || !h) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    if (int rc = settle_fixed(c)) return rc;
    if (!c->keep) return fail(c, CPPROB_HIP_ESTATE, "the run kept no history (keep_history = 0): traces and ancestors do not exist; its statistics are the filtering ones");
    const size_t vsz = c->is_int ? sizeof(int32_t) : sizeof(double);
    if (c->ssz == vsz) return copy_rows(c, h, c->d_values, vsz, (size_t)c->T, n_bytes, (size_t)c->rs);
    // narrow store: widen to the value type on the device first (d_paths doubles as the staging buffer)
    if (!c->d_paths) HIP_TRY(c, hipMalloc(&c->d_paths, (size_t)c->cap_T * c->cap_particles * vsz));
    dispatch_model(c, [&](auto m) {
        using S = typename decltype(m)::store_t; using V = typename decltype(m)::value_t;
        hipLaunchKernelGGL((widen_rows_kernel<S, V>), GRID1(c->n), static_cast<const S*>(c->d_values), c->rs, c->T, c->n, c->ld, static_cast<V*>(c->d_paths));
    });
    HIP_TRY(c, hipGetLastError());
    return copy_rows(c, h, c->d_paths, vsz, (size_t)c->T, n_bytes, (size_t)c->ld);
}

int cpprob_hip_copy_ancestors(cpprob_hip_ctx* c, int32_t* h, size_t n_bytes)
{
    if (!c || !h) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    if (int rc = settle_fixed(c)) return rc;
    if (!c->keep) return fail(c, CPPROB_HIP_ESTATE, "the run kept no history (keep_history = 0): traces and ancestors do not exist; its statistics are the filtering ones");
    if (c->cfg.algorithm != CPPROB_HIP_ALG_SMC) return fail(c, CPPROB_HIP_ESTATE, "SIS keeps no ancestors (every trace is its own line)");
    if (int rc = copy_rows(c, h, c->d_anc, sizeof(int32_t), (size_t)c->T, n_bytes, (size_t)c->rs)) return rc;
    if (c->final_from_fixed) {
        // the fixed-point step stores a row only where it resampled: the identity of the other rows is written here
        std::vector<int32_t> res((size_t)c->T, 0);
        HIP_TRY(c, hipMemcpy(res.data(), c->d_resampled, (size_t)c->T * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (int t = 1; t < c->T; ++t)
            if (!res[(size_t)t - 1]) for (int64_t i = 0; i < c->n; ++i) h[(size_t)t * (size_t)c->n + (size_t)i] = (int32_t)i;
    }
    return 0;
}

int cpprob_hip_copy_logw(cpprob_hip_ctx* c, double* h, size_t n_bytes)
{
    if (!c || !h) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    if (int rc = settle_fixed(c)) return rc;
    if (c->final_from_counts) {
        // the run kept no log-weight array (they are a function of the final states): write it out now
        HIP_TRY(c, hipSetDevice(c->device));
        dispatch_model(c, [&](auto m) {
            using M = decltype(m);
            if constexpr (M::kWeightTable == 3 && sizeof(typename M::store_t) == 1) {
                const double* ll = &c->h_ll_tab[(size_t)(c->T - 1) * 3];
                hipLaunchKernelGGL(logw_from_states_kernel<M>, dim3((unsigned)((c->ld + kThreads - 1) / kThreads)), dim3(kThreads), 0, c->stream,
                                   static_cast<const typename M::store_t*>(c->d_values) + (int64_t)(c->keep ? c->T - 1 : ((c->T - 1) & 1)) * c->rs, c->n, c->ld, ll[0], ll[1], ll[2],
                                   c->d_logw[c->cur]);
            }
        });
    }
    return copy_rows(c, h, c->d_logw[c->cur], sizeof(double), 1, n_bytes, (size_t)c->ld);
}

int cpprob_hip_copy_paths(cpprob_hip_ctx* c, void* h, size_t n_bytes)
{
    if (!c || !h) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (!c->ran) return fail(c, CPPROB_HIP_ESTATE, "no finished run");
    if (int rc = settle_fixed(c)) return rc;
    if (!c->keep) return fail(c, CPPROB_HIP_ESTATE, "the run kept no history (keep_history = 0): traces and ancestors do not exist; its statistics are the filtering ones");
    const size_t vsz = c->is_int ? sizeof(int32_t) : sizeof(double);
    // SIS: every trace is its own line -- the paths are the values (and a fused-read-out run keeps no linear weights to re-run the read-out on)
    if (c->cfg.algorithm == CPPROB_HIP_ALG_SIS) return cpprob_hip_copy_values(c, h, n_bytes);
    if (!c->d_paths) HIP_TRY(c, hipMalloc(&c->d_paths, (size_t)c->cap_T * c->cap_particles * vsz));
    dispatch_model(c, [&](auto m) { launch_smooth<decltype(m)>(c, true); });
    HIP_TRY(c, hipGetLastError());
    return copy_rows(c, h, c->d_paths, vsz, (size_t)c->T, n_bytes, (size_t)c->ld);
}

// ---- building blocks ----------------------------------------------------------------------

#define BB_PRELUDE(c)                                                         \
    if (!(c)) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");         \
    HIP_TRY(c, hipSetDevice((c)->device));

int cpprob_hip_philox_blocks(cpprob_hip_ctx* c, uint64_t seed, uint64_t pid0, uint64_t draw, size_t n, uint32_t* d_out)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(philox_blocks_kernel, GRID1(n), seed, pid0, draw, (int64_t)n, d_out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_draw_normal(cpprob_hip_ctx* c, uint64_t seed, uint64_t pid0, uint64_t draw, double mean, double sigma, size_t n, double* d_out)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(draw_normal_kernel, GRID1(n), seed, pid0, draw, mean, sigma, (int64_t)n, d_out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_draw_uniform_smallint(cpprob_hip_ctx* c, uint64_t seed, uint64_t pid0, uint64_t draw, int64_t a, int64_t b, size_t n, int32_t* d_out)
{
    BB_PRELUDE(c);
    if (b < a) return fail(c, CPPROB_HIP_EINVAL, "uniform_smallint needs a <= b");
    if (n == 0) return 0;
    hipLaunchKernelGGL(draw_smallint_kernel, GRID1(n), seed, pid0, draw, a, b, (int64_t)n, d_out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

static int make_dw(cpprob_hip_ctx* c, const double* w, int32_t k, DiscreteW& dw)
{
    if (!w || k < 1 || k > 8) return fail(c, CPPROB_HIP_EINVAL, "discrete: need 1 <= k <= 8 weights");
    std::memset(&dw, 0, sizeof dw);
    for (int i = 0; i < k; ++i) dw.w[i] = w[i];
    dw.k = k;
    return 0;
}

int cpprob_hip_draw_discrete(cpprob_hip_ctx* c, uint64_t seed, uint64_t pid0, uint64_t draw, const double* h_w, int32_t k, size_t n, int32_t* d_out)
{
    BB_PRELUDE(c);
    DiscreteW dw;
    if (int rc = make_dw(c, h_w, k, dw)) return rc;
    if (n == 0) return 0;
    hipLaunchKernelGGL(draw_discrete_kernel, GRID1(n), seed, pid0, draw, dw, (int64_t)n, d_out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_draw_uniform_real(cpprob_hip_ctx* c, uint64_t seed, uint64_t pid0, uint64_t draw, double a, double b, size_t n, double* d_out)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(draw_uniform_real_kernel, GRID1(n), seed, pid0, draw, a, b, (int64_t)n, d_out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_draw_poisson(cpprob_hip_ctx* c, uint64_t seed, uint64_t pid0, uint64_t draw, double mean, size_t n, int32_t* d_out)
{
    BB_PRELUDE(c);
    if (!(mean >= 0.0)) return fail(c, CPPROB_HIP_EINVAL, "poisson needs mean >= 0");
    if (n == 0) return 0;
    hipLaunchKernelGGL(draw_poisson_kernel, GRID1(n), seed, pid0, draw, mean, (int64_t)n, d_out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_logpdf_normal(cpprob_hip_ctx* c, const double* x, const double* mean, const double* sigma, size_t n, double* out)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(logpdf_normal_kernel, GRID1(n), x, mean, sigma, (int64_t)n, out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_logpdf_uniform_real(cpprob_hip_ctx* c, const double* x, const double* a, const double* b, size_t n, double* out)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(logpdf_uniform_real_kernel, GRID1(n), x, a, b, (int64_t)n, out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_logpdf_poisson(cpprob_hip_ctx* c, const int32_t* x, const double* mean, size_t n, double* out)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(logpdf_poisson_kernel, GRID1(n), x, mean, (int64_t)n, out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_logpdf_uniform_smallint(cpprob_hip_ctx* c, const int32_t* x, int64_t a, int64_t b, size_t n, double* out)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(logpdf_smallint_kernel, GRID1(n), x, a, b, (int64_t)n, out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_logpdf_discrete(cpprob_hip_ctx* c, const int32_t* x, const double* h_w, int32_t k, size_t n, double* out)
{
    BB_PRELUDE(c);
    DiscreteW dw;
    if (int rc = make_dw(c, h_w, k, dw)) return rc;
    if (n == 0) return 0;
    hipLaunchKernelGGL(logpdf_discrete_kernel, GRID1(n), x, dw, (int64_t)n, out);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

}  // extern "C"
namespace {
// the fixed-point weight of a log-weight x against the reference 0, as a double (cpprob/detail/fixed_mass.hpp: fix_weight)
__global__ void fix_weight_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)fix_weight(x[i], 0.0);
}
}  // namespace
extern "C" {
int cpprob_hip_fastmath(cpprob_hip_ctx* c, int32_t which, const double* d_x, size_t n, double* d_out0, double* d_out1)
{
    BB_PRELUDE(c);
    if (which < 0 || which > 3) return fail(c, CPPROB_HIP_EINVAL, "which: 0 log01, 1 sincospi02, 2 exp_nonpos, 3 fix_weight");
    if (!d_x || !d_out0 || (which == 1 && !d_out1)) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (n == 0) return 0;
    if (which == 3) hipLaunchKernelGGL(fix_weight_kernel, GRID1(n), d_x, (int64_t)n, d_out0);
    else hipLaunchKernelGGL(fastmath_kernel, GRID1(n), (int)which, d_x, (int64_t)n, d_out0, d_out1);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_logsumexp_ess(cpprob_hip_ctx* c, const double* d_logw, size_t n, double* h_out3)
{
    BB_PRELUDE(c);
    if (!d_logw || !h_out3) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (n == 0) { h_out3[0] = 0; h_out3[1] = 0; h_out3[2] = 0; return 0; }   // empty: value-initialised (empirical_distribution.hpp:131-133)
    if (int rc = bb_normalise(c, d_logw, n, (double)n)) return rc;
    StepCtrl h{};
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_bb_ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    h_out3[0] = h.M; h_out3[1] = h.M + std::log(h.W); h_out3[2] = h.ess;
    return 0;
}

}  // extern "C"
template <class Col>
static int column_stats(cpprob_hip_ctx* c, const typename Col::value_t* d_x, const double* d_logw, size_t n, double* h_raw, double* h_lse_ess)
{
    using V = typename Col::value_t;
    if (int rc = bb_normalise(c, d_logw, n, (double)n)) return rc;
    const int nb = (int)((n + kTile - 1) / kTile);
    const int64_t ld = (int64_t)nb * kTile;
    V* col = static_cast<V*>(c->d_bb_col);
    hipLaunchKernelGGL(pad_copy_kernel<V>, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, c->stream, d_x, (int64_t)n, ld, col);
    SmoothArgs<Col> a{};
    a.values = col; a.anc = nullptr; a.wrel = c->d_bb_wrel; a.bf = c->d_bb_bf; a.ctrl = c->d_bb_ctrl; a.resampled = nullptr; a.T = 1; a.n = (int64_t)n; a.ld = ld; a.rs = ld;
    a.identity = 1; a.stats_part = c->d_bb_stats_part; a.paths = nullptr;
    const int grid = std::min(nb, 2048);
    hipLaunchKernelGGL(smooth_kernel<Col>, dim3(grid), dim3(kThreads), (size_t)kWaves * Col::kStats * sizeof(double), c->stream, a);
    // is_int = 1 -> plain normalised sums
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kThreads), 0, c->stream, c->d_bb_stats_part, grid, 1, Col::kStats, 1, c->d_bb_ctrl, c->d_bb_stats, 1);
    HIP_TRY(c, hipGetLastError());
    StepCtrl h{};
    HIP_TRY(c, hipMemcpyAsync(&h, c->d_bb_ctrl, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(h_raw, c->d_bb_stats, Col::kStats * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    h_lse_ess[0] = h.M + std::log(h.W); h_lse_ess[1] = h.ess;
    return 0;
}
// The same for `n_cols` columns against ONE log-weight array: one normalisation, the columns as the rows of a T = n_cols "trace" (the
// read-out kernel with identity ancestors), one finalize, one copy -- five launches and one synchronisation, whatever the number of
// predict hits (the per-column form costs six launches and a synchronisation each: 0.75 ms for the 16 hits of hmm<16>).
template <class Col>
static int columns_stats(cpprob_hip_ctx* c, const typename Col::value_t* d_x, size_t n_cols, size_t col_stride, const double* d_logw, size_t n, double* h_raw /*[n_cols][kStats]*/,
                         double* h_lse_ess)
{
    using V = typename Col::value_t;
    constexpr size_t kChunk = 64;                                  // columns per launch: LDS of the read-out = 4 waves x columns x kStats doubles
    if (int rc = bb_normalise(c, d_logw, n, (double)n)) return rc;
    const int nb = (int)((n + kTile - 1) / kTile);
    const int64_t ld = (int64_t)nb * kTile;
    const int grid = std::min(nb, 2048);
    const size_t chunk = std::min(n_cols, kChunk);
    // columns that already sit a whole number of tiles apart are read where they lie (the caller keeps the slots behind n finite: they weigh
    // nothing); others are copied into tile-padded scratch first (80 MB each way for one column of 10^7 doubles: 40 us of a 0.2 ms run)
    const bool in_place = col_stride % (size_t)kTile == 0 && col_stride >= (size_t)ld;
    const size_t col_bytes = in_place ? 0 : chunk * (size_t)ld * sizeof(V), part_doubles = chunk * Col::kStats * (size_t)grid, stat_doubles = chunk * Col::kStats;
    if (col_bytes > c->bb_cols_bytes) { dfree(c->d_bb_cols); HIP_TRY(c, hipMalloc(&c->d_bb_cols, col_bytes)); c->bb_cols_bytes = col_bytes; }
    if (part_doubles > c->bb_cols_part) { dfree(c->d_bb_cols_part); HIP_TRY(c, hipMalloc(&c->d_bb_cols_part, part_doubles * sizeof(double))); c->bb_cols_part = part_doubles; }
    if (stat_doubles > c->bb_cols_stat) { dfree(c->d_bb_cols_stat); HIP_TRY(c, hipMalloc(&c->d_bb_cols_stat, stat_doubles * sizeof(double))); c->bb_cols_stat = stat_doubles; }
    const size_t stats_bytes = n_cols * Col::kStats * sizeof(double), ride_off = (stats_bytes + sizeof(StepCtrl) + 15) / 16 * 16;
    if (int rc = pin_reserve(c, ride_off + c->ride_bytes)) return rc;
    for (size_t k0 = 0; k0 < n_cols; k0 += chunk) {
        const size_t nk = std::min(chunk, n_cols - k0);
        const V* cols = d_x + k0 * col_stride;
        const int64_t rs = in_place ? (int64_t)col_stride : ld;
        if (!in_place) {
            V* pad = static_cast<V*>(c->d_bb_cols);
            hipLaunchKernelGGL(pad_copy_cols_kernel<V>, dim3((unsigned)((ld + 255) / 256), (unsigned)nk), dim3(256), 0, c->stream, d_x + k0 * col_stride, (int64_t)n, (int64_t)col_stride, ld, pad);
            cols = pad;
        }
        SmoothArgs<Col> a{};
        a.values = const_cast<V*>(cols); a.anc = nullptr; a.wrel = c->d_bb_wrel; a.bf = c->d_bb_bf; a.ctrl = c->d_bb_ctrl; a.resampled = nullptr; a.T = (int)nk; a.n = (int64_t)n; a.ld = rs; a.rs = rs;
        a.identity = 1; a.stats_part = c->d_bb_cols_part; a.paths = nullptr;
        hipLaunchKernelGGL(smooth_kernel<Col>, dim3(grid), dim3(kThreads), (size_t)kWaves * nk * Col::kStats * sizeof(double), c->stream, a);
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)nk), dim3(kThreads), 0, c->stream, c->d_bb_cols_part, grid, (int)nk, Col::kStats, 1, c->d_bb_ctrl, c->d_bb_cols_stat, 1);
        HIP_TRY(c, hipGetLastError());
        // (results through ONE short launch into pinned memory, the caller's riding read-back with the last chunk's: see lineage_stats)
        PinPack pk;
        pk.add(c->d_bb_cols_stat, pin_data(c) + k0 * Col::kStats * sizeof(double), nk * Col::kStats * sizeof(double));
        if (k0 + chunk >= n_cols) {
            pk.add(c->d_bb_ctrl, pin_data(c) + stats_bytes, sizeof(StepCtrl));
            pk.add(c->ride_src, pin_data(c) + ride_off, c->ride_bytes);
        }
        if (int rc = pk.launch(c)) return rc;
        if (k0 + chunk < n_cols) HIP_TRY(c, hipStreamSynchronize(c->stream));     // (the scratch is reused by the next chunk)
    }
    const bool riding = c->ride_bytes != 0;
    if (int rc = pin_wait(c)) return rc;
    StepCtrl h;
    std::memcpy(h_raw, pin_data(c), stats_bytes);
    std::memcpy(&h, pin_data(c) + stats_bytes, sizeof h);
    if (riding) { std::memcpy(c->ride_dst, pin_data(c) + ride_off, c->ride_bytes); c->ride_src = nullptr; c->ride_dst = nullptr; c->ride_bytes = 0; }
    h_lse_ess[0] = h.M + std::log(h.W); h_lse_ess[1] = h.ess;
    return 0;
}
// first_row[t] .. first_row[t + 1]: the records made in generation t's slots (h_gen[h] = the generation of record h, non-decreasing)
static int upload_first_rows(cpprob_hip_ctx* c, const int32_t* h_gen, int32_t H, int32_t T)
{
    std::vector<int32_t> first((size_t)T + 1, 0);
    int32_t prev = 0;
    for (int32_t h = 0; h < H; ++h) {
        if (h_gen[h] < prev || h_gen[h] >= T) return fail(c, CPPROB_HIP_EINVAL, "h_gen must be non-decreasing and < T");
        prev = h_gen[h];
        first[(size_t)h_gen[h] + 1] += 1;
    }
    for (int t = 0; t < T; ++t) first[(size_t)t + 1] += first[(size_t)t];
    if (first == c->bb_first_host) return 0;
    c->bb_first_host.clear();
    if (first.size() > c->bb_first_cap) { dfree(c->d_bb_first); HIP_TRY(c, hipMalloc(&c->d_bb_first, first.size() * sizeof(int32_t))); c->bb_first_cap = first.size(); }
    HIP_TRY(c, hipStreamSynchronize(c->stream));                                                                       // (a launch in flight may still read the old table)
    HIP_TRY(c, hipMemcpy(c->d_bb_first, first.data(), first.size() * sizeof(int32_t), hipMemcpyHostToDevice));     // (synchronous: `first` is a local)
    c->bb_first_host = first;
    return 0;
}

// statistics of per-step records along the final particles' lineages against the final weights (lineage_stats_kernel)
template <class Col>
static int lineage_stats(cpprob_hip_ctx* c, const int32_t* d_anc, const int32_t* d_resampled, int32_t T, size_t n, const typename Col::value_t* d_cols, const int32_t* h_gen,
                         int32_t H, const double* d_logw, double* h_raw /*[H][kStats]*/, double* h_lse_ess)
{
    constexpr int kChunk = 64;
    if (int rc = upload_first_rows(c, h_gen, H, T)) return rc;
    if (int rc = bb_normalise(c, d_logw, n, (double)n)) return rc;
    const int nb = (int)((n + kTile - 1) / kTile);
    const int grid = std::min(nb, 2048);
    const int chunk = std::min<int>(H, kChunk);
    const size_t part_doubles = (size_t)chunk * Col::kStats * (size_t)grid, stat_doubles = (size_t)chunk * Col::kStats;
    if (part_doubles > c->bb_cols_part) { dfree(c->d_bb_cols_part); HIP_TRY(c, hipMalloc(&c->d_bb_cols_part, part_doubles * sizeof(double))); c->bb_cols_part = part_doubles; }
    if (stat_doubles > c->bb_cols_stat) { dfree(c->d_bb_cols_stat); HIP_TRY(c, hipMalloc(&c->d_bb_cols_stat, stat_doubles * sizeof(double))); c->bb_cols_stat = stat_doubles; }
    const size_t stats_bytes = (size_t)H * Col::kStats * sizeof(double), ride_off = (stats_bytes + sizeof(StepCtrl) + 15) / 16 * 16;
    if (int rc = pin_reserve(c, ride_off + c->ride_bytes)) return rc;
    for (int h0 = 0; h0 < H; h0 += chunk) {
        const int nk = std::min(chunk, H - h0);
        LineageStatsArgs<Col> a{};
        a.anc = d_anc; a.resampled = d_resampled; a.T = T; a.n = (int64_t)n; a.cols = d_cols; a.first_row = c->d_bb_first; a.h0 = h0; a.h1 = h0 + nk;
        a.wrel = c->d_bb_wrel; a.bf = c->d_bb_bf; a.ctrl = c->d_bb_ctrl; a.stats_part = c->d_bb_cols_part;
        hipLaunchKernelGGL(lineage_stats_kernel<Col>, dim3(grid), dim3(kThreads), (size_t)kWaves * nk * Col::kStats * sizeof(double), c->stream, a);
        hipLaunchKernelGGL(finalize_kernel, dim3((unsigned)nk), dim3(kThreads), 0, c->stream, c->d_bb_cols_part, grid, nk, Col::kStats, 1, c->d_bb_ctrl, c->d_bb_cols_stat, 1);
        HIP_TRY(c, hipGetLastError());
        PinPack pk;
        pk.add(c->d_bb_cols_stat, pin_data(c) + (size_t)h0 * Col::kStats * sizeof(double), (size_t)nk * Col::kStats * sizeof(double));
        if (h0 + chunk >= H) {                                              // (the last chunk's launch carries the rest)
            pk.add(c->d_bb_ctrl, pin_data(c) + stats_bytes, sizeof(StepCtrl));
            pk.add(c->ride_src, pin_data(c) + ride_off, c->ride_bytes);
        }
        if (int rc = pk.launch(c)) return rc;
        if (h0 + chunk < H) HIP_TRY(c, hipStreamSynchronize(c->stream));     // (the scratch is reused by the next chunk)
    }
    const bool riding = c->ride_bytes != 0;
    if (int rc = pin_wait(c)) return rc;
    StepCtrl h;
    std::memcpy(h_raw, pin_data(c), stats_bytes);
    std::memcpy(&h, pin_data(c) + stats_bytes, sizeof h);
    if (riding) { std::memcpy(c->ride_dst, pin_data(c) + ride_off, c->ride_bytes); c->ride_src = nullptr; c->ride_dst = nullptr; c->ride_bytes = 0; }
    h_lse_ess[0] = h.M + std::log(h.W); h_lse_ess[1] = h.ess;
    return 0;
}

// A read-back armed by cpprob_hip_readback_with_next_result rides the NEXT statistics call and no other: whatever way that call ends
// (bad arguments, a failed allocation, a device error), the caller's destination pointer does not outlive it in the context.
struct RideDisarm {
    cpprob_hip_ctx* c;
    ~RideDisarm() { if (c) { c->ride_src = nullptr; c->ride_dst = nullptr; c->ride_bytes = 0; } }
};

static int lineage_args_ok(cpprob_hip_ctx* c, const void* d_anc, const void* d_resampled, int32_t T, size_t n, const void* d_cols, const int32_t* h_gen, int32_t H,
                           const void* d_logw, const void* h_out)
{
    if (!d_anc || !d_resampled || !d_cols || !h_gen || !d_logw || !h_out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (T < 1 || H < 1 || n == 0) return fail(c, CPPROB_HIP_EINVAL, "need T >= 1, H >= 1 and a non-empty population");
    if (n > (size_t)INT32_MAX - kTile) return fail(c, CPPROB_HIP_EINVAL, "population too large for int32 ancestors");
    return 0;
}

extern "C" {

int cpprob_hip_readback_with_next_result(cpprob_hip_ctx* c, const void* d_src, void* h_dst, size_t bytes)
{
    BB_PRELUDE(c);
    if (bytes != 0 && (!d_src || !h_dst)) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (bytes > ((size_t)1 << 20) || (bytes & 3)) return fail(c, CPPROB_HIP_EINVAL, "a read-back that rides a result's is a small one (<= 1 MiB, whole 4-byte words)");
    c->ride_src = bytes ? d_src : nullptr; c->ride_dst = bytes ? h_dst : nullptr; c->ride_bytes = bytes;
    return 0;
}

int cpprob_hip_lineage_prepare(cpprob_hip_ctx* c, const int32_t* h_gen, int32_t H, int32_t T)
{
    BB_PRELUDE(c);
    if (!h_gen || T < 1 || H < 1) return fail(c, CPPROB_HIP_EINVAL, "need h_gen, T >= 1 and H >= 1");
    return upload_first_rows(c, h_gen, H, T);
}

int cpprob_hip_lineage_moments(cpprob_hip_ctx* c, const int32_t* d_anc, const int32_t* d_resampled, int32_t T, size_t n, const double* d_cols, const int32_t* h_gen, int32_t H,
                               const double* d_logw, double* h_out4)
{
    RideDisarm ride_guard{c};
    BB_PRELUDE(c);
    if (int rc = lineage_args_ok(c, d_anc, d_resampled, T, n, d_cols, h_gen, H, d_logw, h_out4)) return rc;
    std::vector<double> raw((size_t)H * 2);
    double le[2];
    if (int rc = lineage_stats<ColumnReal>(c, d_anc, d_resampled, T, n, d_cols, h_gen, H, d_logw, raw.data(), le)) return rc;
    for (int32_t k = 0; k < H; ++k) {
        h_out4[4 * k] = raw[2 * k];
        h_out4[4 * k + 1] = raw[2 * k + 1] - raw[2 * k] * raw[2 * k];     // variance(mean) = raw_moment(2) - mean*mean  (stats_printer.hpp:78-81)
        h_out4[4 * k + 2] = le[0]; h_out4[4 * k + 3] = le[1];
    }
    return 0;
}

int cpprob_hip_lineage_hist(cpprob_hip_ctx* c, const int32_t* d_anc, const int32_t* d_resampled, int32_t T, size_t n, const int32_t* d_cols, const int32_t* h_gen, int32_t H,
                            const double* d_logw, int32_t k, double* h_out, double* h_lse_ess)
{
    RideDisarm ride_guard{c};
    BB_PRELUDE(c);
    if (int rc = lineage_args_ok(c, d_anc, d_resampled, T, n, d_cols, h_gen, H, d_logw, h_out)) return rc;
    if (k < 1 || k > 8) return fail(c, CPPROB_HIP_EINVAL, "need 1 <= k <= 8");
    std::vector<double> raw((size_t)H * 8);
    double le[2];
    if (int rc = lineage_stats<ColumnInt8>(c, d_anc, d_resampled, T, n, d_cols, h_gen, H, d_logw, raw.data(), le)) return rc;
    for (int32_t j = 0; j < H; ++j)
        for (int s2 = 0; s2 < k; ++s2) h_out[(size_t)j * (size_t)k + s2] = raw[(size_t)j * 8 + s2];
    if (h_lse_ess) { h_lse_ess[0] = le[0]; h_lse_ess[1] = le[1]; }
    return 0;
}

int cpprob_hip_weighted_moments_columns(cpprob_hip_ctx* c, const double* d_x, size_t n_cols, size_t col_stride, const double* d_logw, size_t n, double* h_out4)
{
    RideDisarm ride_guard{c};
    BB_PRELUDE(c);
    if (!d_x || !d_logw || !h_out4) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (n == 0 || n_cols == 0 || col_stride < n) return fail(c, CPPROB_HIP_EINVAL, "empty distribution / columns closer than their length");
    std::vector<double> raw(n_cols * 2);
    double le[2];
    if (int rc = columns_stats<ColumnReal>(c, d_x, n_cols, col_stride, d_logw, n, raw.data(), le)) return rc;
    for (size_t k = 0; k < n_cols; ++k) {
        h_out4[4 * k] = raw[2 * k];
        h_out4[4 * k + 1] = raw[2 * k + 1] - raw[2 * k] * raw[2 * k];     // variance(mean) = raw_moment(2) - mean*mean  (:78-81)
        h_out4[4 * k + 2] = le[0]; h_out4[4 * k + 3] = le[1];
    }
    return 0;
}

int cpprob_hip_weighted_hist_columns(cpprob_hip_ctx* c, const int32_t* d_x, size_t n_cols, size_t col_stride, const double* d_logw, size_t n, int32_t k, double* h_out,
                                     double* h_lse_ess)
{
    RideDisarm ride_guard{c};
    BB_PRELUDE(c);
    if (!d_x || !d_logw || !h_out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (k < 1 || k > 8) return fail(c, CPPROB_HIP_EINVAL, "need 1 <= k <= 8");
    if (n == 0 || n_cols == 0 || col_stride < n) return fail(c, CPPROB_HIP_EINVAL, "empty distribution / columns closer than their length");
    std::vector<double> raw(n_cols * 8);
    double le[2];
    if (int rc = columns_stats<ColumnInt8>(c, d_x, n_cols, col_stride, d_logw, n, raw.data(), le)) return rc;
    for (size_t j = 0; j < n_cols; ++j)
        for (int s2 = 0; s2 < k; ++s2) h_out[j * (size_t)k + s2] = raw[j * 8 + s2];
    if (h_lse_ess) { h_lse_ess[0] = le[0]; h_lse_ess[1] = le[1]; }
    return 0;
}

int cpprob_hip_weighted_moments(cpprob_hip_ctx* c, const double* d_x, const double* d_logw, size_t n, double* h_out4)
{
    BB_PRELUDE(c);
    if (!d_x || !d_logw || !h_out4) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (n == 0) return fail(c, CPPROB_HIP_EINVAL, "empty distribution");
    double raw[2], le[2];
    if (int rc = column_stats<ColumnReal>(c, d_x, d_logw, n, raw, le)) return rc;
    h_out4[0] = raw[0];
    h_out4[1] = raw[1] - raw[0] * raw[0];     // variance(mean) = raw_moment(2) - mean*mean  (:78-81)
    h_out4[2] = le[0]; h_out4[3] = le[1];
    return 0;
}

int cpprob_hip_weighted_hist(cpprob_hip_ctx* c, const int32_t* d_x, const double* d_logw, size_t n, int32_t k, double* h_out)
{
    BB_PRELUDE(c);
    if (!d_x || !d_logw || !h_out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (k < 1 || k > 8) return fail(c, CPPROB_HIP_EINVAL, "need 1 <= k <= 8");
    if (n == 0) return fail(c, CPPROB_HIP_EINVAL, "empty distribution");
    double raw[8], le[2];
    if (int rc = column_stats<ColumnInt8>(c, d_x, d_logw, n, raw, le)) return rc;
    for (int s = 0; s < k; ++s) h_out[s] = raw[s];
    return 0;
}

int cpprob_hip_resample(cpprob_hip_ctx* c, int32_t kind, const double* d_logw, size_t n_in, uint64_t seed, uint64_t step, uint64_t j0, size_t n_out,
                        uint64_t n_total_out, int32_t* d_anc)
{
    BB_PRELUDE(c);
    if (!d_logw || !d_anc) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (n_in == 0) return fail(c, CPPROB_HIP_EINVAL, "cannot resample an empty population");
    if (n_in > (size_t)INT32_MAX - kTile) return fail(c, CPPROB_HIP_EINVAL, "population too large for int32 ancestors");
    if (n_total_out == 0 || j0 + n_out > n_total_out) return fail(c, CPPROB_HIP_EINVAL, "output range outside [0, n_total_out)");
    if (n_out == 0) return 0;
    if (int rc = bb_normalise(c, d_logw, n_in, (double)n_in)) return rc;
    const int nb_in = (int)((n_in + kTile - 1) / kTile);
    if (kind == CPPROB_HIP_RESAMPLE_MULTINOMIAL) {
        if ((size_t)nb_in * kTile > c->bb_cdf_cap) { dfree(c->d_bb_cdf); HIP_TRY(c, hipMalloc(&c->d_bb_cdf, (size_t)nb_in * kTile * sizeof(double))); c->bb_cdf_cap = (size_t)nb_in * kTile; }
        hipLaunchKernelGGL(cdf_kernel, dim3(nb_in), dim3(kThreads), 0, c->stream, c->d_bb_wrel, c->d_bb_bc, c->d_bb_bf, c->d_bb_ctrl, c->d_bb_cdf);
        hipLaunchKernelGGL(multinomial_kernel, GRID1(n_out), c->d_bb_cdf, (int64_t)n_in, c->d_bb_ctrl, seed, step, j0, (int64_t)n_out, (int64_t)n_out, d_anc, 0);
    } else if (kind == CPPROB_HIP_RESAMPLE_SYSTEMATIC || kind == CPPROB_HIP_RESAMPLE_STRATIFIED) {
        ResampleArgs a{};
        a.wrel = c->d_bb_wrel; a.n_in = (int64_t)n_in; a.bc = c->d_bb_bc; a.bf = c->d_bb_bf; a.nb = nb_in; a.ctrl = c->d_bb_ctrl; a.seed = seed; a.step = step; a.j0 = j0;
        a.n_total_out = n_total_out; a.n_out = (int64_t)n_out; a.anc = d_anc;
        const int nb_out = (int)((n_out + kTile - 1) / kTile);
        ProfScope ps(c, 5);
        if (kind == CPPROB_HIP_RESAMPLE_SYSTEMATIC) hipLaunchKernelGGL(resample_kernel<RS_SYSTEMATIC>, dim3(nb_out), dim3(kThreads), 0, c->stream, a);
        else hipLaunchKernelGGL(resample_kernel<RS_STRATIFIED>, dim3(nb_out), dim3(kThreads), 0, c->stream, a);
    } else {
        return fail(c, CPPROB_HIP_EINVAL, "unknown resampler");
    }
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_smc_bookkeep(cpprob_hip_ctx* c, int32_t kind, const double* d_logw, size_t n, uint64_t seed, int32_t step, int32_t last, double ess_frac,
                            double* d_ess, int32_t* d_resampled, double* d_log_z, int32_t* d_anc)
{
    BB_PRELUDE(c);
    if (!d_logw || !d_ess || !d_resampled || !d_log_z || !d_anc) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (n == 0 || n > (size_t)INT32_MAX - kTile) return fail(c, CPPROB_HIP_EINVAL, "population size out of range");
    if (kind < CPPROB_HIP_RESAMPLE_SYSTEMATIC || kind > CPPROB_HIP_RESAMPLE_MULTINOMIAL) return fail(c, CPPROB_HIP_EINVAL, "unknown resampler");
    if (step < 0) return fail(c, CPPROB_HIP_EINVAL, "step out of range");
    if (int rc = ensure_bb(c, n)) return rc;
    const int nb = (int)((n + kTile - 1) / kTile);
    hipLaunchKernelGGL(weights_partials_kernel, dim3(nb), dim3(kThreads), 0, c->stream, d_logw, (int64_t)n, c->d_bb_part, c->d_bb_wrel);
    ScanArgs sa{};
    sa.part = c->d_bb_part; sa.nb = nb; sa.bc = c->d_bb_bc; sa.bf = c->d_bb_bf; sa.ctrl = c->d_bb_ctrl; sa.t = step; sa.T = last ? step + 1 : step + 2;
    sa.n_pop = (double)n; sa.n_local = (double)n; sa.ess_frac = ess_frac; sa.force_no_resample = 0; sa.phase = 0; sa.seed = seed;
    sa.ess_trace = d_ess; sa.resampled = d_resampled; sa.log_z_out = d_log_z;
    hipLaunchKernelGGL(scan_partials_kernel, dim3(1), dim3(kScanThreads), 0, c->stream, sa);
    if (!last) {
        if (kind == CPPROB_HIP_RESAMPLE_MULTINOMIAL) {
            if ((size_t)nb * kTile > c->bb_cdf_cap) { dfree(c->d_bb_cdf); HIP_TRY(c, hipMalloc(&c->d_bb_cdf, (size_t)nb * kTile * sizeof(double))); c->bb_cdf_cap = (size_t)nb * kTile; }
            hipLaunchKernelGGL(cdf_kernel, dim3(nb), dim3(kThreads), 0, c->stream, c->d_bb_wrel, c->d_bb_bc, c->d_bb_bf, c->d_bb_ctrl, c->d_bb_cdf);
            hipLaunchKernelGGL(multinomial_kernel, GRID1(n), c->d_bb_cdf, (int64_t)n, c->d_bb_ctrl, seed, (uint64_t)step + 1, (uint64_t)0, (int64_t)n, (int64_t)n, d_anc, 1);
        } else {
            ResampleArgs a{};
            a.wrel = c->d_bb_wrel; a.n_in = (int64_t)n; a.bc = c->d_bb_bc; a.bf = c->d_bb_bf; a.nb = nb; a.ctrl = c->d_bb_ctrl; a.seed = seed; a.step = (uint64_t)step + 1;
            a.j0 = 0; a.n_total_out = n; a.n_out = (int64_t)n; a.anc = d_anc; a.run_ctrl = 1; a.identity_unless_resampling = 1;
            if (kind == CPPROB_HIP_RESAMPLE_SYSTEMATIC) hipLaunchKernelGGL(resample_kernel<RS_SYSTEMATIC>, dim3(nb), dim3(kThreads), 0, c->stream, a);
            else hipLaunchKernelGGL(resample_kernel<RS_STRATIFIED>, dim3(nb), dim3(kThreads), 0, c->stream, a);
        }
    }
    HIP_TRY(c, hipGetLastError());
    return 0;
}

}  // extern "C"
namespace {
// (re)lays the two-copy hierarchy out for populations of nb tiles
int ensure_bbf(cpprob_hip_ctx* c, int nb)
{
    if (nb == c->bbf_nb && c->d_bbf_hier) return 0;
    size_t per_copy = 0, off[kHierMaxLevels] = {0, 0, 0};
    int nl = 0;
    HierTable t{};
    for (size_t e = (size_t)nb;; e = (e + 63) / 64) {
        if (nl >= kHierMaxLevels) return fail(c, CPPROB_HIP_EUNSUPPORTED, "population too large for the three-level mass hierarchy");
        off[nl] = per_copy; t.n_ent[nl] = (int)e; per_copy += e * (nl == 0 ? 1 : kHierStride); ++nl;
        if (e <= 64) break;
    }
    t.n_lev = nl;
    c->bbf_q0_off = per_copy; c->bbf_m0_off = per_copy + (size_t)nb; per_copy += 2 * (size_t)nb;
    if ((size_t)nb > c->bbf_cap_nb || !c->d_bbf_hier) {
        dfree(c->d_bbf_hier); dfree(c->d_bbf_q);
        HIP_TRY(c, hipMalloc(&c->d_bbf_hier, 2 * per_copy * sizeof(uint64_t)));
        HIP_TRY(c, hipMalloc(&c->d_bbf_q, (size_t)nb * kTile * sizeof(uint32_t)));
        c->bbf_cap_nb = (size_t)nb;
    }
    for (int k = 0; k < 3; ++k)
        for (int l = 0; l < nl; ++l) t.lvl[k][l] = c->d_bbf_hier + (size_t)(k & 1) * per_copy + off[l];
    if (!c->d_bbf_table) HIP_TRY(c, hipMalloc(&c->d_bbf_table, sizeof(HierTable)));
    HIP_TRY(c, hipMemcpyAsync(c->d_bbf_table, &t, sizeof t, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_bbf_hier, 0, 2 * per_copy * sizeof(uint64_t), c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));               // (t is a local)
    c->bbf_table = t; c->bbf_per_copy = per_copy; c->bbf_nb = nb; c->bbf_phase = 0;
    return 0;
}
void bbf_view(const cpprob_hip_ctx* c, int copy, FHier& f)
{
    const HierTable& t = c->bbf_table;
    for (int l = 0; l < kHierMaxLevels; ++l) { f.h.lvl[l] = l < t.n_lev ? t.lvl[copy][l] : t.lvl[copy][0]; f.h.n_ent[l] = t.n_ent[l]; }
    f.h.n_lev = t.n_lev; f.h.table = c->d_bbf_table; f.h.copy = copy;
    const int top = t.n_lev - 1;
    f.h.top = t.lvl[copy][top]; f.h.top_n = t.n_ent[top]; f.h.top_stride = top == 0 ? 1 : kHierStride;
    f.h.to_next = 0; f.h.to_clear = (int64_t)((copy ^ 1) - copy) * (int64_t)c->bbf_per_copy;
    f.q0 = c->d_bbf_hier + (size_t)copy * c->bbf_per_copy + c->bbf_q0_off;
    f.m0 = c->d_bbf_hier + (size_t)copy * c->bbf_per_copy + c->bbf_m0_off;
}
}  // namespace
extern "C" {

int cpprob_hip_smc_bookkeep_fixed(cpprob_hip_ctx* c, const double* d_logw, size_t n, uint64_t seed, int32_t step, int32_t last, double ess_frac,
                                  double* d_ess, int32_t* d_resampled, double* d_log_z, int32_t* d_anc)
{
    return cpprob_hip_smc_bookkeep_fixed_rs(c, CPPROB_HIP_RESAMPLE_SYSTEMATIC, d_logw, n, seed, step, last, ess_frac, d_ess, d_resampled, d_log_z, d_anc);
}

int cpprob_hip_smc_bookkeep_fixed_rs(cpprob_hip_ctx* c, int32_t kind, const double* d_logw, size_t n, uint64_t seed, int32_t step, int32_t last, double ess_frac,
                                     double* d_ess, int32_t* d_resampled, double* d_log_z, int32_t* d_anc)
{
    BB_PRELUDE(c);
    if (kind < CPPROB_HIP_RESAMPLE_SYSTEMATIC || kind > CPPROB_HIP_RESAMPLE_MULTINOMIAL) return fail(c, CPPROB_HIP_EINVAL, "unknown resampler");
    if (!d_logw || !d_ess || !d_resampled || !d_log_z || (!last && !d_anc)) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (n == 0 || n > (size_t)(1ull << 28)) return fail(c, CPPROB_HIP_EINVAL, "population size out of range (1 .. 2^28: the squares' 64-bit sum)");
    if (step < 0) return fail(c, CPPROB_HIP_EINVAL, "step out of range");
    const int nb = (int)((n + kTile - 1) / kTile);
    if (int rc = ensure_bbf(c, nb)) return rc;
    FHier f{};
    bbf_view(c, c->bbf_phase, f);
    c->bbf_phase ^= 1;
    hipLaunchKernelGGL(bbf_max_kernel, dim3(nb), dim3(kThreads), 0, c->stream, d_logw, (int64_t)n, f);
    hipLaunchKernelGGL(bbf_quantize_kernel, dim3(nb), dim3(kThreads), 0, c->stream, d_logw, (int64_t)n, f, c->d_bbf_q);
    BbfArgs a{};
    a.f = f; a.q = c->d_bbf_q; a.n = (int64_t)n; a.nb = nb; a.u0 = host_resample_u0(seed, (uint64_t)step + 1); a.ess_frac = ess_frac; a.step = step; a.last = last ? 1 : 0;
    a.ess = d_ess; a.resampled = d_resampled; a.log_z = d_log_z; a.anc = d_anc; a.seed = seed;
    if (kind == CPPROB_HIP_RESAMPLE_MULTINOMIAL && !last) {
        // strata form: the counts of this one resampling (they do not depend on the weights: one short launch, two above 64 tiles)
        const int k = strata_levels(nb);
        const size_t words = ((size_t)1 << k) + 1;
        if (words > c->bbf_strata_cap) {
            dfree(c->d_bbf_strata); dfree(c->d_bbf_strata_top);
            HIP_TRY(c, hipMalloc(&c->d_bbf_strata, words * sizeof(uint32_t)));
            HIP_TRY(c, hipMalloc(&c->d_bbf_strata_top, 64 * sizeof(uint32_t)));
            c->bbf_strata_cap = words;
        }
        StrataArgs sa{};
        sa.seed = seed; sa.t0 = step + 1; sa.k = k; sa.n_out = (uint32_t)n; sa.offs = c->d_bbf_strata;
        if (k <= kStrataTop) hipLaunchKernelGGL(multinomial_strata_kernel, dim3(1, 1), dim3(kThreads), 0, c->stream, sa);
        else {
            HIP_TRY(c, hipMemsetAsync(c->d_bbf_strata_top, 0, 64 * sizeof(uint32_t), c->stream));
            sa.top = c->d_bbf_strata_top; sa.top_clear = nullptr;
            hipLaunchKernelGGL(multinomial_strata_top_kernel, dim3(strata_groups(k), 1), dim3(kThreads), 0, c->stream, sa);
            hipLaunchKernelGGL(multinomial_strata_bottom_kernel, dim3(64, 1), dim3(kThreads), 0, c->stream, sa);
        }
        a.strata_offs = c->d_bbf_strata; a.strata_k = k;
    }
    const dim3 grid(last ? 1 : nb);
    if (kind == CPPROB_HIP_RESAMPLE_STRATIFIED) hipLaunchKernelGGL(bbf_ancestors_kernel<kFixStratified>, grid, dim3(kThreads), 0, c->stream, a);
    else if (kind == CPPROB_HIP_RESAMPLE_MULTINOMIAL) hipLaunchKernelGGL(bbf_ancestors_kernel<kFixMultinomial>, grid, dim3(kThreads), 0, c->stream, a);
    else hipLaunchKernelGGL(bbf_ancestors_kernel<kFixSystematic>, grid, dim3(kThreads), 0, c->stream, a);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

// ---- the unchanged-model step with the resampling inside the model's launch (include/cpprob_hip.h; cpprob/gpu.hpp) ----
}  // extern "C"
namespace {
struct GenericCtrlBlock { double ref_cur, gap_max; };      // (= cpprob::device::StepCtrl2)

// Exact-reference form of the unchanged-model step: the hierarchy's entries are 256-particle BLOCKS (one per workgroup of the model's
// step kernel); these two passes run a 1024-particle tile per workgroup, so every wavefront owns one block and publishes it itself.
constexpr int kGenBlock = 256;
__global__ __launch_bounds__(kThreads) void generic_max_kernel(const double* __restrict__ logw, int64_t n, FHier f, int n_blocks)
{
    const int64_t j0 = (int64_t)blockIdx.x * kTile + (int64_t)threadIdx.x * kPPT;
    double m = -INFINITY;
#pragma unroll
    for (int k = 0; k < kPPT; ++k) if (j0 + k < n) m = fmax(m, logw[j0 + k]);
    const uint64_t mw = wave_max_u64(dkey(m));
    const int blk = (int)blockIdx.x * kWaves + wave_id();
    if (lane_id() == 0 && blk < n_blocks) bbf_publish_max(f, blk, n_blocks, mw);
}

__global__ __launch_bounds__(kThreads) void generic_quantize_kernel(const double* __restrict__ logw, int64_t n, FHier f, int n_blocks, uint32_t* __restrict__ q, GenericCtrlBlock* ctrl,
                                                                int given, double ref_given)
{
    const double ref = given ? ref_given : bbf_top_max(f);              // (every wavefront: the same <= 64 words)
    if (blockIdx.x == 0 && threadIdx.x == 0) ctrl->ref_cur = ref;
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
    const int blk = (int)blockIdx.x * kWaves + wave_id();
    if (lane_id() == 0 && blk < n_blocks) bbf_publish_mass(f, blk, n_blocks, sw, qw);
}

__global__ __launch_bounds__(kWave) void generic_finish_kernel(FHier f, int T, double n_pop, double gap_limit, GenericCtrlBlock* ctrl, double* ess, int32_t* resampled,
                                                               double* log_z, int32_t* flags)
{
    const FTot t = ftot(f);
    if (threadIdx.x != 0) return;
    const FixedDecision d = fixed_decide(t.S, t.Q, n_pop, 0.0, false);
    const double ref = ctrl->ref_cur;
    const double gap = d.W > 0.0 ? ref - t.M : 1e300;
    ctrl->gap_max = T == 1 ? gap : fmax(ctrl->gap_max, gap);
    if (gap < 0.0) *flags = 4;
    else if (gap > gap_limit && *flags == 0) *flags = 5;
    ess[T - 1] = d.ess;
    resampled[T - 1] = 0;
    const double lz = T == 1 ? 0.0 : *log_z;
    *log_z = lz + (ref + log(d.W / n_pop));
}

// (re)lays the three-copy hierarchy out for populations of nb 256-particle blocks
int ensure_generic(cpprob_hip_ctx* c, int nb, int block)
{
    if (nb != c->gen_nb || block != c->gen_block || !c->d_gen_hier) {
        size_t per_copy = 0, off[kHierMaxLevels] = {0, 0, 0};
        int nl = 0;
        HierTable t{};
        for (size_t e = (size_t)nb;; e = (e + 63) / 64) {
            if (nl >= kHierMaxLevels) return fail(c, CPPROB_HIP_EUNSUPPORTED, "population too large for the three-level mass hierarchy");
            off[nl] = per_copy; t.n_ent[nl] = (int)e; per_copy += e * (nl == 0 ? 1 : kHierStride); ++nl;
            if (e <= 64) break;
        }
        t.n_lev = nl;
        c->gen_q0_off = per_copy; c->gen_m0_off = per_copy + (size_t)nb; per_copy += 2 * (size_t)nb;
        // (whole tiles, and one more: the step kernel's walk reads 1024-particle chunks that start at any block)
        const size_t q_words = ((size_t)nb * (size_t)block + kTile - 1) / kTile * kTile + kTile;
        if (3 * per_copy > c->gen_cap_hier || q_words > c->gen_cap_q || !c->d_gen_hier) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            dfree(c->d_gen_hier); dfree(c->d_gen_q[0]); dfree(c->d_gen_q[1]);
            c->gen_cap_hier = 0; c->gen_cap_q = 0; c->gen_nb = 0;
            HIP_TRY(c, hipMalloc(&c->d_gen_hier, 3 * per_copy * sizeof(uint64_t)));
            for (int k = 0; k < 2; ++k) { HIP_TRY(c, hipMalloc(&c->d_gen_q[k], q_words * sizeof(uint32_t))); HIP_TRY(c, hipMemsetAsync(c->d_gen_q[k], 0, q_words * sizeof(uint32_t), c->stream)); }
            c->gen_cap_hier = 3 * per_copy; c->gen_cap_q = q_words;
        }
        for (int k = 0; k < 3; ++k)
            for (int l = 0; l < kHierMaxLevels; ++l) t.lvl[k][l] = c->d_gen_hier + (size_t)k * per_copy + off[l < nl ? l : 0];
        if (!c->d_gen_table) HIP_TRY(c, hipMalloc(&c->d_gen_table, sizeof(HierTable)));
        if (!c->d_gen_ctrl) HIP_TRY(c, hipMalloc(&c->d_gen_ctrl, sizeof(GenericCtrlBlock)));
        HIP_TRY(c, hipMemcpyAsync(c->d_gen_table, &t, sizeof t, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));               // (t is a local)
        c->gen_table = t; c->gen_per_copy = per_copy; c->gen_nb = nb; c->gen_block = block;
        for (int l = 0; l < kHierMaxLevels; ++l) c->gen_off[l] = off[l];
    }
    return 0;
}
// copy `read` as the launch's view; the copy written sits `to_next` words further, the copy cleared `to_clear`
void generic_view(const cpprob_hip_ctx* c, int read, int next, int clear, FHier& f)
{
    const HierTable& t = c->gen_table;
    for (int l = 0; l < kHierMaxLevels; ++l) { f.h.lvl[l] = t.lvl[read][l]; f.h.n_ent[l] = t.n_ent[l]; }
    f.h.n_lev = t.n_lev; f.h.table = c->d_gen_table; f.h.copy = read;
    const int top = t.n_lev - 1;
    f.h.top = t.lvl[read][top]; f.h.top_n = t.n_ent[top]; f.h.top_stride = top == 0 ? 1 : kHierStride;
    f.h.to_next = (int64_t)(next - read) * (int64_t)c->gen_per_copy; f.h.to_clear = (int64_t)(clear - read) * (int64_t)c->gen_per_copy;
    f.q0 = c->d_gen_hier + (size_t)read * c->gen_per_copy + c->gen_q0_off;
    f.m0 = c->d_gen_hier + (size_t)read * c->gen_per_copy + c->gen_m0_off;
}
}  // namespace
extern "C" {

static int generic_begin(cpprob_hip_ctx* c, size_t n, int block, cpprob_hip_generic_layout* out)
{
    if (!out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (block != kGenBlock && block != kTile) return fail(c, CPPROB_HIP_EINVAL, "the hierarchy's entries are blocks of 256 particles or tiles of 1024");
    if (n == 0 || n > (size_t)(1ull << 28)) return fail(c, CPPROB_HIP_EINVAL, "population size out of range (1 .. 2^28: the squares' 64-bit sum)");
    if (n > (size_t)kCountsMaxTiles * (size_t)block) return fail(c, CPPROB_HIP_EINVAL, "population too large for the three-level mass hierarchy (64^3 entries)");
    const int nb = (int)((n + (size_t)block - 1) / (size_t)block);
    if (int rc = ensure_generic(c, nb, block)) return rc;
    // a run starts from clean upper levels in every copy
    HIP_TRY(c, hipMemsetAsync(c->d_gen_hier, 0, 3 * c->gen_per_copy * sizeof(uint64_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_gen_ctrl, 0, sizeof(GenericCtrlBlock), c->stream));
    std::memset(out, 0, sizeof *out);
    out->hier = c->d_gen_hier; out->per_copy = c->gen_per_copy;
    for (int l = 0; l < kHierMaxLevels; ++l) { out->lvl_off[l] = c->gen_off[l < c->gen_table.n_lev ? l : 0]; out->n_ent[l] = c->gen_table.n_ent[l]; }
    out->n_lev = c->gen_table.n_lev; out->q0_off = c->gen_q0_off; out->m0_off = c->gen_m0_off;
    out->table = c->d_gen_table; out->q[0] = c->d_gen_q[0]; out->q[1] = c->d_gen_q[1]; out->ctrl = c->d_gen_ctrl; out->blocks = nb; out->block = block;
    return 0;
}
int cpprob_hip_generic_begin(cpprob_hip_ctx* c, size_t n, cpprob_hip_generic_layout* out)
{
    BB_PRELUDE(c);
    return generic_begin(c, n, kGenBlock, out);
}
int cpprob_hip_generic_begin_tiles(cpprob_hip_ctx* c, size_t n, cpprob_hip_generic_layout* out)
{
    BB_PRELUDE(c);
    return generic_begin(c, n, kTile, out);
}

static int generic_exact_passes(cpprob_hip_ctx* c, int32_t t, const double* d_logw, size_t n, bool max_pass, bool mass_pass, int given, double ref)
{
    if (!d_logw || t < 0) return fail(c, CPPROB_HIP_EINVAL, "bad argument");
    const int nb = (int)((n + kGenBlock - 1) / kGenBlock), nt = (int)((n + kTile - 1) / kTile);
    if (nb != c->gen_nb || c->gen_block != kGenBlock || !c->d_gen_hier) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_generic_begin was not called for this population size");
    FHier f{};
    generic_view(c, t % 3, t % 3, (t + 1) % 3, f);
    if (max_pass) hipLaunchKernelGGL(generic_max_kernel, dim3(nt), dim3(kThreads), 0, c->stream, d_logw, (int64_t)n, f, nb);
    if (mass_pass) hipLaunchKernelGGL(generic_quantize_kernel, dim3(nt), dim3(kThreads), 0, c->stream, d_logw, (int64_t)n, f, nb, c->d_gen_q[t & 1], static_cast<GenericCtrlBlock*>(c->d_gen_ctrl), given, ref);
    HIP_TRY(c, hipGetLastError());
    return 0;
}
int cpprob_hip_generic_quantize(cpprob_hip_ctx* c, int32_t t, const double* d_logw, size_t n)
{
    BB_PRELUDE(c);
    return generic_exact_passes(c, t, d_logw, n, true, true, 0, 0.0);
}
int cpprob_hip_generic_max(cpprob_hip_ctx* c, int32_t t, const double* d_logw, size_t n)
{
    BB_PRELUDE(c);
    return generic_exact_passes(c, t, d_logw, n, true, false, 0, 0.0);
}
int cpprob_hip_generic_quantize_ref(cpprob_hip_ctx* c, int32_t t, const double* d_logw, size_t n, double ref)
{
    BB_PRELUDE(c);
    return generic_exact_passes(c, t, d_logw, n, false, true, 1, ref);
}
int cpprob_hip_generic_totals(cpprob_hip_ctx* c, int32_t t, size_t n, uint64_t* d_out3)
{
    BB_PRELUDE(c);
    if (!d_out3 || t < 0) return fail(c, CPPROB_HIP_EINVAL, "bad argument");
    const int nb = (int)((n + (size_t)c->gen_block - 1) / (size_t)c->gen_block);
    if (nb != c->gen_nb || !c->d_gen_hier) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_generic_begin was not called for this population size");
    FHier f{};
    generic_view(c, t % 3, t % 3, t % 3, f);
    hipLaunchKernelGGL(fixed_totals_kernel, dim3(1), dim3(kWave), 0, c->stream, f, d_out3);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_generic_finish(cpprob_hip_ctx* c, int32_t T, size_t n, double gap_limit, double* d_ess, int32_t* d_resampled, double* d_log_z, int32_t* d_flags)
{
    BB_PRELUDE(c);
    if (!d_ess || !d_resampled || !d_log_z || !d_flags || T < 1) return fail(c, CPPROB_HIP_EINVAL, "bad argument");
    const int nb = (int)((n + (size_t)c->gen_block - 1) / (size_t)c->gen_block);
    if (nb != c->gen_nb || !c->d_gen_hier) return fail(c, CPPROB_HIP_ESTATE, "cpprob_hip_generic_begin was not called for this population size");
    FHier f{};
    const int k = (T - 1) % 3;
    generic_view(c, k, k, k, f);
    hipLaunchKernelGGL(generic_finish_kernel, dim3(1), dim3(kWave), 0, c->stream, f, (int)T, (double)n, gap_limit, static_cast<GenericCtrlBlock*>(c->d_gen_ctrl), d_ess, d_resampled,
                       d_log_z, d_flags);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

double cpprob_hip_systematic_offset(uint64_t seed, uint64_t step) { return host_resample_u0(seed, step); }

int cpprob_hip_lineage_gather(cpprob_hip_ctx* c, const int32_t* d_anc, const int32_t* d_resampled, int32_t T, size_t n, const void* d_cols, int32_t is_int,
                              const int32_t* h_gen, int32_t H, void* d_out)
{
    BB_PRELUDE(c);
    if (!d_anc || !d_resampled || !d_cols || !h_gen || !d_out) return fail(c, CPPROB_HIP_EINVAL, "NULL argument");
    if (T < 1 || H < 0) return fail(c, CPPROB_HIP_EINVAL, "bad T / H");
    if (n == 0 || H == 0) return 0;
    if (int rc = upload_first_rows(c, h_gen, H, T)) return rc;
    int32_t* d_first = c->d_bb_first;
    if (is_int) hipLaunchKernelGGL(lineage_gather_kernel<int32_t>, GRID1(n), d_anc, d_resampled, (int)T, (int64_t)n, static_cast<const int32_t*>(d_cols), (const int32_t*)d_first, static_cast<int32_t*>(d_out));
    else hipLaunchKernelGGL(lineage_gather_kernel<double>, GRID1(n), d_anc, d_resampled, (int)T, (int64_t)n, static_cast<const double*>(d_cols), (const int32_t*)d_first, static_cast<double*>(d_out));
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_gather_f64(cpprob_hip_ctx* c, const double* src, const int32_t* idx, size_t n, double* dst)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather_kernel<double>, GRID1(n), src, idx, (int64_t)n, dst);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_gather_i32(cpprob_hip_ctx* c, const int32_t* src, const int32_t* idx, size_t n, int32_t* dst)
{
    BB_PRELUDE(c);
    if (n == 0) return 0;
    hipLaunchKernelGGL(gather_kernel<int32_t>, GRID1(n), src, idx, (int64_t)n, dst);
    HIP_TRY(c, hipGetLastError());
    return 0;
}

int cpprob_hip_profile_enable(cpprob_hip_ctx* c, int32_t on)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    c->profile = on != 0;
    return 0;
}

int cpprob_hip_profile_read(cpprob_hip_ctx* c, double* h_ms, int64_t* h_calls, int32_t reset)
{
    if (!c) return fail(nullptr, CPPROB_HIP_EINVAL, "ctx is NULL");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (auto& ep : c->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ep.a, ep.b) == hipSuccess) { c->prof_ms[ep.cls] += ms; c->prof_calls[ep.cls] += ep.count; }
        c->ev_free.push_back(ep);
    }
    c->ev_used.clear();
    for (int k = 0; k < CPPROB_HIP_N_KERNEL_CLASSES; ++k) {
        if (h_ms) h_ms[k] = c->prof_ms[k];
        if (h_calls) h_calls[k] = c->prof_calls[k];
        if (reset) { c->prof_ms[k] = 0; c->prof_calls[k] = 0; }
    }
    return 0;
}

}  // extern "C"

#include "group.hpp"
