// gpu.hpp -- turns an UNCHANGED CPProb model function into MI355X kernels (hipcc, C++17 only).
//
// A model translation unit looks like this:
//
//     #include "cpprob/gpu.hpp"
//     #pragma clang force_cuda_host_device begin
//     #include "models/models.hpp"                  // the reference's model source, untouched
//     #pragma clang force_cuda_host_device end
//     CPPROB_REGISTER_MODEL(models::hmm<16>);
//
// and from then on `cpprob::inference(StateType::sis | smc, models::hmm<16>, observes, n, file)` -- the
// reference's call, from any C++14 host code -- runs the model body on the GPU, one particle per lane.
//
//   SIS : one launch; every lane runs the model to completion (cpprob.hpp:194-201 for all i at once).
//   SMC : one launch per observe (trace replay): the lane re-runs the model from
//         the top, `sample` statements already executed by its ancestor return the stored values, the
//         next ones draw fresh, the step's observe adds the incremental weight and ends the lane.
//         Between launches one C-ABI building block (cpprob_hip_smc_bookkeep) normalises the weights, tests the ESS and
//         resamples ON THE DEVICE -- no host synchronisation inside a run; the stored sample values of the chosen
//         ancestors are carried along by the replay itself (each step rewrites the full trace), so the
//         final launch regenerates every predict of every surviving particle.
// Cost: O(T^2) statement executions for T observes -- the price of not editing the model; the built-in
// kernels (CPPROB_REGISTER_BUILTIN) are the O(T) fast path for the models they cover.
// Restriction: the number and order of observe / predict statements must not depend on sampled values.  The
// number of sample statements may (rejection-sampling loops): SIS runs them as they come; SMC keeps 4x the
// dry run's count of trace rows per particle and reports an error if a particle needs more.
#ifndef CPPROB_COMPAT_GPU_HPP
#define CPPROB_COMPAT_GPU_HPP
#if !defined(__HIPCC__) && !defined(__HIP__)
#error "cpprob/gpu.hpp needs hipcc (-std=c++17 --offload-arch=gfx950); host-only code includes cpprob/cpprob.hpp"
#endif

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <tuple>
#include <vector>

#include <boost/math/distributions/normal.hpp>

#include "cpprob/cpprob.hpp"
#include "cpprob/detail/device_vector.hpp"
#include "cpprob/distributions/multivariate_normal.hpp"
#include "cpprob/ndarray.hpp"

namespace cpprob {
namespace gpu {

using ModelKernelArgs = device::LaunchArgs;     // (cpprob/detail/device_trace.hpp: the statements read it from the kernel-argument segment)

// How the kernel reaches the model body: a function (by address, as a template argument) or a functor
// (by type, default-constructed on the lane -- e.g. the reference's models::Gauss<>, models.hpp:51-65).
template <class FP, FP F>
struct FunctionCaller {
    using observes_t = tuple_observes_t<FP>;
    CPPROB_HD static void call(const observes_t& obs) { call_f_tuple(F, obs); }
};
template <class Functor>
struct FunctorCaller {
    using observes_t = tuple_observes_t<Functor>;
    CPPROB_HD static void call(const observes_t& obs) { call_f_tuple(Functor{}, obs); }
};

template <class Caller, class Tuple>
__global__ __launch_bounds__(device::kLaneBlock) void model_kernel(ModelKernelArgs a, const Tuple* __restrict__ observes)
{
    const int64_t i = (int64_t)blockIdx.x * device::kLaneBlock + threadIdx.x;
    if (i >= a.n) return;
    const bool resampled = a.resampled_prev && *a.resampled_prev != 0;
    const int64_t src = (a.anc && resampled) ? (int64_t)a.anc[i] : i;
    const double carried = (!resampled && a.logw_in) ? a.logw_in[i] : 0.0;   // equal weights after resampling
    device::begin_lane((int32_t)src, a.nstored_in ? (uint32_t)a.nstored_in[src] : 0u, carried);
    Caller::call(*observes);                                      // the model body, cpprob.hpp:199
    device::finish_lane();                                        // finish_trace(): the particle's log_w_
}

template <class Tuple> struct observes_bytewise_copyable;
template <class... A> struct observes_bytewise_copyable<std::tuple<A...>>
{
    static constexpr bool value = (std::is_trivially_copyable<A>::value && ...);
};

template <class T>
struct DevBuf {
    T* p = nullptr;
    DevBuf() = default;
    explicit DevBuf(size_t count) { if (count && hipMalloc(&p, count * sizeof(T)) != hipSuccess) throw std::runtime_error("hipMalloc failed"); }
    ~DevBuf() { if (p) (void)hipFree(p); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

inline void hip_check(hipError_t e, const char* what)
{
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// One attempt with S trace rows per particle; returns 0, or -- nothing in res is valid then -- 1 when some particle needed more rows,
// 3 when the lanes of a wavefront executed different statements under windowed replay (the model's statement counts do depend on
// sampled values: the caller repeats the run with full replay).
template <class Caller>
int generic_attempt(StateType algorithm, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt,
                     Result& res, HostStore* store, const std::size_t S)
{
    using Tuple = typename Caller::observes_t;
    // (std::tuple itself is not trivially copyable in libstdc++ even when every element is: check the elements)
    static_assert(observes_bytewise_copyable<Tuple>::value,
                  "CPPROB_REGISTER_MODEL: the observes tuple is copied to the device bytewise, so every model argument must be trivially "
                  "copyable (arithmetic types, std::array of them); models with std::vector / NDArray arguments own host heap memory");
    Context ctx(opt.device);
    hip_check(hipSetDevice(opt.device), "hipSetDevice");
    hipStream_t stream = static_cast<hipStream_t>(cpprob_hip_stream(ctx.get()));
    const int64_t ld = (int64_t)n;
    const size_t n_real = st.real_rows(), n_int = st.int_ids.size();        // a vector-valued real predict owns one column per component
    const int T = (int)st.n_observe;
    const bool smc = algorithm == StateType::smc;

    DevBuf<Tuple> d_obs(1);
    hip_check(hipMemcpy(d_obs.p, observes_v, sizeof(Tuple), hipMemcpyHostToDevice), "copy observes");
    DevBuf<double> d_real(n_real * n), d_logw0(n), d_logw1(smc ? n : 0);
    DevBuf<int32_t> d_int(n_int * n), d_anc(smc ? n : 0), d_ns0(smc ? n : 0), d_ns1(smc ? n : 0);
    DevBuf<uint64_t> d_tr0(smc ? S * n : 0), d_tr1(smc ? S * n : 0);
    // what the host reads when the run is over, side by side -- one copy: [T] step ESS, log evidence | [T] resampling decisions, overflow flag
    const size_t tail_doubles = (size_t)T + 1, tail_ints = (size_t)T + 1, tail_bytes = tail_doubles * sizeof(double) + tail_ints * sizeof(int32_t);
    DevBuf<unsigned char> d_tail(tail_bytes);
    double* const d_ess_p = reinterpret_cast<double*>(d_tail.p);
    double* const d_logz_p = d_ess_p + T;
    int32_t* const d_res_p = reinterpret_cast<int32_t*>(d_tail.p + tail_doubles * sizeof(double));
    int32_t* const d_overflow_p = d_res_p + T;
    hip_check(hipMemsetAsync(d_tail.p, 0, tail_bytes, stream), "hipMemsetAsync");   // on the launches' own (non-blocking) stream
    std::vector<unsigned char> h_tail(tail_bytes);
    auto read_tail = [&]() {
        hip_check(hipMemcpyAsync(h_tail.data(), d_tail.p, tail_bytes, hipMemcpyDeviceToHost, stream), "copy the run's tail");
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
    };
    const double* const h_ess_p = reinterpret_cast<const double*>(h_tail.data());
    const int32_t* const h_res_p = reinterpret_cast<const int32_t*>(h_tail.data() + tail_doubles * sizeof(double));
    double* logw[2] = {d_logw0.p, d_logw1.p};
    int32_t* ns[2] = {d_ns0.p, d_ns1.p};
    uint64_t* tr[2] = {d_tr0.p, d_tr1.p};
    const dim3 grid((unsigned)((n + device::kLaneBlock - 1) / device::kLaneBlock)), block(device::kLaneBlock);

    ModelKernelArgs a{};
    a.n = (int64_t)n; a.ld = ld; a.seed = opt.seed; a.trace_cap = (uint32_t)S; a.overflow = d_overflow_p; a.pid0 = opt.particle_offset;
    a.pred_real_cap = (uint32_t)n_real; a.pred_int_cap = (uint32_t)n_int;
    // (windowed replay's buffers: allocated before the clock starts, like the others)
    const bool windowed = smc && st.window >= 0;
    const uint32_t w = (uint32_t)std::max(1, st.window);
    DevBuf<uint64_t> d_c0(windowed ? (size_t)w * n : 0), d_c1(windowed ? (size_t)w * n : 0);
    DevBuf<int32_t> d_anc_all(windowed ? (size_t)T * n : 0);
    DevBuf<double> d_real_gen(windowed ? n_real * n : 0);
    DevBuf<int32_t> d_int_gen(windowed ? n_int * n : 0);
    // SMC bookkeeping between two launches of the model body, on the device: systematic resampling runs on fixed-point weights
    // (integer masses: three short launches, cpprob_amd/csrc/bookkeep_fixed.hpp), the other resamplers on the floating-point CDF
    auto bookkeep = [&](const double* lw, int t, bool last, int32_t* anc_out, double* ess, int32_t* resd, double* logz) {
        if (opt.resampler == CPPROB_HIP_RESAMPLE_SYSTEMATIC && n <= (std::size_t(1) << 28))
            ctx.check(cpprob_hip_smc_bookkeep_fixed(ctx.get(), lw, n, opt.seed, t, last ? 1 : 0, opt.ess_threshold, ess, resd, logz, anc_out), "cpprob_hip_smc_bookkeep_fixed");
        else
            ctx.check(cpprob_hip_smc_bookkeep(ctx.get(), opt.resampler, lw, n, opt.seed, t, last ? 1 : 0, opt.ess_threshold, ess, resd, logz, anc_out), "cpprob_hip_smc_bookkeep");
    };
    // The context sizes its own scratch (hierarchy of sums, integer weights, normalisation partials) at the first call that needs it:
    // one bookkeeping pass and one normalisation over zeroed log-weights before the clock starts -- allocation, like the buffers above.
    // (Their outputs are overwritten: step 0 starts the evidence, the flag word is cleared again.)
    if (smc && T > 0) {
        hip_check(hipMemsetAsync(d_logw0.p, 0, n * sizeof(double), stream), "hipMemsetAsync");
        bookkeep(d_logw0.p, 0, true, d_anc.p, d_ess_p, d_res_p, d_logz_p);
        double warm[3];
        ctx.check(cpprob_hip_logsumexp_ess(ctx.get(), d_logw0.p, n, warm), "cpprob_hip_logsumexp_ess");      // (synchronises)
        hip_check(hipMemsetAsync(d_tail.p, 0, tail_bytes, stream), "hipMemsetAsync");
        hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");
        if (windowed && !store) {                                  // (the read-out's table of which record belongs to which generation: known from the dry run)
            std::vector<int32_t> g;
            for (int s2 : (n_real ? st.real_row_step : st.int_hit_step)) g.push_back(std::min(s2, T - 1));
            if (!g.empty()) ctx.check(cpprob_hip_lineage_prepare(ctx.get(), g.data(), (int32_t)g.size(), T), "cpprob_hip_lineage_prepare");
        }
    }
    const auto t_start = std::chrono::steady_clock::now();
    double log_z = 0.0;
    int cur = 0, n_resampled = 0;
    bool stats_on_walk = false;                                       // windowed SMC without a particle store: the read-out rode the lineage walk
    std::vector<double> walk_real, walk_int;
    double walk_lse_ess[2] = {0.0, 0.0};
    bool smc_log_z_done = false;
    res.step_ess.clear();
    if (!smc) {
        a.logw_out = logw[0]; a.pred_real = d_real.p; a.pred_int = d_int.p; a.first_observe = 0; a.stop_after = -1;
        hipLaunchKernelGGL((model_kernel<Caller, Tuple>), grid, block, 0, stream, a, (const Tuple*)d_obs.p);
        hip_check(hipGetLastError(), "model_kernel");
    } else if (windowed) {
        // Windowed replay: the host probe found that a step depends on its ancestor's last `w` samples only.  Per step ONE gather of
        // w carried values and one row of new ones -- the traffic of the hand-fused kernels -- instead of re-reading and re-writing
        // the whole trace; every launch records the predicts of its own step, ancestors are kept per step, and the traces are read
        // out once at the end by walking the lineages (cpprob_hip_lineage_gather).
        uint64_t* carry[2] = {d_c0.p, d_c1.p};
        a.windowed = 1; a.win = w;
        for (int t = 0; t < T; ++t) {
            const bool last = t + 1 == T;
            a.anc = t > 0 ? d_anc_all.p + (size_t)t * n : nullptr;
            a.resampled_prev = t > 0 ? d_res_p + (t - 1) : nullptr;
            a.logw_in = t > 0 ? logw[cur] : nullptr;
            a.logw_out = logw[cur ^ 1];
            a.carry_in = t > 0 ? carry[cur] : nullptr; a.carry_out = last ? nullptr : carry[cur ^ 1];
            a.fresh_lo = t > 0 ? (int32_t)st.samples_before_observe[(size_t)t - 1] : 0;
            a.next_fresh = (int32_t)st.samples_before_observe[(size_t)t];
            a.pred_real = d_real_gen.p; a.pred_int = d_int_gen.p;
            a.first_observe = t; a.stop_after = last ? -1 : t;
            hipLaunchKernelGGL((model_kernel<Caller, Tuple>), grid, block, 0, stream, a, (const Tuple*)d_obs.p);
            hip_check(hipGetLastError(), "model_kernel");
            cur ^= 1;
            bookkeep(logw[cur], t, last, last ? d_anc.p : d_anc_all.p + (size_t)(t + 1) * n, d_ess_p, d_res_p, d_logz_p);
        }
        // traces: hit h was recorded in the slots of generation step(h); follow every final particle's lineage back to it.  When
        // nobody asked for the traces themselves, StatsPrinter's numbers are taken on that walk (cpprob_hip_lineage_moments / _hist).
        auto gens = [&](const std::vector<int>& steps) { std::vector<int32_t> g; for (int s2 : steps) g.push_back(std::min(s2, T - 1)); return g; };
        if (!store) {
            if (n_real) {
                const std::vector<int32_t> g = gens(st.real_row_step);
                walk_real.resize(4 * n_real);
                ctx.check(cpprob_hip_lineage_moments(ctx.get(), d_anc_all.p, d_res_p, T, n, d_real_gen.p, g.data(), (int32_t)g.size(), logw[cur], walk_real.data()), "cpprob_hip_lineage_moments");
            }
            if (n_int) {
                const std::vector<int32_t> g = gens(st.int_hit_step);
                walk_int.resize(8 * n_int);
                ctx.check(cpprob_hip_lineage_hist(ctx.get(), d_anc_all.p, d_res_p, T, n, d_int_gen.p, g.data(), (int32_t)g.size(), logw[cur], 8, walk_int.data(), walk_lse_ess), "cpprob_hip_lineage_hist");
            }
            stats_on_walk = true;
        } else {
        if (n_real) {
            const std::vector<int32_t> g = gens(st.real_row_step);
            ctx.check(cpprob_hip_lineage_gather(ctx.get(), d_anc_all.p, d_res_p, T, n, d_real_gen.p, 0, g.data(), (int32_t)g.size(), d_real.p), "cpprob_hip_lineage_gather");
        }
        if (n_int) {
            const std::vector<int32_t> g = gens(st.int_hit_step);
            ctx.check(cpprob_hip_lineage_gather(ctx.get(), d_anc_all.p, d_res_p, T, n, d_int_gen.p, 1, g.data(), (int32_t)g.size(), d_int.p), "cpprob_hip_lineage_gather");
        }
        }
        read_tail();
        res.step_ess.assign(h_ess_p, h_ess_p + T);
        log_z = h_ess_p[T];
        for (int t = 0; t < T; ++t) n_resampled += h_res_p[t];
        smc_log_z_done = true;
        res.replay_window = (int)w;
    } else {
        for (int t = 0; t < T; ++t) {
            const bool last = t + 1 == T;
            a.anc = t > 0 ? d_anc.p : nullptr;
            a.resampled_prev = t > 0 ? d_res_p + (t - 1) : nullptr;
            a.logw_in = t > 0 ? logw[cur] : nullptr;
            a.logw_out = logw[cur ^ 1];
            a.trace_in = t > 0 ? tr[cur] : nullptr; a.trace_out = tr[cur ^ 1];
            a.nstored_in = t > 0 ? ns[cur] : nullptr; a.nstored_out = ns[cur ^ 1];
            a.pred_real = last ? d_real.p : nullptr; a.pred_int = last ? d_int.p : nullptr;
            a.first_observe = t; a.stop_after = last ? -1 : t;
            hipLaunchKernelGGL((model_kernel<Caller, Tuple>), grid, block, 0, stream, a, (const Tuple*)d_obs.p);
            hip_check(hipGetLastError(), "model_kernel");
            cur ^= 1;
            // normalise, ESS test (thesis p.37), evidence, ancestors of the next generation: all on the device
            bookkeep(logw[cur], t, last, d_anc.p, d_ess_p, d_res_p, d_logz_p);
        }
        read_tail();
        res.step_ess.assign(h_ess_p, h_ess_p + T);
        log_z = h_ess_p[T];
        for (int t = 0; t < T; ++t) n_resampled += h_res_p[t];
        smc_log_z_done = true;
    }
    fill_predict_names(res, st);
    double lse_ess[2] = {0.0, 0.0};
    bool have_norm = false;
    // StatsPrinter's numbers of every predict hit: all columns of a kind in one device pass against the final weights
    if (n_real) {
        std::vector<double> o4(4 * n_real);
        if (stats_on_walk) o4 = walk_real;
        else ctx.check(cpprob_hip_weighted_moments_columns(ctx.get(), d_real.p, n_real, n, logw[cur], n, o4.data()), "cpprob_hip_weighted_moments_columns");
        lse_ess[0] = o4[2]; lse_ess[1] = o4[3]; have_norm = true;
        for (size_t k = 0, row = 0; k < st.real_ids.size(); ++k) {
            PredictStats& p = res.predicts[k];
            for (size_t d = 0; d < st.real_width[k]; ++d, ++row) { p.mean_nd.push_back(o4[4 * row]); p.variance_nd.push_back(o4[4 * row + 1]); }
            p.mean = p.mean_nd[0]; p.variance = p.variance_nd[0];
        }
    }
    const size_t n_real_hits = st.real_ids.size();
    if (n_int) {
        std::vector<double> h(8 * n_int);
        if (stats_on_walk) { h = walk_int; if (!have_norm) { lse_ess[0] = walk_lse_ess[0]; lse_ess[1] = walk_lse_ess[1]; } }
        else ctx.check(cpprob_hip_weighted_hist_columns(ctx.get(), d_int.p, n_int, n, logw[cur], n, 8, h.data(), have_norm ? nullptr : lse_ess), "cpprob_hip_weighted_hist_columns");
        have_norm = true;
        for (size_t k = 0; k < n_int; ++k) {
            int top = 8;
            while (top > 1 && h[8 * k + top - 1] == 0.0) --top;
            res.predicts[n_real_hits + k].probabilities.assign(h.begin() + 8 * k, h.begin() + 8 * k + top);
        }
    }
    if (!have_norm) {                                                  // (a model without predicts: the weights' own pass)
        double o3[3];
        ctx.check(cpprob_hip_logsumexp_ess(ctx.get(), logw[cur], n, o3), "cpprob_hip_logsumexp_ess");
        lse_ess[0] = o3[1]; lse_ess[1] = o3[2];
    }
    if (!smc_log_z_done) log_z += lse_ess[0] - std::log((double)n);   // SIS: evidence = mean weight
    res.n_particles = n; res.log_evidence = log_z; res.log_norm = lse_ess[0]; res.ess = lse_ess[1]; res.n_resampled = n_resampled; res.used_builtin = false;
    res.run_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();   // (everything StatsPrinter prints; the calls above synchronised)
    if (store) {
        store->n = n;
        store->logw.resize(n); store->real.resize(n_real * n); store->ints.resize(n_int * n);
        hip_check(hipMemcpyAsync(store->logw.data(), logw[cur], n * sizeof(double), hipMemcpyDeviceToHost, stream), "copy logw");
        if (n_real) hip_check(hipMemcpyAsync(store->real.data(), d_real.p, n_real * n * sizeof(double), hipMemcpyDeviceToHost, stream), "copy real predicts");
        if (n_int) hip_check(hipMemcpyAsync(store->ints.data(), d_int.p, n_int * n * sizeof(int32_t), hipMemcpyDeviceToHost, stream), "copy int predicts");
    }
    if (!smc) read_tail(); else hip_check(hipStreamSynchronize(stream), "hipStreamSynchronize");    // (SMC: the tail was read when the last launch had been issued)
    const int32_t overflow = h_res_p[T];
    if (overflow == 2)
        throw std::runtime_error("cpprob::inference: a particle executed more predict statements than the model's dry run did; the number "
                                 "and order of observe / predict statements must not depend on sampled values on the device path");
    return overflow == 3 ? 3 : (overflow != 0 ? 1 : 0);
}

template <class Caller>
void generic_launcher(StateType algorithm, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt,
                      Result& res, HostStore* store)
{
    // trace rows: the structural dry run saw st.n_sample sample statements; particles of models with data-dependent loops
    // (rejection sampling) may execute more -- start with head-room, and on overflow repeat the run with 4x the rows
    std::size_t S = 4 * st.n_sample + 16;
    detail::TraceStructure st_full = st;
    st_full.window = -1;
    const detail::TraceStructure* use = &st;
    res.markov_crosscheck = 0;
    if (algorithm == StateType::smc && st.window >= 0 && opt.markov_crosscheck) {
        // The host probe saw a handful of traces; a dependence on older samples that shows on one trace in a thousand passes it.  So
        // the window is certified on the device before it is used: a pilot population under windowed replay and under full replay
        // -- same particle ids, same seed, hence the same variates, weights and ancestors if the window is right -- must agree in
        // every bit of every weight and predict.  Once per model, trace shape and window in this process.
        static std::mutex mu;
        static std::map<std::tuple<std::size_t, std::size_t, int>, bool> certified;
        const auto key = std::make_tuple((std::size_t)st.n_observe, (std::size_t)st.n_sample, st.window);
        bool known = false, ok = false;
        { std::lock_guard<std::mutex> lock(mu); const auto it = certified.find(key); if (it != certified.end()) { known = true; ok = it->second; } }
        if (!known) {
            const std::size_t n_pilot = std::min<std::size_t>(n, 8192);
            Result r_win, r_full;
            HostStore s_win, s_full;
            const int rc_win = generic_attempt<Caller>(algorithm, observes_v, n_pilot, st, opt, r_win, &s_win, S);
            const int rc_full = rc_win == 0 ? generic_attempt<Caller>(algorithm, observes_v, n_pilot, st_full, opt, r_full, &s_full, S) : 1;
            auto same = [](const std::vector<double>& x, const std::vector<double>& y) {
                return x.size() == y.size() && (x.empty() || std::memcmp(x.data(), y.data(), x.size() * sizeof(double)) == 0);
            };
            ok = rc_win == 0 && rc_full == 0 && same(s_win.logw, s_full.logw) && same(s_win.real, s_full.real) && s_win.ints == s_full.ints &&
                 std::memcmp(&r_win.log_evidence, &r_full.log_evidence, sizeof(double)) == 0;
            std::lock_guard<std::mutex> lock(mu);
            certified[key] = ok;
        }
        res.markov_crosscheck = ok ? 1 : -1;
        if (!ok) use = &st_full;
    }
    for (int attempt = 0; attempt < 5; ++attempt) {
        const int rc = generic_attempt<Caller>(algorithm, observes_v, n, *use, opt, res, store, S);
        if (rc == 0) return;
        if (rc == 3) use = &st_full;                               // the device saw what the host probe did not: replay the whole trace
        else S *= 4;
    }
    throw std::runtime_error("cpprob::inference(smc): a particle executed more than " + std::to_string(S / 4) + " sample statements "
                             "(data-dependent loop, e.g. rejection sampling, that rarely terminates); use StateType::sis for this model");
}

// Device view (cpprob/detail/device_vector.hpp): the launcher receives the HOST function's observes tuple (std::vector elements) and
// hands the device the view's tuple (fixed-capacity elements), converted element by element.
template <class To, class From, std::size_t... I>
To convert_observes(const From& from, std::index_sequence<I...>)
{
    return To(typename std::tuple_element<I, To>::type(std::get<I>(from))...);
}
template <class HostTuple, class Caller>
void generic_launcher_view(StateType algorithm, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt,
                           Result& res, HostStore* store)
{
    using DevTuple = typename Caller::observes_t;
    static_assert(std::tuple_size<HostTuple>::value == std::tuple_size<DevTuple>::value, "host model and device view take the same number of arguments");
    const DevTuple dev = convert_observes<DevTuple>(*static_cast<const HostTuple*>(observes_v), std::make_index_sequence<std::tuple_size<DevTuple>::value>{});
    generic_launcher<Caller>(algorithm, &dev, n, st, opt, res, store);
}

template <class HostFP, HostFP H, class DevFP, DevFP D>
bool register_model_view(const char* name)
{
    Entry e; e.name = name; e.generic = &generic_launcher_view<tuple_observes_t<HostFP>, FunctionCaller<DevFP, D>>; e.generic_vectors = true;
    return add_entry(Key{reinterpret_cast<const void*>(H), 0}, e);
}

template <class FP, FP F>
bool register_model(const char* name)
{
    Entry e; e.name = name; e.generic = &generic_launcher<FunctionCaller<FP, F>>;
    return add_entry(Key{reinterpret_cast<const void*>(F), 0}, e);
}

// functor models (e.g. models::Gauss<>): keyed by type, called through a default-constructed instance
template <class Functor>
bool register_functor(const char* name)
{
    Entry e; e.name = name; e.generic = &generic_launcher<FunctorCaller<Functor>>;
    return add_entry(Key{nullptr, typeid(Functor).hash_code()}, e);
}

}  // namespace gpu
}  // namespace cpprob

#define CPPROB_REGISTER_MODEL(fn) \
    static const bool CPPROB_PP_CAT(cpprob_reg_model_, __LINE__) = ::cpprob::gpu::register_model<decltype(&fn), &fn>(#fn)
// host function, its instantiation inside namespace cpprob_device_view (see cpprob/device_view_begin.hpp)
#define CPPROB_REGISTER_MODEL_VIEW(host_fn, device_fn) \
    static const bool CPPROB_PP_CAT(cpprob_reg_view_, __LINE__) = ::cpprob::gpu::register_model_view<decltype(&host_fn), &host_fn, decltype(&device_fn), &device_fn>(#host_fn)
#define CPPROB_REGISTER_FUNCTOR(type) \
    static const bool CPPROB_PP_CAT(cpprob_reg_functor_, __LINE__) = ::cpprob::gpu::register_functor<type>(#type)

#endif
