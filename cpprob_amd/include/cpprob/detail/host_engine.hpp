// Host driver of a device inference run: RAII over the C ABI (include/cpprob_hip.h), the built-in-model
// path, the posterior file dump in the reference grammar, and the glue cpprob::inference uses.
// Plain C++14.  Errors from the C ABI become std::runtime_error (the reference has no error returns
// on this path: SURVEY section 8(b)).
#ifndef CPPROB_COMPAT_DETAIL_HOST_ENGINE_HPP
#define CPPROB_COMPAT_DETAIL_HOST_ENGINE_HPP
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <limits>
#include <memory>
#include <algorithm>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

#include "cpprob/detail/registry.hpp"
#include "cpprob/detail/traits.hpp"
#include "cpprob_hip.h"

namespace cpprob {
namespace gpu {

class Context {
public:
    explicit Context(int device)
    {
        const int rc = cpprob_hip_create(device, &h_);
        if (rc) throw std::runtime_error(std::string("cpprob_hip_create: ") + cpprob_hip_last_error(nullptr));
    }
    ~Context() { if (h_) cpprob_hip_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    cpprob_hip_ctx* get() const { return h_; }
    void check(int rc, const char* what) const
    {
        if (rc) throw std::runtime_error(std::string(what) + ": " + cpprob_hip_last_error(h_));
    }
private:
    cpprob_hip_ctx* h_ = nullptr;
};

// Contexts are kept between two cpprob::inference calls (the reference's call is `inference(...)` again and again, src/main.cpp:96-100):
// a context made per call cost 1.6-2.2 ms of stream and buffer creation around a 0.2 ms run of 10^6 HMM particles.  A call leases one
// for its device (concurrent calls get contexts of their own); cpprob::gpu::release_device_resources() destroys the idle ones -- they
// are deliberately not torn down by static destructors, which run when the HIP runtime may already be gone.
struct ContextPool { std::mutex mu; std::vector<std::pair<int, std::unique_ptr<Context>>> idle; };
CPPROB_REGISTRY_VISIBLE inline ContextPool& context_pool() { static ContextPool* p = new ContextPool; return *p; }
inline void release_contexts()
{
    ContextPool& p = context_pool();
    std::lock_guard<std::mutex> lock(p.mu);
    p.idle.clear();
}
class ContextLease {
public:
    explicit ContextLease(int device) : device_(device)
    {
        ContextPool& p = context_pool();
        {
            std::lock_guard<std::mutex> lock(p.mu);
            for (auto it = p.idle.begin(); it != p.idle.end(); ++it)
                if (it->first == device) { c_ = std::move(it->second); p.idle.erase(it); break; }
        }
        if (!c_) { c_.reset(new Context(device)); add_release_hook(&release_contexts); }
    }
    ~ContextLease()
    {
        // (a call that threw leaves its context in an unknown state: destroyed, not returned)
        if (!ok_) return;
        ContextPool& p = context_pool();
        std::lock_guard<std::mutex> lock(p.mu);
        p.idle.emplace_back(device_, std::move(c_));
    }
    Context& operator*() const { return *c_; }
    Context* operator->() const { return c_.get(); }
    void done() { ok_ = true; }
private:
    int device_;
    std::unique_ptr<Context> c_;
    bool ok_ = false;
};

// ---- flatten the observes tuple into doubles (scalars and std::array<double, N>) -----------------
template <class T, std::enable_if_t<std::is_arithmetic<T>::value, int> = 0>
void flatten_one(std::vector<double>& out, T x) { out.push_back(static_cast<double>(x)); }
template <class T, std::size_t N> void flatten_one(std::vector<double>& out, const std::array<T, N>& a) { for (const auto& x : a) out.push_back(static_cast<double>(x)); }
template <class T> void flatten_one(std::vector<double>& out, const std::vector<T>& a) { for (const auto& x : a) out.push_back(static_cast<double>(x)); }
template <class Tuple, std::size_t... I>
void flatten_impl(std::vector<double>& out, const Tuple& t, std::index_sequence<I...>) { (void)std::initializer_list<int>{(flatten_one(out, std::get<I>(t)), 0)...}; }
template <class... A>
std::vector<double> flatten(const std::tuple<A...>& t) { std::vector<double> out; flatten_impl(out, t, std::index_sequence_for<A...>{}); return out; }

// ---- posterior files: StateInfer::dump_predicts / dump_ids (src/cpprob/state.cpp:250-267) -----------
// line i of <file>.real / .int:  ([(id v) (id v) ...] logw)   scientific, precision digits10 = 15
inline void dump_posterior(const std::string& file, const detail::TraceStructure& st, const HostStore& hs, std::size_t max_particles)
{
    const std::size_t n = (max_particles && max_particles < hs.n) ? max_particles : hs.n;
    auto write = [&](const std::string& path, const std::vector<std::size_t>& ids, bool is_int) {
        if (ids.empty()) { std::remove(path.c_str()); return; }          // all-empty files are removed (state.cpp:166-174)
        std::ofstream f(path.c_str(), std::ios::app);                     // append mode, as the reference (state.cpp:264)
        f.precision(std::numeric_limits<double>::digits10);
        f << std::scientific;
        for (std::size_t i = 0; i < n; ++i) {
            f << "([";
            std::size_t row = 0;
            for (std::size_t k = 0; k < ids.size(); ++k) {
                if (k) f << ' ';
                f << '(' << ids[k] << ' ';
                if (is_int) f << hs.ints[k * hs.n + i];
                else {
                    const std::size_t w = st.real_width[k];          // an NDArray prints as [v0 v1 ...] (ndarray.hpp:273-288)
                    if (w != 1) f << '[';
                    for (std::size_t d = 0; d < w; ++d) { if (d) f << ' '; f << hs.real[(row + d) * hs.n + i]; }
                    if (w != 1) f << ']';
                    row += w;
                }
                f << ')';
            }
            f << "] " << hs.logw[i] << ")\n";
        }
    };
    write(file + ".int", st.int_ids, true);
    write(file + ".real", st.real_ids, false);
    std::remove((file + ".any").c_str());
    std::ofstream ids((file + ".ids").c_str());
    for (const auto& a : st.addresses) ids << a << std::endl;
}

inline void fill_predict_names(Result& res, const detail::TraceStructure& st)
{
    res.predicts.clear();
    for (std::size_t id : st.real_ids) { PredictStats p; p.address = st.addresses[id]; p.is_int = false; res.predicts.push_back(p); }
    for (std::size_t id : st.int_ids) { PredictStats p; p.address = st.addresses[id]; p.is_int = true; res.predicts.push_back(p); }
}

// ---- built-in models over several GPUs: cpprob_hip_group_* (one joint population, exact global resampling) -------------------
class Group {
public:
    explicit Group(const std::vector<int>& devices)
    {
        std::vector<std::int32_t> d(devices.begin(), devices.end());
        const int rc = cpprob_hip_group_create(d.data(), static_cast<std::int32_t>(d.size()), static_cast<std::int32_t>(d.size()), 0, nullptr, &h_);
        if (rc) throw std::runtime_error(std::string("cpprob_hip_group_create: ") + cpprob_hip_group_last_error(nullptr));
    }
    ~Group() { if (h_) cpprob_hip_group_destroy(h_); }
    Group(const Group&) = delete;
    Group& operator=(const Group&) = delete;
    cpprob_hip_group* get() const { return h_; }
    void check(int rc, const char* what) const
    {
        if (rc) throw std::runtime_error(std::string(what) + ": " + cpprob_hip_group_last_error(h_));
    }
private:
    cpprob_hip_group* h_ = nullptr;
};

inline void run_builtin_group(StateType algorithm, int model_id, const std::vector<double>& obs, std::size_t n, const detail::TraceStructure& st,
                              const Options& opt, Result& res, HostStore* store)
{
    const std::size_t world = opt.devices.size();
    Group grp(opt.devices);
    cpprob_hip_config cfg{};
    cfg.algorithm = algorithm == StateType::smc ? CPPROB_HIP_ALG_SMC : CPPROB_HIP_ALG_SIS;
    cfg.model = model_id;
    cfg.resampler = opt.resampler;
    cfg.resample_scope = CPPROB_HIP_SCOPE_EXCHANGE;
    cfg.keep_history = 1;
    cfg.ess_threshold = opt.ess_threshold;
    cfg.seed = opt.seed;
    cfg.n_particles = n; cfg.particle_offset = 0; cfg.n_global = n;
    grp.check(cpprob_hip_group_begin(grp.get(), &cfg, obs.data(), obs.size(), nullptr), "cpprob_hip_group_begin");
    grp.check(cpprob_hip_group_sync(grp.get()), "cpprob_hip_group_sync");   // (begin's buffer clears are allocation: not the run's time)
    const auto t0 = std::chrono::steady_clock::now();
    grp.check(cpprob_hip_group_run(grp.get(), 0), "cpprob_hip_group_run");
    cpprob_hip_summary s{};
    std::int32_t reruns = 0;
    // (sizes from the model structure: results() fills exactly n_predict * stats_per_predict doubles)
    const std::size_t K_guess = st.int_ids.empty() ? 2 : 3;
    const std::size_t T_guess = st.int_ids.empty() ? st.real_rows() : st.int_ids.size();
    std::vector<double> stats(T_guess * K_guess);
    grp.check(cpprob_hip_group_results(grp.get(), &s, stats.data(), stats.size(), &reruns), "cpprob_hip_group_results");
    res.run_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::size_t T = static_cast<std::size_t>(s.n_predict), K = static_cast<std::size_t>(s.stats_per_predict);
    if ((s.is_int ? st.int_ids.size() : st.real_rows()) != T || T * K != stats.size())
        throw std::runtime_error("built-in model kernel and the model function disagree on the number of predict statements");
    res.n_particles = n; res.log_evidence = s.log_evidence; res.ess = s.ess_final; res.log_norm = s.log_norm; res.n_resampled = s.n_resampled;
    res.used_builtin = true; res.n_gpus = static_cast<int>(world); res.exchange_reruns = reruns;
    fill_predict_names(res, st);
    if (s.is_int) {
        for (std::size_t t = 0; t < T; ++t) res.predicts[t].probabilities.assign(stats.begin() + t * K, stats.begin() + (t + 1) * K);
    } else {
        std::size_t row = 0;
        for (std::size_t k = 0; k < st.real_ids.size(); ++k) {
            PredictStats& p = res.predicts[k];
            for (std::size_t d = 0; d < st.real_width[k]; ++d, ++row) { p.mean_nd.push_back(stats[row * K]); p.variance_nd.push_back(stats[row * K + 1]); }
            p.mean = p.mean_nd[0]; p.variance = p.variance_nd[0];
        }
    }
    res.step_ess.assign(T, 0.0);
    cpprob_hip_ctx* c0 = cpprob_hip_group_context(grp.get(), 0);
    if (cpprob_hip_infer_step_trace(c0, res.step_ess.data(), nullptr)) throw std::runtime_error(std::string("cpprob_hip_infer_step_trace: ") + cpprob_hip_last_error(c0));
    if (store) {
        // the shards' traces side by side: particle i of the population lives on rank floor(i * world / n) (contiguous, equal shards)
        store->n = n;
        store->logw.resize(n);
        if (s.is_int) store->ints.resize(T * n); else store->real.resize(T * n);
        std::size_t begin = 0;
        for (std::size_t r = 0; r < world; ++r) {
            const std::size_t nr = n / world + (r < n % world ? 1 : 0);
            cpprob_hip_ctx* c = cpprob_hip_group_context(grp.get(), static_cast<std::int32_t>(r));
            auto chk = [&](int rc, const char* what) { if (rc) throw std::runtime_error(std::string(what) + ": " + cpprob_hip_last_error(c)); };
            chk(cpprob_hip_copy_logw(c, store->logw.data() + begin, nr * sizeof(double)), "cpprob_hip_copy_logw");
            if (s.is_int) {
                std::vector<std::int32_t> part(T * nr);
                chk(cpprob_hip_copy_paths(c, part.data(), part.size() * sizeof(std::int32_t)), "cpprob_hip_copy_paths");
                for (std::size_t t = 0; t < T; ++t) std::copy(part.begin() + t * nr, part.begin() + (t + 1) * nr, store->ints.begin() + t * n + begin);
            } else {
                std::vector<double> part(T * nr);
                chk(cpprob_hip_copy_paths(c, part.data(), part.size() * sizeof(double)), "cpprob_hip_copy_paths");
                for (std::size_t t = 0; t < T; ++t) std::copy(part.begin() + t * nr, part.begin() + (t + 1) * nr, store->real.begin() + t * n + begin);
            }
            begin += nr;
        }
    }
}

// ---- built-in models: the hand-fused kernels behind cpprob_hip_infer_* ---------------------------
inline void run_builtin(StateType algorithm, int model_id, const std::vector<double>& obs, std::size_t n, const detail::TraceStructure& st,
                        const Options& opt, Result& res, HostStore* store)
{
    const auto t_setup = std::chrono::steady_clock::now();
    ContextLease lease(opt.device);
    Context& ctx = *lease;
    cpprob_hip_config cfg{};
    cfg.algorithm = algorithm == StateType::smc ? CPPROB_HIP_ALG_SMC : CPPROB_HIP_ALG_SIS;
    cfg.model = model_id;
    cfg.resampler = opt.resampler;
    cfg.resample_scope = CPPROB_HIP_SCOPE_GLOBAL;
    cfg.keep_history = (opt.keep_history || algorithm != StateType::smc) ? 1 : 0;
    if (!cfg.keep_history && store) throw std::runtime_error("cpprob::inference: a filtering-only run (options().keep_history = false) keeps no traces to dump: set options().dump = false");
    cfg.ess_threshold = opt.ess_threshold;
    cfg.seed = opt.seed;
    cfg.n_particles = n; cfg.particle_offset = 0; cfg.n_global = n;
    ctx.check(cpprob_hip_infer_begin(ctx.get(), &cfg, obs.data(), obs.size()), "cpprob_hip_infer_begin");
    ctx.check(cpprob_hip_sync(ctx.get()), "cpprob_hip_sync");          // (begin's buffer clears are allocation: not the run's time)
    const auto t0 = std::chrono::steady_clock::now();
    res.setup_seconds = std::chrono::duration<double>(t0 - t_setup).count();      // (context, buffers, clears: what a call adds to its run)
    ctx.check(cpprob_hip_infer_run(ctx.get(), 0), "cpprob_hip_infer_run");
    // (everything the result holds in ONE read-back behind one stream synchronisation; sizes from the model structure: the engine
    //  fills exactly n_predict * stats_per_predict doubles -- histograms of up to 8 bins, or {mean, variance})
    cpprob_hip_summary s{};
    const std::size_t T_guess = st.int_ids.empty() ? st.real_rows() : st.int_ids.size();
    const std::size_t T_cap = std::max<std::size_t>(std::max(T_guess, obs.size()), 1);     // (the engine's rows: one per observation at most)
    std::vector<double> stats(T_cap * 8);
    res.step_ess.assign(T_cap, 0.0);
    ctx.check(cpprob_hip_infer_results(ctx.get(), &s, stats.data(), stats.size(), res.step_ess.data(), nullptr), "cpprob_hip_infer_results");
    res.run_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::size_t T = static_cast<std::size_t>(s.n_predict), K = static_cast<std::size_t>(s.stats_per_predict);
    if ((s.is_int ? st.int_ids.size() : st.real_rows()) != T || T != T_guess)
        throw std::runtime_error("built-in model kernel and the model function disagree on the number of predict statements");
    stats.resize(T * K); res.step_ess.resize(T);
    res.n_particles = n; res.log_evidence = s.log_evidence; res.ess = s.ess_final; res.log_norm = s.log_norm; res.n_resampled = s.n_resampled;
    res.used_builtin = true;
    fill_predict_names(res, st);
    if (s.is_int) {
        for (std::size_t t = 0; t < T; ++t) res.predicts[t].probabilities.assign(stats.begin() + t * K, stats.begin() + (t + 1) * K);
    } else {
        std::size_t row = 0;                                         // a vector-valued hit owns one engine row per component
        for (std::size_t k = 0; k < st.real_ids.size(); ++k) {
            PredictStats& p = res.predicts[k];
            for (std::size_t d = 0; d < st.real_width[k]; ++d, ++row) { p.mean_nd.push_back(stats[row * K]); p.variance_nd.push_back(stats[row * K + 1]); }
            p.mean = p.mean_nd[0]; p.variance = p.variance_nd[0];
        }
    }
    if (store) {
        store->n = n;
        store->logw.resize(n);
        ctx.check(cpprob_hip_copy_logw(ctx.get(), store->logw.data(), n * sizeof(double)), "cpprob_hip_copy_logw");
        if (s.is_int) { store->ints.resize(T * n); ctx.check(cpprob_hip_copy_paths(ctx.get(), store->ints.data(), T * n * sizeof(std::int32_t)), "cpprob_hip_copy_paths"); }
        else { store->real.resize(T * n); ctx.check(cpprob_hip_copy_paths(ctx.get(), store->real.data(), T * n * sizeof(double)), "cpprob_hip_copy_paths"); }
    }
    // ---- replicates (error bars): further seeds, up to three runs in flight on contexts of their own ----
    const int R = opt.replicates;
    if (R > 1) {
        const std::size_t H = res.predicts.size();
        res.n_replicates = R;
        res.replicate_values.assign(H, std::vector<double>(static_cast<std::size_t>(R), 0.0));
        std::vector<double> lz(static_cast<std::size_t>(R), 0.0);
        auto value_of = [&](const std::vector<double>& stv, std::size_t hit) {          // mean of component 0 / P(x = 0)
            if (s.is_int) return stv[hit * K];
            std::size_t row = 0;
            for (std::size_t k = 0; k < hit; ++k) row += st.real_width[k];
            return stv[row * K];
        };
        for (std::size_t h = 0; h < H; ++h) res.replicate_values[h][0] = value_of(stats, h);
        lz[0] = s.log_evidence;
        const int lanes = R - 1 < 3 ? R - 1 : 3;
        std::vector<std::unique_ptr<Context>> extra;
        for (int l = 0; l < lanes; ++l) {
            extra.emplace_back(new Context(opt.device));
            extra.back()->check(cpprob_hip_infer_begin(extra.back()->get(), &cfg, obs.data(), obs.size()), "cpprob_hip_infer_begin");
        }
        std::vector<int> pending(static_cast<std::size_t>(lanes), -1);
        auto harvest = [&](int l) {
            const int r = pending[static_cast<std::size_t>(l)];
            if (r < 0) return;
            cpprob_hip_summary sr{};
            std::vector<double> str(T * K);
            extra[l]->check(cpprob_hip_infer_summary(extra[l]->get(), &sr), "cpprob_hip_infer_summary");
            extra[l]->check(cpprob_hip_infer_stats(extra[l]->get(), str.data(), str.size()), "cpprob_hip_infer_stats");
            lz[static_cast<std::size_t>(r)] = sr.log_evidence;
            for (std::size_t h = 0; h < H; ++h) res.replicate_values[h][static_cast<std::size_t>(r)] = value_of(str, h);
            pending[static_cast<std::size_t>(l)] = -1;
        };
        const auto tr0 = std::chrono::steady_clock::now();
        for (int r = 1; r < R; ++r) {
            const int l = (r - 1) % lanes;
            harvest(l);                                               // synchronises that context only; the others keep running
            extra[l]->check(cpprob_hip_infer_run(extra[l]->get(), static_cast<std::uint64_t>(r)), "cpprob_hip_infer_run");
            pending[static_cast<std::size_t>(l)] = r;
        }
        for (int l = 0; l < lanes; ++l) harvest(l);
        res.replicates_seconds = res.run_seconds + std::chrono::duration<double>(std::chrono::steady_clock::now() - tr0).count();
        res.predict_mean.assign(H, 0.0); res.predict_sd.assign(H, 0.0);
        for (std::size_t h = 0; h < H; ++h) {
            double m = 0, v = 0;
            for (double x : res.replicate_values[h]) m += x;
            m /= R;
            for (double x : res.replicate_values[h]) v += (x - m) * (x - m);
            res.predict_mean[h] = m; res.predict_sd[h] = std::sqrt(v / (R - 1));
        }
        double mx = lz[0], acc = 0, ml = 0, vl = 0;
        for (double x : lz) { mx = std::max(mx, x); ml += x; }
        for (double x : lz) acc += std::exp(x - mx);
        ml /= R;
        for (double x : lz) vl += (x - ml) * (x - ml);
        res.log_evidence_mean = mx + std::log(acc / R);
        res.log_evidence_sd = std::sqrt(vl / (R - 1));
    }
    lease.done();                                                     // (the context goes back to the pool)
}

// ---- unchanged models over several GPUs ----------------------------------------------------------------------------------------
// Importance sampling needs no communication until the shards' sums meet: every device runs the model body for its
// contiguous block of particles -- global particle ids select the random streams, so the traces are those ONE device would have
// drawn -- on a host thread of its own, and the shards' self-normalised numbers are combined by their evidence:
//   w_r = exp(L_r - L), L = logsumexp_r L_r;  mean = sum w_r mean_r;  E[x^2] = sum w_r (var_r + mean_r^2);  P = sum w_r P_r;
//   ESS = 1 / sum_r w_r^2 / ESS_r;  log evidence = L - log N.
// StateType::smc: every shard is an ISLAND -- an independent SMC run of its own particles, resampled among themselves -- and the
// islands are combined the same way with L_r = log(n_r Z_r), Z_r the island's evidence estimate: a consistent estimator of the same
// posterior, but not the joint population's resampling (that needs the replayed traces to migrate: not built; the built-in models
// have it, cpprob_hip_group_*).
inline void combine_shards(std::vector<Result>& rr, const std::vector<HostStore>* hs, const std::vector<std::size_t>& begin, std::size_t n,
                           const detail::TraceStructure& st, bool islands, Result& res, HostStore* store);
inline void run_generic_sharded(StateType algorithm, const Entry& e, const void* observes_v, std::size_t n, const detail::TraceStructure& st, const Options& opt,
                                Result& res, HostStore* store)
{
    const bool smc = algorithm == StateType::smc;
    const std::size_t world = opt.devices.size();
    std::vector<Result> rr(world);
    std::vector<HostStore> hs(world);
    std::vector<std::string> errs(world);
    std::vector<std::size_t> begin(world + 1, 0);
    for (std::size_t r = 0; r < world; ++r) begin[r + 1] = begin[r] + n / world + (r < n % world ? 1 : 0);
    std::vector<std::thread> th;
    for (std::size_t r = 0; r < world; ++r)
        th.emplace_back([&, r] {
            try {
                Options o = opt;
                o.device = opt.devices[r]; o.devices.clear(); o.particle_offset = begin[r]; o.dump = false;
                e.generic(algorithm, observes_v, begin[r + 1] - begin[r], st, o, rr[r], store ? &hs[r] : nullptr);
            } catch (const std::exception& ex) { errs[r] = ex.what(); }
        });
    for (auto& t : th) t.join();
    for (std::size_t r = 0; r < world; ++r)
        if (!errs[r].empty()) throw std::runtime_error("cpprob::inference (shard " + std::to_string(r) + "): " + errs[r]);
    combine_shards(rr, store ? &hs : nullptr, begin, n, st, smc, res, store);
    if (smc && store) {
        // an island's log-weights are relative to ITS last resampling: on the population's common scale island r's particles carry
        // log(n_r Z_r) - logsumexp(its weights) more -- what the combined statistics above weigh it by, so that the files say what Result says
        for (std::size_t r = 0; r < world; ++r) {
            double mx = -std::numeric_limits<double>::infinity(), acc = 0;
            for (double v : hs[r].logw) mx = std::max(mx, v);
            for (double v : hs[r].logw) acc += std::exp(v - mx);
            const double shift = rr[r].log_norm - (mx + std::log(acc));
            for (std::size_t i = begin[r]; i < begin[r + 1]; ++i) store->logw[i] += shift;
        }
    }
}

// The shards' self-normalised numbers combined by their masses (log_norm = log of the shard's weight sum on the population's common
// scale): w_r = exp(L_r - L), L = logsumexp_r L_r;  mean = sum w_r mean_r;  E[x^2] = sum w_r (var_r + mean_r^2);  P = sum w_r P_r;
// ESS = 1 / sum_r w_r^2 / ESS_r.  islands: the shards are independent SMC runs, L_r = log(n_r Z_r) and the evidence is the combination's;
// otherwise (SIS shards; the shards of a joint SMC population, whose evidence the caller holds) the shards' weights already share a scale.
inline void combine_shards(std::vector<Result>& rr, const std::vector<HostStore>* hs, const std::vector<std::size_t>& begin, std::size_t n,
                           const detail::TraceStructure& st, bool islands, Result& res, HostStore* store)
{
    const std::size_t world = rr.size();
    const bool smc = islands;
    // (an island's mass: its particles times its evidence estimate; an SIS shard's: the sum of its weights -- the same thing)
    if (smc)
        for (std::size_t r = 0; r < world; ++r) rr[r].log_norm = rr[r].log_evidence + std::log(static_cast<double>(begin[r + 1] - begin[r]));
    double L = -std::numeric_limits<double>::infinity();
    for (const auto& x : rr) L = std::max(L, x.log_norm);
    double acc = 0;
    for (const auto& x : rr) acc += std::exp(x.log_norm - L);
    L += std::log(acc);
    res = rr[0];
    if (smc) { res.n_resampled = 0; for (const auto& x : rr) res.n_resampled = std::max(res.n_resampled, x.n_resampled); res.step_ess.clear(); }
    res.n_particles = n; res.log_norm = L; res.log_evidence = L - std::log(static_cast<double>(n)); res.n_gpus = static_cast<int>(world);
    double inv_ess = 0, secs = 0;
    for (const auto& x : rr) { const double w = std::exp(x.log_norm - L); inv_ess += w * w / x.ess; secs = std::max(secs, x.run_seconds); }
    res.ess = 1.0 / inv_ess; res.run_seconds = secs;
    for (std::size_t k = 0; k < res.predicts.size(); ++k) {
        PredictStats& p = res.predicts[k];
        if (p.is_int) {
            std::size_t top = 0;
            for (const auto& x : rr) top = std::max(top, x.predicts[k].probabilities.size());
            p.probabilities.assign(top, 0.0);
            for (const auto& x : rr) { const double w = std::exp(x.log_norm - L); for (std::size_t s2 = 0; s2 < x.predicts[k].probabilities.size(); ++s2) p.probabilities[s2] += w * x.predicts[k].probabilities[s2]; }
            while (p.probabilities.size() > 1 && p.probabilities.back() == 0.0) p.probabilities.pop_back();
        } else {
            const std::size_t D = p.mean_nd.size();
            std::vector<double> m1(D, 0.0), m2(D, 0.0);
            for (const auto& x : rr) {
                const double w = std::exp(x.log_norm - L);
                for (std::size_t d = 0; d < D; ++d) { const double m = x.predicts[k].mean_nd[d]; m1[d] += w * m; m2[d] += w * (x.predicts[k].variance_nd[d] + m * m); }
            }
            for (std::size_t d = 0; d < D; ++d) { p.mean_nd[d] = m1[d]; p.variance_nd[d] = m2[d] - m1[d] * m1[d]; }
            p.mean = p.mean_nd[0]; p.variance = p.variance_nd[0];
        }
    }
    if (store && hs) {
        // the shards' traces side by side, in global particle order
        const std::size_t n_real = st.real_rows(), n_int = st.int_ids.size();
        store->n = n; store->logw.resize(n); store->real.resize(n_real * n); store->ints.resize(n_int * n);
        for (std::size_t r = 0; r < world; ++r) {
            const std::size_t nr = begin[r + 1] - begin[r];
            std::copy((*hs)[r].logw.begin(), (*hs)[r].logw.end(), store->logw.begin() + begin[r]);
            for (std::size_t row = 0; row < n_real; ++row) std::copy((*hs)[r].real.begin() + row * nr, (*hs)[r].real.begin() + (row + 1) * nr, store->real.begin() + row * n + begin[r]);
            for (std::size_t row = 0; row < n_int; ++row) std::copy((*hs)[r].ints.begin() + row * nr, (*hs)[r].ints.begin() + (row + 1) * nr, store->ints.begin() + row * n + begin[r]);
        }
    }
}

// ---- Markov probe (cpprob/detail/host_trace.hpp: ProbeState): the smallest window of past samples a step depends on, or -1 ------
template <class Func, class ObsTuple>
int probe_markov_window(const Func& f, const ObsTuple& obs, const detail::TraceStructure& st)
{
    const std::size_t T = st.n_observe;
    if (T < 2 || st.vector_statements || st.samples_before_observe.size() != T || st.n_sample == 0) return -1;
    auto run = [&](const detail::ProbeState& cfg) -> bool {
        detail::ProbeState& p = detail::probe();
        p = cfg; p.active = true; p.failed = false; p.ordinal = 0; p.n_obs = 0; p.log.clear();
        // (whatever the model throws on substituted values -- its own checks, a distribution's parameter check -- means "this model
        //  does look at old samples": not Markov, full replay; never an error of a valid inference)
        try { call_f_tuple(f, obs); }
        catch (...) { detail::probe().active = false; return false; }
        p.active = false;
        return !p.failed;
    };
    detail::BoundProbe& bp = detail::bound_probe();
    bp.expected = &st.observe_bound; bp.varies = false;
    struct BoundGuard { detail::BoundProbe& b; ~BoundGuard() { b.expected = nullptr; } } guard{bp};
    const int windows[] = {1, 2, 4, 8};
    for (int w : windows) {
        bool ok = true;
        for (std::uint32_t rep = 0; rep < 8 && ok; ++rep) {
            detail::ProbeState a;
            a.base_seed = 1000003u * (rep + 1); a.n_steps = T; a.record_all = true; a.ordinal_cap = 64 * st.n_sample + 1024;
            if (!run(a)) return -1;
            const std::vector<std::vector<double>> ref = detail::probe().log;
            const std::vector<double> values = detail::probe().values;
            if (detail::probe().ordinal != st.n_sample || detail::probe().n_obs != T) return -1;      // sample / observe counts depend on sampled values
            for (std::size_t step = 1; step < T && ok; ++step) {
                const std::size_t fresh_lo = st.samples_before_observe[step - 1];
                if (fresh_lo <= static_cast<std::size_t>(w)) continue;                                // nothing older than the window yet
                detail::ProbeState b = a;
                b.record_all = false; b.step = step; b.dummy_below = fresh_lo - static_cast<std::size_t>(w); b.replay_below = fresh_lo; b.values = values;
                if (!run(b)) { ok = false; break; }
                const auto& lg = detail::probe().log;
                if (detail::probe().n_obs != T || lg.size() <= step || ref.size() <= step || lg[step] != ref[step]) ok = false;
            }
        }
        if (ok) return w;
    }
    return -1;
}

// ---- the body of cpprob::inference -------------------------------------------------------------------
template <class Func, class... Args>
void run_inference(StateType algorithm, const Func& f, const std::tuple<Args...>& observes, std::size_t n, const std::string& file)
{
    using ObsTuple = tuple_observes_t<Func>;
    static_assert(std::tuple_size<ObsTuple>::value == sizeof...(Args), "the observes tuple must have one element per model argument");
    const ObsTuple obs(observes);                                   // implicit conversions, as call_f_tuple's forwarding does

    // structural dry run on the host (one trace): statement counts, predict types and addresses
    detail::TraceStructure st;
    {
        const StateType saved = algorithm;
        State::set(StateType::dryrun);
        detail::recorder() = &st;
        try { call_f_tuple(f, obs); } catch (...) { detail::recorder() = nullptr; State::set(saved); throw; }
        detail::recorder() = nullptr;
        State::set(saved);
    }
    if (st.n_observe == 0) throw std::runtime_error("cpprob::inference: the model executes no observe statement");
    if (algorithm == StateType::smc && options().markov_probe) {
        st.window = probe_markov_window(f, obs, st);
        bool known = st.observe_bound.size() == st.n_observe;
        for (double b : st.observe_bound) if (!(b == b) || b == std::numeric_limits<double>::infinity() || b == -std::numeric_limits<double>::infinity()) known = false;
        st.bounds_fixed = st.window >= 0 && known && !detail::bound_probe().varies;
    }
    if (st.n_other_predicts) throw std::runtime_error("cpprob::inference: non-scalar predicts (.any file) are not supported by the device engine");

    const Key key = key_of(f, std::integral_constant<bool, detail::fn_traits<std::remove_cv_t<std::remove_reference_t<Func>>>::is_function>{});
    const Entry* e = find_entry(key);
    if (!e) throw std::runtime_error("cpprob::inference: this model has no device code: compile its source with hipcc and add "
                                     "CPPROB_REGISTER_MODEL(<model>) (or CPPROB_REGISTER_BUILTIN) -- there is no CPU fallback");
    if (st.vector_statements && e->builtin_model < 0 && !e->generic_vectors)
        throw std::runtime_error("cpprob::inference: vector-valued statements (multivariate_normal_distribution / NDArray) need the model's device "
                                 "view (CPPROB_REGISTER_MODEL_VIEW, cpprob/device_view_begin.hpp) or a built-in kernel: std::vector cannot live in device code");
    const Options& opt = options();
    Result& res = last_result();
    res = Result();
    HostStore hs;
    HostStore* store = opt.dump ? &hs : nullptr;
    if (opt.devices.size() > 1) {
        const bool builtin = e->builtin_model >= 0 && (opt.prefer_builtin || !e->generic);
        if (builtin) run_builtin_group(algorithm, e->builtin_model, flatten(obs), n, st, opt, res, store);
        else if (e->generic) {
            // smc: ONE joint population where the model has a joint form (a replay window, systematic resampling: cpprob/gpu.hpp,
            // generic_joint_launcher); otherwise islands -- a different estimator, reported as such (Result::joint)
            bool joint = false;
            if (algorithm == StateType::smc && e->generic_joint && !opt.islands) joint = e->generic_joint(algorithm, &obs, n, st, opt, res, store);
            if (!joint) run_generic_sharded(algorithm, *e, &obs, n, st, opt, res, store);
        }
        else throw std::runtime_error("cpprob::inference: registry entry without a launcher");
    }
    else if (e->builtin_model >= 0 && (opt.prefer_builtin || !e->generic || (st.vector_statements && !e->generic_vectors)))
        run_builtin(algorithm, e->builtin_model, flatten(obs), n, st, opt, res, store);
    else if (e->generic) e->generic(algorithm, &obs, n, st, opt, res, store);
    else throw std::runtime_error("cpprob::inference: registry entry without a launcher");
    if (opt.dump) dump_posterior(file, st, hs, opt.dump_max_particles);      // finish_trace() x n + finish_infer()
}

}  // namespace gpu
}  // namespace cpprob
#endif
