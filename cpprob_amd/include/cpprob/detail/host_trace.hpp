// Host side of the statement API: one structural dry run of the model (what StateType::dryrun does in
// the reference, src/main.cpp:108-112) records how many sample / observe / predict statements a trace
// executes, the type and address of every predict hit, and nothing else.  No inference ever runs on
// the host: the particle loop lives on the device.
#ifndef CPPROB_COMPAT_DETAIL_HOST_TRACE_HPP
#define CPPROB_COMPAT_DETAIL_HOST_TRACE_HPP
#include <cxxabi.h>
#include <execinfo.h>

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

#include "cpprob/distributions/utils_base.hpp"
#include "cpprob/ndarray.hpp"

namespace cpprob {
namespace detail {

struct TraceStructure {
    std::size_t n_sample = 0, n_observe = 0;
    // one entry per predict hit, in execution order
    std::vector<std::size_t> real_ids, int_ids;         // address id of each hit (TraceInfer::register_addr_predict, trace.hpp:37-41)
    std::vector<std::size_t> real_width;                // components of each real hit: 1, or the size of an NDArray predict (state.hpp:330-337)
    bool vector_statements = false;                     // a sample / observe / predict of this trace is vector-valued
    std::size_t real_rows() const { std::size_t r = 0; for (auto w : real_width) r += w; return r; }
    std::vector<std::string> addresses;                 // id -> address (the .ids file, state.cpp:250-260)
    std::size_t n_other_predicts = 0;                   // predicts of non-scalar type (the .any file): not carried by the device engine
    // SMC structure: samples executed before observe t is reached, and the step each predict hit (row) belongs to = observes
    // executed before it (a predict placed after observe t reports the state that observe weighed: it runs in step t+1's launch)
    std::vector<std::size_t> samples_before_observe;
    std::vector<int> real_row_step, int_hit_step;
    int window = -1;                                    // >= 0: the model passed the Markov probe -- a step needs only its ancestor's last `window` samples
    // observe #m's largest possible log-density (the distribution's density at its mode, logpdf_max) as the dry run saw it; NaN where
    // unknown.  bounds_fixed: every trace of the Markov probe saw the same values (the distributions' scale parameters do not depend
    // on sampled values) -- the device engine then takes its fixed-point weights against them, and verifies each generation
    std::vector<double> observe_bound;
    bool bounds_fixed = false;
    std::size_t id_of(const std::string& addr)
    {
        auto it = ids_.find(addr);
        if (it != ids_.end()) return it->second;
        const std::size_t id = addresses.size();
        ids_.emplace(addr, id);
        addresses.push_back(addr);
        return id;
    }
private:
    std::unordered_map<std::string, std::size_t> ids_;
};

// Address of a statement that was given none: the chain of call sites from the model's entry function down to the statement, as
// the reference derives it (src/cpprob/utils.cpp:71-128): walk the call stack, drop the innermost frames that belong to namespace
// cpprob, keep everything up to the outermost frame of namespace models, and print "[outer+0xoff ... inner+0xoff]" with demangled
// names -- so two textual call sites get two addresses, and a loop's iterations share one (stats_printer.hpp:106-118 then tells the
// hits of one address apart by their order).  Host only: the device ignores addresses (hit k of a trace is column k).
inline std::string frame_name(const char* symbol_line)
{
    const std::string s(symbol_line);                     // "object(mangled+0xoff) [0xaddr]"
    const std::size_t open = s.find_last_of('('), close = s.find_last_of(')'), plus = s.find_last_of('+');
    if (open == std::string::npos || close == std::string::npos || close < open) return s;
    const std::size_t name_end = (plus != std::string::npos && plus > open && plus < close) ? plus : close;
    const std::string mangled = s.substr(open + 1, name_end - open - 1);
    int status = 1;
    char* dem = mangled.empty() ? nullptr : abi::__cxa_demangle(mangled.c_str(), nullptr, nullptr, &status);
    std::string out = (status == 0 && dem) ? std::string(dem) : mangled;
    std::free(dem);
    return out + s.substr(name_end, close - name_end);   // + "+0xoff": the return address inside the function = the call site
}
inline bool in_namespace(const std::string& frame, const std::string& ns)
{
    // a demangled name mentions its namespace at the start, or after the return type's space (templates carry their return type)
    if (frame.compare(0, ns.size(), ns) == 0) return true;
    for (std::size_t p = frame.find(" " + ns); p != std::string::npos; p = frame.find(" " + ns, p + 1))
        if (p == 0 || frame[p - 1] != ',') return true;
    return false;
}
inline std::string call_site_address()
{
    void* frames[100];
    const int n = backtrace(frames, 100);
    char** lines = backtrace_symbols(frames, n);
    if (!lines) return "<predict>";
    std::vector<std::string> names;
    for (int i = 0; i < n; ++i) names.push_back(frame_name(lines[i]));
    std::free(lines);
    int inner = 0, outer = n - 1;
    while (inner < n && in_namespace(names[(std::size_t)inner], "cpprob::")) ++inner;
    while (outer >= 0 && !in_namespace(names[(std::size_t)outer], "models::")) --outer;
    if (inner >= n || outer < 0) return "<predict>";       // (the model is not in namespace models: one shared address)
    std::string out = "[";
    for (int i = outer; i >= inner; --i) { if (i != outer) out += ' '; out += names[(std::size_t)i]; }
    return out + "]";
}

inline TraceStructure*& recorder() { static thread_local TraceStructure* r = nullptr; return r; }
inline std::mt19937& host_rng() { static thread_local std::mt19937 rng{20260101u}; return rng; }

// Markov probe (host): does a step of the model -- the statements between two observes -- depend on anything but the last w
// sampled values?  One execution with every sample drawn (per-ordinal seeds), one with the samples older than the window replaced
// by value-initialised ones (what the device's windowed replay hands the model): the step's statements must see the same draws,
// the same observe log-densities and the same predict values.  Probed for every step and several seeds by run_inference; a model
// that passes runs SMC with O(T) instead of O(T^2) replay traffic (cpprob/gpu.hpp).
struct ProbeState {
    bool active = false, failed = false;
    std::uint32_t base_seed = 0;
    std::size_t dummy_below = 0, replay_below = 0, ordinal = 0, n_obs = 0, ordinal_cap = 0;   // ordinals < dummy_below: value-initialised; < replay_below: the reference run's
    std::vector<double> values;                         // reference run: every drawn value, by ordinal
    std::size_t step = 0, n_steps = 0;
    bool record_all = false;                            // reference run: log every step; otherwise only `step`
    std::vector<std::vector<double>> log;               // per step
    bool in_step(std::size_t m) const { return m < n_steps ? true : false; }
    std::size_t step_of_now() const { return n_obs < n_steps ? n_obs : n_steps - 1; }
    void put(double v)
    {
        const std::size_t st = step_of_now();
        if (record_all || st == step) { if (log.size() <= st) log.resize(st + 1); log[st].push_back(v); }
    }
};
struct ProbeAbort {};
inline ProbeState& probe() { static thread_local ProbeState p; return p; }
// (kept across the probe's runs: every run's observe #m must see the bound the structural dry run recorded)
struct BoundProbe { const std::vector<double>* expected = nullptr; bool varies = false; };
inline BoundProbe& bound_probe() { static thread_local BoundProbe b; return b; }
inline bool same_bits(double a, double b) { return std::memcmp(&a, &b, sizeof a) == 0; }

// variates a value consumes: one per component
template <class T> std::size_t width_of(const T&) { return 1; }
template <class T> std::size_t width_of(const NDArray<T>& x) { return x.size(); }
template <class V> auto width_of_sized(const V& x, int) -> decltype(x.size()) { return x.size(); }
template <class V> std::size_t width_of_sized(const V&, long) { return 1; }

template <class V> std::enable_if_t<std::is_arithmetic<V>::value> probe_put_value(ProbeState& p, const V& v) { p.put(static_cast<double>(v)); }
template <class V> std::enable_if_t<!std::is_arithmetic<V>::value> probe_put_value(ProbeState& p, const V&) { p.failed = true; }   // vector values: full replay

template <class R> std::enable_if_t<std::is_arithmetic<R>::value, R> probe_from_double(double v, ProbeState&) { return static_cast<R>(v); }
template <class R> std::enable_if_t<!std::is_arithmetic<R>::value, R> probe_from_double(double, ProbeState& p) { p.failed = true; return R(); }
template <class V> std::enable_if_t<std::is_arithmetic<V>::value> probe_keep_value(ProbeState& p, std::size_t j, const V& v)
{
    if (p.values.size() <= j) p.values.resize(j + 1, 0.0);
    p.values[j] = static_cast<double>(v);
}
template <class V> std::enable_if_t<!std::is_arithmetic<V>::value> probe_keep_value(ProbeState& p, std::size_t, const V&) { p.failed = true; }

template <class Distribution>
auto host_sample(Distribution& distr)
{
    std::decay_t<Distribution> copy = distr;
    ProbeState& pb = probe();
    if (pb.active) {
        using R = decltype(copy(host_rng()));
        const std::size_t j = pb.ordinal++;
        if (j > pb.ordinal_cap) throw ProbeAbort{};                             // a loop that does not end on substituted values
        if (j < pb.dummy_below) return R();
        if (j < pb.replay_below && j < pb.values.size()) return probe_from_double<R>(pb.values[j], pb);      // the window: as the reference run drew it
        std::mt19937 g(pb.base_seed + 7919u * static_cast<std::uint32_t>(j));   // the draw depends on the ordinal and the distribution only
        R v = copy(g);
        probe_put_value(pb, v);
        if (pb.record_all) probe_keep_value(pb, j, v);
        return v;
    }
    auto value = copy(host_rng());                      // cpprob.hpp:33-35: distr(get_rng())
    if (TraceStructure* r = recorder()) {
        r->n_sample += std::is_arithmetic<decltype(value)>::value ? 1 : width_of_sized(value, 0);
        if (!std::is_arithmetic<decltype(value)>::value) r->vector_statements = true;
    }
    return value;
}

template <class Distribution, class X>
void host_observe(Distribution& distr, const X& x)
{
    ProbeState& pb = probe();
    if (pb.active) {
        pb.put(static_cast<double>(logpdf<std::decay_t<Distribution>>()(distr, x)));      // what the step adds to the weight
        BoundProbe& bp = bound_probe();
        if (bp.expected && (pb.n_obs >= bp.expected->size() || !same_bits((*bp.expected)[pb.n_obs], logpdf_max<std::decay_t<Distribution>>()(distr)))) bp.varies = true;
        ++pb.n_obs;
        return;
    }
    if (TraceStructure* r = recorder()) {
        r->samples_before_observe.push_back(r->n_sample); ++r->n_observe;
        r->observe_bound.push_back(logpdf_max<std::decay_t<Distribution>>()(distr));
    }
}

template <class V>
auto host_predict_dispatch(const V& x, const std::string& addr, int) -> decltype((void)x.size(), (void)x.begin(), void())
{
    if (probe().active) { probe().failed = true; return; }
    TraceStructure* r = recorder();                                                          // (a device-view NDArray: same bookkeeping as NDArray)
    if (!r) return;
    r->real_ids.push_back(r->id_of(addr));
    r->real_width.push_back(x.size());
    for (std::size_t d = 0; d < x.size(); ++d) r->real_row_step.push_back(static_cast<int>(r->n_observe));
    r->vector_statements = true;
}
template <class T>
void host_predict_dispatch(const T& x, const std::string& addr, long)
{
    if (probe().active) { probe_put_value(probe(), x); return; }
    TraceStructure* r = recorder();
    if (!r) return;
    using V = std::decay_t<T>;
    if (std::is_integral<V>::value) { r->int_ids.push_back(r->id_of(addr)); r->int_hit_step.push_back(static_cast<int>(r->n_observe)); }                  // state.hpp:312-318
    else if (std::is_floating_point<V>::value) { r->real_ids.push_back(r->id_of(addr)); r->real_width.push_back(1); r->real_row_step.push_back(static_cast<int>(r->n_observe)); }   // state.hpp:320-326
    else { r->id_of(addr); ++r->n_other_predicts; }
}
template <class T>
void host_predict(const T& x, const std::string& addr) { host_predict_dispatch(x, addr, 0); }
template <class T>
void host_predict(const NDArray<T>& x, const std::string& addr)                            // state.hpp:330-337: NDArray -> the real list
{
    if (probe().active) { probe().failed = true; return; }
    TraceStructure* r = recorder();
    if (!r) return;
    r->real_ids.push_back(r->id_of(addr));
    r->real_width.push_back(x.size());
    for (std::size_t d = 0; d < x.size(); ++d) r->real_row_step.push_back(static_cast<int>(r->n_observe));
    r->vector_statements = true;
}

}  // namespace detail
}  // namespace cpprob
#endif
