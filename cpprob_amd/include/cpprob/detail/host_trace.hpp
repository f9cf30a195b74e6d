// Host side of the statement API: one structural dry run of the model (what StateType::dryrun does in
// the reference, src/main.cpp:108-112) records how many sample / observe / predict statements a trace
// executes, the type and address of every predict hit, and nothing else.  No inference ever runs on
// the host: the particle loop lives on the device.
#ifndef CPPROB_COMPAT_DETAIL_HOST_TRACE_HPP
#define CPPROB_COMPAT_DETAIL_HOST_TRACE_HPP
#include <cstddef>
#include <random>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

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

inline TraceStructure*& recorder() { static thread_local TraceStructure* r = nullptr; return r; }
inline std::mt19937& host_rng() { static thread_local std::mt19937 rng{20260101u}; return rng; }

// variates a value consumes: one per component
template <class T> std::size_t width_of(const T&) { return 1; }
template <class T> std::size_t width_of(const NDArray<T>& x) { return x.size(); }

template <class Distribution>
auto host_sample(Distribution& distr)
{
    std::decay_t<Distribution> copy = distr;
    auto value = copy(host_rng());                      // cpprob.hpp:33-35: distr(get_rng())
    if (TraceStructure* r = recorder()) {
        r->n_sample += width_of(value);
        if (!std::is_arithmetic<decltype(value)>::value) r->vector_statements = true;
    }
    return value;
}

inline void host_observe() { if (recorder()) ++recorder()->n_observe; }

template <class T>
void host_predict(const T&, const std::string& addr)
{
    TraceStructure* r = recorder();
    if (!r) return;
    using V = std::decay_t<T>;
    if (std::is_integral<V>::value) r->int_ids.push_back(r->id_of(addr));                  // state.hpp:312-318
    else if (std::is_floating_point<V>::value) { r->real_ids.push_back(r->id_of(addr)); r->real_width.push_back(1); }   // state.hpp:320-326
    else { r->id_of(addr); ++r->n_other_predicts; }
}
template <class T>
void host_predict(const NDArray<T>& x, const std::string& addr)                            // state.hpp:330-337: NDArray -> the real list
{
    TraceStructure* r = recorder();
    if (!r) return;
    r->real_ids.push_back(r->id_of(addr));
    r->real_width.push_back(x.size());
    r->vector_statements = true;
}

}  // namespace detail
}  // namespace cpprob
#endif
