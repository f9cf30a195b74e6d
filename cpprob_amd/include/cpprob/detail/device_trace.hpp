// Device side of the statement API (hipcc only): what cpprob::sample / observe / predict do when
// the model body runs on a GPU lane.  Replaces the reference's process-global trace record
// (StateInfer::trace_, src/cpprob/state.cpp:148; TraceInfer, include/cpprob/trace.hpp:34-63) by a
// per-lane record in LDS, one particle per lane.
//
//   sample  #j : SIS -> a fresh draw (Philox block of (particle id, ordinal j), cpprob/detail/rng.hpp);
//                SMC step t -> the stored value of the lane's ancestor for j < n_stored (trace replay),
//                a fresh draw otherwise; every value is written to the lane's new trace column.
//   observe #m : log_w += logpdf for first_observe <= m; after observe #stop_after the lane is `done`
//                and every later statement is a no-op (SMC stops all particles at the same observe).
//   predict    : k-th real / int hit -> column k of the particle store (only when the pointers are set).
#ifndef CPPROB_COMPAT_DETAIL_DEVICE_TRACE_HPP
#define CPPROB_COMPAT_DETAIL_DEVICE_TRACE_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include <boost/random/discrete_distribution.hpp>
#include <boost/random/normal_distribution.hpp>
#include <boost/random/poisson_distribution.hpp>
#include <boost/random/uniform_real_distribution.hpp>
#include <boost/random/uniform_smallint.hpp>

#include "cpprob/detail/device_vector.hpp"
#include "cpprob/detail/rng.hpp"
#include "cpprob/distributions/utils_distributions.hpp"

namespace cpprob {
namespace device {

constexpr int kLaneBlock = 256;

struct LaneCtx {
    uint64_t seed, pid;
    double log_w;
    const uint64_t* trace_in;     // ancestor's stored sample values, stride ld (nullptr: none)
    uint64_t* trace_out;          // this lane's new trace column, stride ld (nullptr: do not record)
    double* pred_real;            // this lane's predict columns, stride ld (nullptr: do not record)
    int32_t* pred_int;
    int64_t ld;
    uint32_t n_sample, n_observe, n_pred_real, n_pred_int;
    uint32_t n_stored;            // samples available in trace_in
    uint32_t trace_cap;           // rows of the trace buffers; a longer trace raises *overflow (rejection loops)
    uint32_t pred_real_cap, pred_int_cap;   // predict columns the host allocated (hits seen by the structural dry run)
    int32_t* overflow;
    uint32_t n_recorded;          // samples executed before the lane was done
    int32_t first_observe;        // observes with a smaller index were weighted in earlier steps
    int32_t stop_after;           // index of the observe that ends this step (-1: run to completion)
    uint32_t done;
    // windowed replay (models that passed the Markov probe): a step replays only the ancestor's last `win` samples, kept per
    // particle in `carry` rows (slot k <-> ordinal fresh_lo - win + k); older samples come back value-initialised, never from memory
    uint32_t windowed, win;
    int32_t fresh_lo;             // first ordinal this step draws itself (= samples before the previous observe)
    int32_t next_fresh;           // first ordinal the NEXT step draws (= samples before this step's observe): what carry_out must end with
    const uint64_t* carry_in; uint64_t* carry_out;
    // what the kernel's epilogue needs
    double carried; double* logw_out; int32_t* nstored_out;
};

// finish_trace() of one lane: the particle's log-weight (and, full replay, how many samples its trace holds)
__device__ inline void finish_lane(LaneCtx& c)
{
    *c.logw_out = c.carried + c.log_w;
    if (c.nstored_out) *c.nstored_out = (int32_t)c.n_recorded;
}

__device__ inline LaneCtx& lane_ctx()
{
    __shared__ LaneCtx s_ctx[kLaneBlock];
    return s_ctx[threadIdx.x];
}

// 8-byte raw slots of the sample trace
template <class T> __device__ inline uint64_t to_raw(T v)
{
    if (std::is_floating_point<T>::value) { const double d = static_cast<double>(v); return __double_as_longlong(d); }
    return static_cast<uint64_t>(static_cast<int64_t>(v));
}
template <class T> __device__ inline T from_raw(uint64_t r)
{
    if (std::is_floating_point<T>::value) return static_cast<T>(__longlong_as_double(static_cast<long long>(r)));
    return static_cast<T>(static_cast<int64_t>(r));
}

// ---- variate generators: stand-ins for boost::random::X::operator()(get_rng()) -------------------------
template <class R>
__device__ inline R draw(const boost::random::normal_distribution<R>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    return static_cast<R>(d.mean() + d.sigma() * cph::draw_std_normal(seed, pid, j));
}
template <class I>
__device__ inline I draw(const boost::random::uniform_smallint<I>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    return static_cast<I>(d.a() + static_cast<I>(cph::smallint_from_word(cph::draw_word(seed, pid, j), 0, static_cast<uint64_t>(d.b() - d.a()))));
}
template <class I, class W>
__device__ inline I draw(const boost::random::discrete_distribution<I, W>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    const auto& w = d.weights();
    return static_cast<I>(cph::discrete_from_u_dyn(cph::u01_32(cph::draw_word(seed, pid, j)), w.p, static_cast<int>(w.n)));
}
template <class R>
__device__ inline R draw(const boost::random::uniform_real_distribution<R>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    return static_cast<R>(cph::draw_uniform_real(seed, pid, j, d.a(), d.b()));
}

template <class I, class R>
__device__ inline I draw(const boost::random::poisson_distribution<I, R>& d, uint64_t seed, uint64_t pid, uint64_t j)
{
    return static_cast<I>(cph::draw_poisson(seed, pid, j, static_cast<double>(d.mean())));
}

template <class Distribution>
__device__ inline typename std::decay_t<Distribution>::result_type sample_impl(Distribution& distr)
{
    using R = typename std::decay_t<Distribution>::result_type;
    if constexpr (is_dev_mvn<std::decay_t<Distribution>>::value) {
        // vector-valued sample (multivariate_normal.hpp:268-274): the components draw one after the other, each its own trace slot
        using T = typename R::value_type;
        LaneCtx& c = lane_ctx();
        R value;
        for (std::size_t i = 0; i < distr.size(); ++i) {
            const uint32_t j = c.n_sample++;
            T v = T();
            if (!c.done) {
                if (j < c.n_stored && j < c.trace_cap) v = from_raw<T>(c.trace_in[(int64_t)j * c.ld]);
                else v = static_cast<T>(distr.mean_at(i) + distr.sigma_at(i) * cph::draw_std_normal(c.seed, c.pid, (uint64_t)j));
                if (c.trace_out) {
                    if (j < c.trace_cap) c.trace_out[(int64_t)j * c.ld] = to_raw<T>(v);
                    else if (c.overflow) *c.overflow = 1;
                }
                c.n_recorded = j + 1;
            }
            value.push_back(v);
        }
        return value;
    } else if constexpr (!std::is_arithmetic<R>::value) {
        // vector-valued statements own host containers: they reach the device through built-in kernels only
        // (cpprob::inference refuses to launch this path for them, host_engine.hpp)
        __builtin_trap();
    } else {
        LaneCtx& c = lane_ctx();
        const uint32_t j = c.n_sample++;
        if (c.done) return R();
        if (c.windowed) {
            const int32_t jj = (int32_t)j, base = c.fresh_lo - (int32_t)c.win;
            if (jj < base) return R();                                    // older than the window: the step does not depend on it (host probe)
            R v;
            if (jj < c.fresh_lo) v = from_raw<R>(c.carry_in[(int64_t)(jj - base) * c.ld]);
            else v = draw(distr, c.seed, c.pid, (uint64_t)j);
            const int32_t out_base = c.next_fresh - (int32_t)c.win;
            if (c.carry_out && jj >= out_base && jj < c.next_fresh) c.carry_out[(int64_t)(jj - out_base) * c.ld] = to_raw<R>(v);
            c.n_recorded = j + 1;
            return v;
        }
        R value;
        if (j < c.n_stored && j < c.trace_cap) value = from_raw<R>(c.trace_in[(int64_t)j * c.ld]);   // (beyond the rows: the run is being repeated anyway)
        else value = draw(distr, c.seed, c.pid, (uint64_t)j);
        if (c.trace_out) {
            if (j < c.trace_cap) c.trace_out[(int64_t)j * c.ld] = to_raw<R>(value);
            else if (c.overflow) *c.overflow = 1;
        }
        c.n_recorded = j + 1;
        return value;
    }
}

template <class Distribution, class X>
__device__ inline void observe_impl(Distribution& distr, const X& x)
{
    if constexpr (is_dev_mvn<std::decay_t<Distribution>>::value) {
        // ONE observe statement whose log-density is the sum over the components (utils_multivariate_normal.hpp:20-33)
        LaneCtx& c = lane_ctx();
        const int32_t m = (int32_t)c.n_observe++;
        if (c.done || m < c.first_observe) return;
        c.log_w += logpdf<std::decay_t<Distribution>>()(distr, x);
        if (m == c.stop_after) c.done = 1;
    } else if constexpr (!std::is_arithmetic<X>::value) {
        __builtin_trap();                                           // heap-backed vector types: compile the model's device view (cpprob/gpu.hpp)
    } else {
        LaneCtx& c = lane_ctx();
        const int32_t m = (int32_t)c.n_observe++;
        if (c.done || m < c.first_observe) return;
        c.log_w += logpdf<std::decay_t<Distribution>>()(distr, x);     // StateInfer::increment_log_prob, state.cpp:212-223
        if (m == c.stop_after) c.done = 1;
        // (ending the wave here once every lane has arrived -- finish_lane + s_endpgm -- was tried: no gain, the statements after the
        //  frontier are the cheap ones: profiles/r02_notes.md)
    }
}

template <class T>
__device__ inline void predict_impl(const T& x)
{
    using V = std::decay_t<T>;
    if constexpr (std::is_integral<V>::value) {                         // state.hpp:312-318 -> predict_int_
        LaneCtx& c = lane_ctx();
        if (c.done) return;
        const uint32_t k = c.n_pred_int++;
        if (c.windowed && (int32_t)c.n_observe < c.first_observe) return;      // an earlier step's hit: recorded by that step's launch
        if (c.pred_int) {
            if (k < c.pred_int_cap) c.pred_int[(int64_t)k * c.ld] = static_cast<int32_t>(x);
            else if (c.overflow) *c.overflow = 2;                       // more predict hits than the dry run: data-dependent predicts
        }
    } else if constexpr (std::is_floating_point<V>::value) {            // state.hpp:320-326 -> predict_real_
        LaneCtx& c = lane_ctx();
        if (c.done) return;
        const uint32_t k = c.n_pred_real++;
        if (c.windowed && (int32_t)c.n_observe < c.first_observe) return;
        if (c.pred_real) {
            if (k < c.pred_real_cap) c.pred_real[(int64_t)k * c.ld] = static_cast<double>(x);
            else if (c.overflow) *c.overflow = 2;
        }
    } else if constexpr (is_dev_ndarray<V>::value) {                    // state.hpp:330-337: an NDArray goes to the real list; one column per component
        LaneCtx& c = lane_ctx();
        if (c.done) return;
        for (std::size_t i = 0; i < x.size(); ++i) {
            const uint32_t k = c.n_pred_real++;
            if (c.pred_real) {
                if (k < c.pred_real_cap) c.pred_real[(int64_t)k * c.ld] = static_cast<double>(x[i]);
                else if (c.overflow) *c.overflow = 2;
            }
        }
    } else {
        __builtin_trap();                                               // heap-backed vector types: compile the model's device view (cpprob/gpu.hpp)
    }
}

}  // namespace device
}  // namespace cpprob
#endif
