// Device side of the statement API (hipcc only): what cpprob::sample / observe / predict do when
// the model body runs on a GPU lane.  Replaces the reference's process-global trace record
// (StateInfer::trace_, src/cpprob/state.cpp:148; TraceInfer, include/cpprob/trace.hpp:34-63) by a
// per-lane record in LDS (the mutable part) and the kernel-argument segment (the rest), one particle per lane.
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

// What every statement of a launch reads and none writes: the kernel's first argument.  A statement fetches the fields it needs
// straight from the kernel-argument segment (scalar loads into scalar registers, wherever in the call tree it sits), so they cost
// no LDS and no vector registers.  model_kernel (cpprob/gpu.hpp) takes this struct as its FIRST parameter.
struct LaunchArgs {
    int64_t n, ld;
    uint64_t seed;
    const int32_t* anc;            // ancestors of this generation (identity where the previous step did not resample); nullptr at step 0
    const int32_t* resampled_prev; // device flag: did the previous step resample?  (the decision is taken on the device)
    const double* logw_in;         // log-weights of the previous generation, carried over when it was not resampled
    double* logw_out;
    const uint64_t* trace_in; uint64_t* trace_out;       // stored sample values, row stride ld: the ancestor's / this lane's new column
    const int32_t* nstored_in; int32_t* nstored_out;
    double* pred_real; int32_t* pred_int;                 // predict columns, row stride ld (nullptr: do not record)
    int32_t first_observe;         // observes with a smaller index were weighted in earlier steps
    int32_t stop_after;            // index of the observe that ends this step (-1: run to completion)
    uint32_t trace_cap;            // rows of the trace buffers; a longer trace raises *overflow (rejection loops)
    int32_t* overflow;
    uint32_t pred_real_cap, pred_int_cap;                 // predict columns the host allocated (hits seen by the structural dry run)
    // windowed replay (models that passed the Markov probe): a step replays only the ancestor's last `win` samples, kept per
    // particle in `carry` rows (slot k <-> ordinal fresh_lo - win + k); older samples come back value-initialised, never from memory
    uint32_t windowed, win;
    int32_t fresh_lo;              // first ordinal this step draws itself (= samples before the previous observe)
    int32_t next_fresh;            // first ordinal the NEXT step draws (= samples before this step's observe): what carry_out must end with
    const uint64_t* carry_in; uint64_t* carry_out;
    uint64_t pid0;                 // global id of lane 0's particle (shards of one population draw from the population's streams)
};
typedef const LaunchArgs __attribute__((address_space(4))) * LaunchArgsPtr;
__device__ inline LaunchArgsPtr launch_args() { return (LaunchArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(); }

// What a lane's statements do write: counters, the running log-weight, the frontier flag.  One workgroup's lanes side by side per
// field (a wavefront's access to a field is one conflict-free LDS instruction); 40 bytes per lane, so the LDS does not limit how
// many wavefronts a CU holds (the 176-byte per-lane record of the first build allowed 3 per SIMD, and every statement's chain of
// dependent LDS round trips went unhidden: 49 us per launch at 10^6 particles).
struct LaneState {
    double log_w[kLaneBlock];
    double carried[kLaneBlock];           // the log-weight the particle brought along (no resampling before this step)
    uint32_t n_sample[kLaneBlock], n_observe[kLaneBlock], n_pred_real[kLaneBlock], n_pred_int[kLaneBlock];
    uint32_t n_recorded[kLaneBlock];      // samples executed before the lane was done
    uint32_t n_stored[kLaneBlock];        // samples available in the ancestor's trace column
    uint32_t done[kLaneBlock];
    int32_t src[kLaneBlock];              // the lane's ancestor (its own index where the previous step did not resample)
    unsigned long long active[kLaneBlock / 64];   // per wavefront: the lanes that carry a particle
    // Windowed replay (the model passed the Markov probe: statement counts do not depend on sampled values): every lane of a
    // wavefront executes the same statements, so the counters are WAVE-UNIFORM -- one packed word per lane, read back through the
    // scalar unit (readfirstlane), and every test on it a scalar branch: a statement behind or before the live window costs one LDS
    // round trip and a handful of scalar instructions instead of a chain of execution-mask regions.
    // bits 0..23 samples, 24..39 observes, 40..49 int predicts, 50..59 real predicts, 63 done
    unsigned long long w[kLaneBlock];
};
constexpr unsigned long long kWSample = 1ull, kWObserve = 1ull << 24, kWPredInt = 1ull << 40, kWPredReal = 1ull << 50, kWDone = 1ull << 63;
__device__ inline unsigned long long wave_uniform(unsigned long long v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (unsigned long long)lo | ((unsigned long long)hi << 32);
}
__device__ inline LaneState& lane_state()
{
    __shared__ LaneState s_state;
    return s_state;
}
__device__ inline int64_t lane_index() { return (int64_t)blockIdx.x * kLaneBlock + threadIdx.x; }     // = the particle id

__device__ inline void begin_lane(int32_t src, uint32_t n_stored, double carried)
{
    LaneState& s = lane_state();
    const int l = threadIdx.x;
    s.log_w[l] = 0.0; s.carried[l] = carried;
    s.n_sample[l] = 0; s.n_observe[l] = 0; s.n_pred_real[l] = 0; s.n_pred_int[l] = 0; s.n_recorded[l] = 0;
    s.n_stored[l] = n_stored; s.done[l] = 0; s.src[l] = src; s.w[l] = 0;
    s.active[l / 64] = __ballot(1);        // (every lane of the wavefront writes the same word)
}

// finish_trace() of one lane: the particle's log-weight (and, full replay, how many samples its trace holds)
__device__ inline void finish_lane()
{
    LaunchArgsPtr A = launch_args();
    const LaneState& s = lane_state();
    const int l = threadIdx.x;
    const int64_t i = lane_index();
    A->logw_out[i] = s.carried[l] + s.log_w[l];
    if (A->nstored_out) A->nstored_out[i] = (int32_t)s.n_recorded[l];
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
    LaunchArgsPtr A = launch_args();
    LaneState& s = lane_state();
    const int l = threadIdx.x;
    if constexpr (is_dev_mvn<std::decay_t<Distribution>>::value) {
        // vector-valued sample (multivariate_normal.hpp:268-274): the components draw one after the other, each its own trace slot
        using T = typename R::value_type;
        R value;
        for (std::size_t i = 0; i < distr.size(); ++i) {
            const uint32_t j = s.n_sample[l]++;
            T v = T();
            if (!s.done[l]) {
                if (j < s.n_stored[l] && j < A->trace_cap) v = from_raw<T>(A->trace_in[(int64_t)j * A->ld + s.src[l]]);
                else v = static_cast<T>(distr.mean_at(i) + distr.sigma_at(i) * cph::draw_std_normal(A->seed, (uint64_t)lane_index() + A->pid0, (uint64_t)j));
                if (A->trace_out) {
                    if (j < A->trace_cap) A->trace_out[(int64_t)j * A->ld + lane_index()] = to_raw<T>(v);
                    else if (A->overflow) *A->overflow = 1;
                }
                s.n_recorded[l] = j + 1;
            }
            value.push_back(v);
        }
        return value;
    } else if constexpr (!std::is_arithmetic<R>::value) {
        // vector-valued statements own host containers: they reach the device through built-in kernels only
        // (cpprob::inference refuses to launch this path for them, host_engine.hpp)
        __builtin_trap();
    } else {
        if (A->windowed) {
            // (scalar control flow: see LaneState::w)
            const unsigned long long w = wave_uniform(s.w[l]);
            s.w[l] = w + kWSample;
            if (w & kWDone) return R();
            const int32_t jj = (int32_t)(w & 0xffffffu), base = A->fresh_lo - (int32_t)A->win;
            if (jj < base) return R();                                    // older than the window: the step does not depend on it (host probe)
            R v;
            if (jj < A->fresh_lo) v = from_raw<R>(A->carry_in[(int64_t)(jj - base) * A->ld + s.src[l]]);
            else v = draw(distr, A->seed, (uint64_t)lane_index() + A->pid0, (uint64_t)jj);
            const int32_t out_base = A->next_fresh - (int32_t)A->win;
            if (A->carry_out && jj >= out_base && jj < A->next_fresh) A->carry_out[(int64_t)(jj - out_base) * A->ld + lane_index()] = to_raw<R>(v);
            return v;
        }
        const uint32_t j = s.n_sample[l]++;
        if (s.done[l]) return R();
        R value;
        if (j < s.n_stored[l] && j < A->trace_cap) value = from_raw<R>(A->trace_in[(int64_t)j * A->ld + s.src[l]]);   // (beyond the rows: the run is being repeated anyway)
        else value = draw(distr, A->seed, (uint64_t)lane_index() + A->pid0, (uint64_t)j);
        if (A->trace_out) {
            if (j < A->trace_cap) A->trace_out[(int64_t)j * A->ld + lane_index()] = to_raw<R>(value);
            else if (A->overflow) *A->overflow = 1;
        }
        s.n_recorded[l] = j + 1;
        return value;
    }
}

template <class Distribution, class X>
__device__ inline void observe_impl(Distribution& distr, const X& x)
{
    constexpr bool vec = is_dev_mvn<std::decay_t<Distribution>>::value;
    if constexpr (!vec && !std::is_arithmetic<X>::value) {
        __builtin_trap();                                           // heap-backed vector types: compile the model's device view (cpprob/gpu.hpp)
    } else {
        // (vector-valued: ONE observe statement whose log-density is the sum over the components, utils_multivariate_normal.hpp:20-33)
        LaunchArgsPtr A = launch_args();
        LaneState& s = lane_state();
        const int l = threadIdx.x;
        if (A->windowed) {
            const unsigned long long own = s.w[l];
            const unsigned long long w = wave_uniform(own);
            s.w[l] = w + kWObserve;
            // (a lane whose counters differ from its wavefront's executed other statements: the counts DO depend on sampled values,
            //  the probe notwithstanding -- reported, and the host repeats the run with full replay)
            if (__ballot(own != w) != 0ull && A->overflow) *A->overflow = 3;
            const int32_t m = (int32_t)((w >> 24) & 0xffffu);
            if (!(w & kWDone) && m >= A->first_observe) {
                s.log_w[l] += logpdf<std::decay_t<Distribution>>()(distr, x);     // StateInfer::increment_log_prob, state.cpp:212-223
                if (m == A->stop_after) { finish_lane(); asm volatile("s_endpgm" ::: "memory"); }      // every lane of the wavefront is here
            }
            return;
        }
        const int32_t m = (int32_t)s.n_observe[l]++;
        if (!s.done[l] && m >= A->first_observe) {
            s.log_w[l] += logpdf<std::decay_t<Distribution>>()(distr, x);     // StateInfer::increment_log_prob, state.cpp:212-223
            if (m == A->stop_after) s.done[l] = 1;
        }
        // Every statement behind the frontier is a no-op: once ALL of the wavefront's lanes have passed it (models whose observes do
        // not depend on sampled values: all at this very statement) the wavefront is finished.  The test is wave-uniform (all lanes
        // or none enter the block); the exit is written as an instruction rather than the builtin, which the compiler would turn
        // into lane masking when it cannot prove the surrounding control flow uniform.
        if (A->stop_after >= 0) {
            const unsigned long long arrived = __ballot(s.done[l] != 0);
            if (__builtin_amdgcn_readfirstlane(arrived == s.active[threadIdx.x / 64] ? 1 : 0)) { finish_lane(); asm volatile("s_endpgm" ::: "memory"); }
        }
    }
}

template <class T>
__device__ inline void predict_impl(const T& x)
{
    using V = std::decay_t<T>;
    LaunchArgsPtr A = launch_args();
    LaneState& s = lane_state();
    const int l = threadIdx.x;
    if constexpr (std::is_integral<V>::value || std::is_floating_point<V>::value) {
        if (A->windowed) {
            constexpr bool is_int = std::is_integral<V>::value;
            const unsigned long long w = wave_uniform(s.w[l]);
            if (w & kWDone) return;
            s.w[l] = w + (is_int ? kWPredInt : kWPredReal);
            if ((int32_t)((w >> 24) & 0xffffu) < A->first_observe) return;           // an earlier step's hit: recorded by that step's launch
            const uint32_t k = (uint32_t)((w >> (is_int ? 40 : 50)) & 0x3ffu);
            if constexpr (is_int) {
                if (A->pred_int) { if (k < A->pred_int_cap) A->pred_int[(int64_t)k * A->ld + lane_index()] = static_cast<int32_t>(x); else if (A->overflow) *A->overflow = 2; }
            } else {
                if (A->pred_real) { if (k < A->pred_real_cap) A->pred_real[(int64_t)k * A->ld + lane_index()] = static_cast<double>(x); else if (A->overflow) *A->overflow = 2; }
            }
            return;
        }
    }
    if constexpr (std::is_integral<V>::value) {                         // state.hpp:312-318 -> predict_int_
        if (s.done[l]) return;
        const uint32_t k = s.n_pred_int[l]++;
        if (A->windowed && (int32_t)s.n_observe[l] < A->first_observe) return;      // an earlier step's hit: recorded by that step's launch
        if (A->pred_int) {
            if (k < A->pred_int_cap) A->pred_int[(int64_t)k * A->ld + lane_index()] = static_cast<int32_t>(x);
            else if (A->overflow) *A->overflow = 2;                     // more predict hits than the dry run: data-dependent predicts
        }
    } else if constexpr (std::is_floating_point<V>::value) {            // state.hpp:320-326 -> predict_real_
        if (s.done[l]) return;
        const uint32_t k = s.n_pred_real[l]++;
        if (A->windowed && (int32_t)s.n_observe[l] < A->first_observe) return;
        if (A->pred_real) {
            if (k < A->pred_real_cap) A->pred_real[(int64_t)k * A->ld + lane_index()] = static_cast<double>(x);
            else if (A->overflow) *A->overflow = 2;
        }
    } else if constexpr (is_dev_ndarray<V>::value) {                    // state.hpp:330-337: an NDArray goes to the real list; one column per component
        if (s.done[l]) return;
        for (std::size_t i = 0; i < x.size(); ++i) {
            const uint32_t k = s.n_pred_real[l]++;
            if (A->pred_real) {
                if (k < A->pred_real_cap) A->pred_real[(int64_t)k * A->ld + lane_index()] = static_cast<double>(x[i]);
                else if (A->overflow) *A->overflow = 2;
            }
        }
    } else {
        __builtin_trap();                                               // heap-backed vector types: compile the model's device view (cpprob/gpu.hpp)
    }
}

}  // namespace device
}  // namespace cpprob
#endif
