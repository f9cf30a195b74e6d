// cpprob.hpp -- the statement API and the inference driver, same names and signatures as the reference
// include/cpprob/cpprob.hpp (sample :28-76, observe :79-90, predict :92-106, inference :173-203), so
// that model sources written against CPProb compile unchanged.  What differs is where the particle
// loop runs: cpprob::inference hands the whole population to the MI355X engine (libcpprob_hip.so).
//
//   * compiled by a C++14 host compiler: the statements only serve the one-trace structural dry run;
//   * compiled by hipcc (C++17): the same statements are also device functions acting on the calling
//     lane's particle (cpprob/detail/device_trace.hpp), which is how an unchanged model body becomes
//     a kernel (cpprob/gpu.hpp).
// StateType::smc is new.  compile / csis (the NN "inference compilation" modes) are out of scope:
// inference() throws std::runtime_error for them.
#ifndef CPPROB_COMPAT_CPPROB_HPP
#define CPPROB_COMPAT_CPPROB_HPP

#include <array>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <utility>

#include <boost/filesystem/path.hpp>

#include "cpprob/detail/hd.hpp"
#include "cpprob/detail/host_engine.hpp"
#include "cpprob/detail/host_trace.hpp"
#include "cpprob/detail/traits.hpp"
#include "cpprob/distributions/utils_distributions.hpp"
#include "cpprob/distributions/utils_multivariate_normal.hpp"
#include "cpprob/state.hpp"
#if defined(CPPROB_DEVICE_COMPILE_AVAILABLE)
#include "cpprob/detail/device_trace.hpp"
#endif

// The statements are part of the model's kernel, not functions it calls: the launch's mode reaches them as compile-time facts of the
// kernel they are inlined into (cpprob/gpu.hpp: __builtin_assume), their counters live in its registers, and the step's observe ends
// its wavefront.  Whatever the optimisation level says about inlining (at -O3 hipcc left them as calls), they are inlined.
#define CPPROB_STATEMENT __attribute__((always_inline))

namespace cpprob {

template <class Distribution, class String>
CPPROB_HD CPPROB_STATEMENT auto sample(Distribution&& distr, const bool control, String&& address) -> typename std::decay_t<Distribution>::result_type
{
    (void)control; (void)address;
#if defined(__HIP_DEVICE_COMPILE__)
    return device::sample_impl(distr);
#else
    return detail::host_sample(distr);
#endif
}

template <class Distribution>
CPPROB_HD CPPROB_STATEMENT auto sample(Distribution&& distr, const bool control = false) -> typename std::decay_t<Distribution>::result_type
{
    (void)control;
#if defined(__HIP_DEVICE_COMPILE__)
    return device::sample_impl(distr);
#else
    return detail::host_sample(distr);
#endif
}

template <class Distribution>
CPPROB_HD CPPROB_STATEMENT void observe(Distribution&& distr, const typename std::decay_t<Distribution>::result_type& x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    device::observe_impl(distr, x);
#else
    detail::host_observe(distr, x);
#endif
}

template <class T, class String>
CPPROB_HD CPPROB_STATEMENT void predict(T&& x, String&& addr)
{
#if defined(__HIP_DEVICE_COMPILE__)
    (void)addr;
    device::predict_impl(x);
#else
    detail::host_predict(x, std::string(addr));
#endif
}

template <class T>
CPPROB_HD CPPROB_STATEMENT void predict(T&& x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    device::predict_impl(x);
#else
    // the reference derives the address from the call stack (utils.cpp:71-128): one per textual call site
    detail::host_predict(x, detail::recorder() ? detail::call_site_address() : std::string());
#endif
}

template <class T> CPPROB_HD void metaobserve(T&&) {}

struct rejection_sampling {
    CPPROB_HD rejection_sampling() {}
    rejection_sampling& operator=(const rejection_sampling&) = delete;
    rejection_sampling& operator=(rejection_sampling&&) = delete;
};
inline void start_rejection_sampling() {}
inline void finish_rejection_sampling() {}

template <class Func, class... Args>
void inference(const StateType algorithm, const Func& f, const std::tuple<Args...>& observes, std::size_t n = 50'000,
               const boost::filesystem::path& file_name = "posterior", const std::string& tcp_addr = "tcp://127.0.0.1:6666")
{
    static_assert(sizeof...(Args) != 0, "The function has to receive the observed values as parameters.");   // cpprob.hpp:182
    (void)tcp_addr;
    if (algorithm != StateType::sis && algorithm != StateType::smc)
        throw std::runtime_error("cpprob::inference: only StateType::sis and StateType::smc run on the device engine "
                                 "(compile / csis need the inference-compilation network, which is out of scope)");
    State::set(algorithm);
    gpu::run_inference(algorithm, f, observes, n, file_name.string());
}

}  // namespace cpprob
#endif
