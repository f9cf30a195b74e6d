// Clean-room stand-in for boost/random/normal_distribution.hpp (Boost 1.66 is not vendored by the
// reference and is absent from this image).  Interface only -- what models and cpprob::logpdf use:
// result_type, param_type, mean(), sigma(), operator()(URNG&).  Usable from device code under hipcc.
#ifndef CPPROB_COMPAT_BOOST_RANDOM_NORMAL_DISTRIBUTION_HPP
#define CPPROB_COMPAT_BOOST_RANDOM_NORMAL_DISTRIBUTION_HPP
#include <cmath>
#include <limits>
#include "cpprob/detail/hd.hpp"

namespace boost { namespace random {

template <class RealType = double>
class normal_distribution {
public:
    using input_type = RealType;
    using result_type = RealType;
    struct param_type {
        using distribution_type = normal_distribution;
        CPPROB_HD explicit param_type(RealType mean = 0, RealType sigma = 1) : mean_(mean), sigma_(sigma) {}
        CPPROB_HD RealType mean() const { return mean_; }
        CPPROB_HD RealType sigma() const { return sigma_; }
    private:
        RealType mean_, sigma_;
    };
    CPPROB_HD explicit normal_distribution(const RealType& mean = 0, const RealType& sigma = 1) : mean_(mean), sigma_(sigma) {}
    CPPROB_HD explicit normal_distribution(const param_type& p) : mean_(p.mean()), sigma_(p.sigma()) {}
    CPPROB_HD RealType mean() const { return mean_; }
    CPPROB_HD RealType sigma() const { return sigma_; }
    CPPROB_HD RealType min() const { return -std::numeric_limits<RealType>::infinity(); }
    CPPROB_HD RealType max() const { return std::numeric_limits<RealType>::infinity(); }
    CPPROB_HD param_type param() const { return param_type(mean_, sigma_); }
    CPPROB_HD void reset() {}
    // Host generator form (Box-Muller on two 32-bit draws of any URNG with 32-bit output)
    template <class URNG>
    RealType operator()(URNG& g)
    {
        const double u = (static_cast<double>(g() - URNG::min()) + 1.0) / (static_cast<double>(URNG::max() - URNG::min()) + 2.0);
        const double v = (static_cast<double>(g() - URNG::min()) + 0.5) / (static_cast<double>(URNG::max() - URNG::min()) + 1.0);
        return mean_ + sigma_ * static_cast<RealType>(std::sqrt(-2.0 * std::log(u)) * std::sin(6.283185307179586476925 * v));
    }
private:
    RealType mean_, sigma_;
};

}  // namespace random
using random::normal_distribution;   // boost::normal_distribution (src/models/gaussian.cpp:10 uses this spelling)
}  // namespace boost
#endif
