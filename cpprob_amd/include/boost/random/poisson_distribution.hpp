// Clean-room stand-in for boost/random/poisson_distribution.hpp: interface only (mean()).
#ifndef CPPROB_COMPAT_BOOST_RANDOM_POISSON_DISTRIBUTION_HPP
#define CPPROB_COMPAT_BOOST_RANDOM_POISSON_DISTRIBUTION_HPP
#include <cmath>
#include "cpprob/detail/hd.hpp"

namespace boost { namespace random {

template <class IntType = int, class RealType = double>
class poisson_distribution {
public:
    using input_type = RealType;
    using result_type = IntType;
    CPPROB_HD explicit poisson_distribution(RealType mean = 1) : mean_(mean) {}
    CPPROB_HD RealType mean() const { return mean_; }
    CPPROB_HD IntType min() const { return 0; }
    template <class URNG>
    IntType operator()(URNG& g)        // Knuth's multiplication method (host only; small means)
    {
        const double L = std::exp(-static_cast<double>(mean_));
        double p = 1.0; IntType k = 0;
        do { ++k; p *= (static_cast<double>(g() - URNG::min()) + 0.5) / (static_cast<double>(URNG::max() - URNG::min()) + 1.0); } while (p > L);
        return k - 1;
    }
private:
    RealType mean_;
};

}  // namespace random
using random::poisson_distribution;
}  // namespace boost
#endif
