// Clean-room stand-in for boost/random/uniform_real_distribution.hpp: interface only.
#ifndef CPPROB_COMPAT_BOOST_RANDOM_UNIFORM_REAL_DISTRIBUTION_HPP
#define CPPROB_COMPAT_BOOST_RANDOM_UNIFORM_REAL_DISTRIBUTION_HPP
#include "cpprob/detail/hd.hpp"

namespace boost { namespace random {

template <class RealType = double>
class uniform_real_distribution {
public:
    using input_type = RealType;
    using result_type = RealType;
    CPPROB_HD explicit uniform_real_distribution(RealType min = 0, RealType max = 1) : min_(min), max_(max) {}
    CPPROB_HD RealType a() const { return min_; }
    CPPROB_HD RealType b() const { return max_; }
    CPPROB_HD RealType min() const { return min_; }
    CPPROB_HD RealType max() const { return max_; }
    CPPROB_HD void reset() {}
    template <class URNG>
    RealType operator()(URNG& g)
    {
        const double u = static_cast<double>(g() - URNG::min()) / (static_cast<double>(URNG::max() - URNG::min()) + 1.0);
        return min_ + (max_ - min_) * static_cast<RealType>(u);
    }
private:
    RealType min_, max_;
};

}  // namespace random
using random::uniform_real_distribution;
}  // namespace boost
#endif
