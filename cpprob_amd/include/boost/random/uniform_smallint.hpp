// Clean-room stand-in for boost/random/uniform_smallint.hpp: interface only (a(), b(), min(), max()).
#ifndef CPPROB_COMPAT_BOOST_RANDOM_UNIFORM_SMALLINT_HPP
#define CPPROB_COMPAT_BOOST_RANDOM_UNIFORM_SMALLINT_HPP
#include "cpprob/detail/hd.hpp"

namespace boost { namespace random {

template <class IntType = int>
class uniform_smallint {
public:
    using input_type = IntType;
    using result_type = IntType;
    CPPROB_HD explicit uniform_smallint(IntType min = 0, IntType max = 9) : min_(min), max_(max) {}
    CPPROB_HD IntType a() const { return min_; }
    CPPROB_HD IntType b() const { return max_; }
    CPPROB_HD IntType min() const { return min_; }
    CPPROB_HD IntType max() const { return max_; }
    CPPROB_HD void reset() {}
    template <class URNG>
    IntType operator()(URNG& g)
    {
        const unsigned long long range = static_cast<unsigned long long>(max_ - min_) + 1ull;
        return min_ + static_cast<IntType>(static_cast<unsigned long long>(g() - URNG::min()) % range);
    }
private:
    IntType min_, max_;
};

}  // namespace random
using random::uniform_smallint;
}  // namespace boost
#endif
