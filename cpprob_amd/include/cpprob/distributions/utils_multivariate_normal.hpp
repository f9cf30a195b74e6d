// logpdf of the diagonal multivariate normal = sum of the components' normal logpdfs
// (reference include/cpprob/distributions/utils_multivariate_normal.hpp:20-33).  Host-only, like the
// distribution object; the device engine evaluates the same sum row by row (ModelGaussianND::loglik).
#ifndef CPPROB_COMPAT_UTILS_MULTIVARIATE_NORMAL_HPP
#define CPPROB_COMPAT_UTILS_MULTIVARIATE_NORMAL_HPP
#include "cpprob/distributions/multivariate_normal.hpp"
#include "cpprob/distributions/utils_distributions.hpp"

namespace cpprob {

template <class RealType>
struct logpdf<multivariate_normal_distribution<RealType>> {
    RealType operator()(const multivariate_normal_distribution<RealType>& distr,
                        const typename multivariate_normal_distribution<RealType>::result_type& x) const
    {
        RealType ret = 0;
        const auto comps = distr.distr();
        auto it = x.begin();
        for (auto c = comps.begin(); c != comps.end() && it != x.end(); ++c, ++it)
            ret += logpdf<boost::random::normal_distribution<RealType>>()(*c, *it);
        return ret;
    }
};

}  // namespace cpprob
#endif
