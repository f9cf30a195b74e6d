// logpdf<D> functors for the Boost-compatible distributions -- same names, same arithmetic and same
// branch order as reference include/cpprob/distributions/utils_{normal_distribution,uniform_smallint,
// discrete,uniform_real,poisson}.hpp.  Callable from host and device (cpprob/detail/dist.hpp holds
// the arithmetic; the built-in kernels call the same functions).
#ifndef CPPROB_COMPAT_UTILS_DISTRIBUTIONS_HPP
#define CPPROB_COMPAT_UTILS_DISTRIBUTIONS_HPP

#include <cstdint>

#include <boost/random/discrete_distribution.hpp>
#include <boost/random/normal_distribution.hpp>
#include <boost/random/poisson_distribution.hpp>
#include <boost/random/uniform_real_distribution.hpp>
#include <boost/random/uniform_smallint.hpp>

#include "cpprob/detail/dist.hpp"
#include "cpprob/detail/hd.hpp"
#include "cpprob/distributions/utils_base.hpp"

namespace cpprob {

template <class RealType>
struct logpdf<boost::random::normal_distribution<RealType>> {      // utils_normal_distribution.hpp:20-45
    CPPROB_HD RealType operator()(const boost::random::normal_distribution<RealType>& distr, const RealType& x) const
    { return static_cast<RealType>(cph::normal_logpdf(x, distr.mean(), distr.sigma())); }
};

template <class IntType>
struct logpdf<boost::random::uniform_smallint<IntType>> {          // utils_uniform_smallint.hpp:17-27
    CPPROB_HD double operator()(const boost::random::uniform_smallint<IntType>& distr, const IntType& x) const
    { return cph::uniform_smallint_logpdf(static_cast<int64_t>(x), static_cast<int64_t>(distr.min()), static_cast<int64_t>(distr.max())); }
};

template <class IntType, class WeightType>
struct logpdf<boost::random::discrete_distribution<IntType, WeightType>> {   // utils_discrete.hpp:17-27
    CPPROB_HD WeightType operator()(const boost::random::discrete_distribution<IntType, WeightType>& distr, const IntType& x) const
    {
        if (x < distr.min() || x > distr.max()) return -INFINITY;
        return log(distr.probabilities()[static_cast<std::size_t>(x)]);
    }
};

template <class RealType>
struct logpdf<boost::random::uniform_real_distribution<RealType>> {   // utils_uniform_real.hpp:21-31
    CPPROB_HD RealType operator()(const boost::random::uniform_real_distribution<RealType>& distr, const RealType& x) const
    { return static_cast<RealType>(cph::uniform_real_logpdf(x, distr.a(), distr.b())); }
};

template <class IntType, class RealType>
struct logpdf<boost::random::poisson_distribution<IntType, RealType>> {   // utils_poisson.hpp:17-36
    CPPROB_HD RealType operator()(const boost::random::poisson_distribution<IntType, RealType>& distr, const IntType& x) const
    { return static_cast<RealType>(cph::poisson_logpdf(static_cast<int64_t>(x), distr.mean())); }
};

// ---- the densities' maxima (utils_base.hpp: logpdf_max) -- each the functor above evaluated at the distribution's mode ----
template <class RealType>
struct logpdf_max<boost::random::normal_distribution<RealType>> {
    double operator()(const boost::random::normal_distribution<RealType>& d) const
    { return static_cast<double>(logpdf<boost::random::normal_distribution<RealType>>()(d, d.mean())); }
};
template <class IntType>
struct logpdf_max<boost::random::uniform_smallint<IntType>> {
    double operator()(const boost::random::uniform_smallint<IntType>& d) const
    { return logpdf<boost::random::uniform_smallint<IntType>>()(d, d.min()); }
};
template <class IntType, class WeightType>
struct logpdf_max<boost::random::discrete_distribution<IntType, WeightType>> {
    double operator()(const boost::random::discrete_distribution<IntType, WeightType>& d) const
    {
        double m = -INFINITY;
        for (IntType x = d.min(); x <= d.max(); ++x) { const double l = static_cast<double>(logpdf<boost::random::discrete_distribution<IntType, WeightType>>()(d, x)); if (l > m) m = l; }
        return m;
    }
};
template <class RealType>
struct logpdf_max<boost::random::uniform_real_distribution<RealType>> {
    double operator()(const boost::random::uniform_real_distribution<RealType>& d) const
    { return static_cast<double>(logpdf<boost::random::uniform_real_distribution<RealType>>()(d, d.a())); }
};
template <class IntType, class RealType>
struct logpdf_max<boost::random::poisson_distribution<IntType, RealType>> {
    double operator()(const boost::random::poisson_distribution<IntType, RealType>& d) const
    {
        if (!(d.mean() > 0)) return __builtin_nan("");
        return static_cast<double>(logpdf<boost::random::poisson_distribution<IntType, RealType>>()(d, static_cast<IntType>(d.mean())));   // mode = floor(mean)
    }
};

}  // namespace cpprob
#endif
