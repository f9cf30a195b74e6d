// Clean-room stand-in for boost/math/distributions/normal.hpp: the density object and pdf/cdf free
// functions that include/models/models.hpp names (normal_rejection_sampling, :82-112).
#ifndef CPPROB_COMPAT_BOOST_MATH_DISTRIBUTIONS_NORMAL_HPP
#define CPPROB_COMPAT_BOOST_MATH_DISTRIBUTIONS_NORMAL_HPP
#include <cmath>
#include "cpprob/detail/hd.hpp"
namespace boost { namespace math {
template <class RealType = double>
class normal_distribution {
public:
    using value_type = RealType;
    CPPROB_HD explicit normal_distribution(RealType mean = 0, RealType sd = 1) : mean_(mean), sd_(sd) {}
    CPPROB_HD RealType mean() const { return mean_; }
    CPPROB_HD RealType standard_deviation() const { return sd_; }
    CPPROB_HD RealType location() const { return mean_; }
    CPPROB_HD RealType scale() const { return sd_; }
private:
    RealType mean_, sd_;
};
using normal = normal_distribution<double>;
template <class RealType>
CPPROB_HD inline RealType pdf(const normal_distribution<RealType>& d, const RealType& x)
{
    const RealType z = (x - d.mean()) / d.standard_deviation();
    return std::exp(-z * z / 2) / (d.standard_deviation() * static_cast<RealType>(2.506628274631000502415765284811045253L));
}
template <class RealType>
CPPROB_HD inline RealType cdf(const normal_distribution<RealType>& d, const RealType& x)
{
    return static_cast<RealType>(0.5) * std::erfc(-(x - d.mean()) / (d.standard_deviation() * static_cast<RealType>(1.414213562373095048801688724209698079L)));
}
}}
#endif
