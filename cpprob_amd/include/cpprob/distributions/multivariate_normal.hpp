// Minimal cpprob::multivariate_normal_distribution (diagonal) so that model headers naming it parse.
// Host-only; vector-valued statements are outside the sis/smc scope table:
// sampling/observing it under the device engine reports "unsupported" at run time.
#ifndef CPPROB_COMPAT_MULTIVARIATE_NORMAL_HPP
#define CPPROB_COMPAT_MULTIVARIATE_NORMAL_HPP
#include <initializer_list>
#include <vector>
#include "cpprob/ndarray.hpp"
namespace cpprob {
template <class RealType = double>
class multivariate_normal_distribution {
public:
    using result_type = NDArray<RealType>;
    multivariate_normal_distribution() = default;
    multivariate_normal_distribution(std::initializer_list<RealType> mean, std::initializer_list<RealType> sigma) : mean_(mean), sigma_(sigma) {}
    template <class Iter>
    multivariate_normal_distribution(Iter first, Iter last, RealType sigma) : mean_(first, last), sigma_(mean_.size(), sigma) {}
    const std::vector<RealType>& mean() const { return mean_; }
    const std::vector<RealType>& sigma() const { return sigma_; }
private:
    std::vector<RealType> mean_, sigma_;
};
}
#endif
