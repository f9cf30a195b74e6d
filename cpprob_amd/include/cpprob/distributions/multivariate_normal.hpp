// cpprob::multivariate_normal_distribution -- independent normal components over an NDArray shape
// (reference include/cpprob/distributions/multivariate_normal.hpp: same constructors, accessors and
// generation order; the second argument is used as the components' sigma, :41-50).
// Host-only object (the structural dry run and user code); on the device its components are rows of the
// particle store (csrc/models.hpp, ModelGaussianND).
#ifndef CPPROB_COMPAT_MULTIVARIATE_NORMAL_HPP
#define CPPROB_COMPAT_MULTIVARIATE_NORMAL_HPP
#include <cstddef>
#include <initializer_list>
#include <vector>

#include <boost/random/normal_distribution.hpp>

#include "cpprob/ndarray.hpp"

namespace cpprob {

template <class RealType = double>
class multivariate_normal_distribution {
public:
    using input_type = std::vector<RealType>;
    using result_type = NDArray<RealType>;

    class param_type {
    public:
        using distribution_type = multivariate_normal_distribution;
        param_type() : mean_{RealType(0)}, sigma_{RealType(1)}, shape_{1} {}
        template <class Iter>
        param_type(Iter mean_first, Iter mean_last, RealType sigma) : mean_(mean_first, mean_last), sigma_(mean_.size(), sigma), shape_{mean_.size()} {}
        template <class IterMean, class IterSigma>
        param_type(IterMean mean_first, IterMean mean_last, IterSigma sigma_first, IterSigma sigma_last)
            : mean_(mean_first, mean_last), sigma_(sigma_first, sigma_last), shape_{mean_.size()} {}
        param_type(const NDArray<RealType>& mean, RealType sigma) : mean_(mean.values()), sigma_(mean_.size(), sigma), shape_(mean.shape()) {}
        param_type(const NDArray<RealType>& mean, const NDArray<RealType>& sigma) : mean_(mean.values()), sigma_(sigma.values()), shape_(mean.shape()) {}

        NDArray<RealType> mean() const { return NDArray<RealType>(mean_, shape_); }
        std::vector<RealType> covariance() const
        {
            std::vector<RealType> c(sigma_);
            for (auto& s : c) s = s * s;
            return c;
        }
        std::vector<std::size_t> shape() const { return shape_; }
        std::vector<boost::random::normal_distribution<RealType>> distr() const
        {
            std::vector<boost::random::normal_distribution<RealType>> d;
            for (std::size_t i = 0; i < mean_.size(); ++i) d.emplace_back(mean_[i], i < sigma_.size() ? sigma_[i] : RealType(1));
            return d;
        }
        friend bool operator==(const param_type& a, const param_type& b) { return a.mean_ == b.mean_ && a.sigma_ == b.sigma_ && a.shape_ == b.shape_; }
        friend bool operator!=(const param_type& a, const param_type& b) { return !(a == b); }
    private:
        std::vector<RealType> mean_, sigma_;
        std::vector<std::size_t> shape_;
    };

    multivariate_normal_distribution() = default;
    explicit multivariate_normal_distribution(const param_type& p) : param_(p) {}
    template <class Iter>
    multivariate_normal_distribution(Iter mean_first, Iter mean_last, RealType sigma) : param_(mean_first, mean_last, sigma) {}
    template <class IterMean, class IterSigma>
    multivariate_normal_distribution(IterMean mean_first, IterMean mean_last, IterSigma sigma_first, IterSigma sigma_last)
        : param_(mean_first, mean_last, sigma_first, sigma_last) {}
    multivariate_normal_distribution(const std::initializer_list<RealType>& mean, RealType sigma) : param_(mean.begin(), mean.end(), sigma) {}
    multivariate_normal_distribution(const std::initializer_list<RealType>& mean, const std::initializer_list<RealType>& sigma)
        : param_(mean.begin(), mean.end(), sigma.begin(), sigma.end()) {}
    multivariate_normal_distribution(const NDArray<RealType>& mean, RealType sigma) : param_(mean, sigma) {}
    multivariate_normal_distribution(const NDArray<RealType>& mean, const NDArray<RealType>& sigma) : param_(mean, sigma) {}

    NDArray<RealType> mean() const { return param_.mean(); }
    std::vector<RealType> covariance() const { return param_.covariance(); }
    std::vector<std::size_t> shape() const { return param_.shape(); }
    std::vector<boost::random::normal_distribution<RealType>> distr() const { return param_.distr(); }
    param_type param() const { return param_; }
    void param(const param_type& p) { param_ = p; }
    void reset() {}

    // the components draw from the generator one after the other, in index order (multivariate_normal.hpp:268-274)
    template <class URNG>
    result_type operator()(URNG& rng)
    {
        std::vector<RealType> x;
        for (auto& d : param_.distr()) x.push_back(d(rng));
        return NDArray<RealType>(std::move(x), param_.shape());
    }
    template <class URNG> result_type operator()(URNG& rng, const param_type& p) { return multivariate_normal_distribution(p)(rng); }

private:
    param_type param_;
};

}  // namespace cpprob
#endif
