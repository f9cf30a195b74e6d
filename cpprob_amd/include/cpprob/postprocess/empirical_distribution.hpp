// EmpiricalDistribution -- the estimators of reference include/cpprob/postprocess/empirical_distribution.hpp:
// max-shifted logsumexp (:125-143), mean (:68-71), variance = raw2 - mean^2 (:78-81) -- moments are NDArrays,
// elementwise for vector-valued predicts, a bare number otherwise -- categorical distribution (:30-40),
// MAP (:42-50), num_points.  Host code: this is the
// post-processing of files that were already written, as in the reference; the in-memory results of
// a device run come from the engine directly (cpprob::gpu::Result).
#ifndef CPPROB_COMPAT_EMPIRICAL_DISTRIBUTION_HPP
#define CPPROB_COMPAT_EMPIRICAL_DISTRIBUTION_HPP

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <map>
#include <utility>
#include <vector>

#include "cpprob/ndarray.hpp"

namespace cpprob {

template <class T, class WeightType = double>
class EmpiricalDistribution {
public:
    void add_point(const T& value, const WeightType logw) { x_logw_.emplace_back(value, logw); }
    std::size_t num_points() const { return x_logw_.size(); }

    std::map<T, WeightType> distribution() const
    {
        std::map<T, WeightType> ret;
        const WeightType log_norm = log_normalisation_constant();
        for (const auto& p : x_logw_) ret[p.first] += std::exp(p.second - log_norm);
        return ret;
    }
    T max_a_posteriori(const std::map<T, WeightType>& distr) const
    {
        return std::max_element(distr.begin(), distr.end(), [](const auto& a, const auto& b) { return a.second < b.second; })->first;
    }
    T max_a_posteriori() const { return max_a_posteriori(distribution()); }

    NDArray<WeightType> raw_moment(const int n) const
    {
        if (x_logw_.empty()) return NDArray<WeightType>();
        const WeightType log_norm = log_normalisation_constant();
        NDArray<WeightType> ret;                                     // empty = zero of the points' shape
        for (const auto& e : x_logw_) {
            const NDArray<WeightType> x(e.first);
            NDArray<WeightType> p = x;
            for (int k = 1; k < n; ++k) p *= x;
            ret += p * std::exp(e.second - log_norm);
        }
        return ret;
    }
    NDArray<WeightType> mean() const { return raw_moment(1); }
    NDArray<WeightType> variance(const NDArray<WeightType>& mean) const { return raw_moment(2) - mean * mean; }
    NDArray<WeightType> variance() const { return variance(mean()); }
    NDArray<WeightType> std() const { return sqrt(variance()); }

    WeightType log_normalisation_constant() const
    {
        if (x_logw_.empty()) return WeightType();
        WeightType max = x_logw_.front().second;
        for (const auto& e : x_logw_) max = std::max(max, e.second);
        WeightType acc = 0;
        for (const auto& e : x_logw_) acc += std::exp(e.second - max);
        return std::log(acc) + max;
    }

private:
    std::vector<std::pair<T, WeightType>> x_logw_;
};

}  // namespace cpprob
#endif
