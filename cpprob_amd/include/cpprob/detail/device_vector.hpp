// Device-capable stand-ins for the containers of vector-valued statements.
//
// A model with vector-valued statements (reference include/models/models.hpp:38-49, src/models/models.cpp:13-47) takes
// std::vector arguments, builds cpprob::multivariate_normal_distribution objects (std::vector members) and handles
// cpprob::NDArray values (std::vector members): none of that can exist on a GPU lane, which has no heap.  The model SOURCE can
// still run there unchanged: the model translation unit includes it a second time, inside namespace cpprob_device_view, with the
// three names mapped onto the fixed-capacity types below (cpprob/gpu.hpp: CPPROB_DEVICE_VIEW_BEGIN / _END), and registers that
// instantiation as the device code of the host function (CPPROB_REGISTER_MODEL_VIEW).  Same statements, same order, same
// arithmetic; storage in registers / scratch instead of the heap; capacity kVecCap elements per value (a longer vector is an error
// on the host, before anything is launched).
#ifndef CPPROB_COMPAT_DETAIL_DEVICE_VECTOR_HPP
#define CPPROB_COMPAT_DETAIL_DEVICE_VECTOR_HPP
#include <cstddef>
#include <initializer_list>
#include <stdexcept>
#include <type_traits>
#include <vector>

#include <boost/random/normal_distribution.hpp>

#include "cpprob/detail/hd.hpp"
#include "cpprob/distributions/utils_distributions.hpp"

namespace cpprob {
namespace device {

constexpr std::size_t kVecCap = 16;

// std::vector's interface as far as model bodies use it, over inline storage.  Trivially copyable: observes travel bytewise.
template <class T, std::size_t N = kVecCap>
class fixed_vector {
public:
    using value_type = T;
    using iterator = T*;
    using const_iterator = const T*;
    fixed_vector() = default;
    CPPROB_HD fixed_vector(std::initializer_list<T> il) { for (const T& x : il) push_back(x); }
    template <class Iter, class = std::enable_if_t<!std::is_arithmetic<Iter>::value>>
    CPPROB_HD fixed_vector(Iter first, Iter last) { for (; first != last; ++first) push_back(static_cast<T>(*first)); }
    CPPROB_HD fixed_vector(std::size_t n, const T& x) { for (std::size_t i = 0; i < n; ++i) push_back(x); }
    fixed_vector(const std::vector<T>& v)                                   // host: an observes tuple on its way to the device
    {
        if (v.size() > N) throw std::runtime_error("cpprob: a vector-valued observe has more components than the device path carries (16)");
        for (const T& x : v) push_back(x);
    }
    CPPROB_HD std::size_t size() const { return n_; }
    CPPROB_HD bool empty() const { return n_ == 0; }
    CPPROB_HD T* begin() { return data_; }
    CPPROB_HD T* end() { return data_ + n_; }
    CPPROB_HD const T* begin() const { return data_; }
    CPPROB_HD const T* end() const { return data_ + n_; }
    CPPROB_HD T& operator[](std::size_t i) { return data_[i]; }
    CPPROB_HD const T& operator[](std::size_t i) const { return data_[i]; }
    CPPROB_HD T& front() { return data_[0]; }
    CPPROB_HD const T& front() const { return data_[0]; }
    CPPROB_HD T& back() { return data_[n_ - 1]; }
    CPPROB_HD const T& back() const { return data_[n_ - 1]; }
    CPPROB_HD const T* data() const { return data_; }
    CPPROB_HD void push_back(const T& x) { if (n_ < N) data_[n_++] = x; }    // (capacity is checked on the host, where observes enter)
    CPPROB_HD void clear() { n_ = 0; }
private:
    T data_[N] = {};
    std::size_t n_ = 0;
};

}  // namespace device

// NDArray as model bodies see it on the device: a flat value with begin / end / size / [] and elementwise arithmetic.
template <class T = double>
class dev_NDArray {
public:
    using value_type = T;
    dev_NDArray() = default;
    CPPROB_HD dev_NDArray(T x) { v_.push_back(x); }
    CPPROB_HD dev_NDArray(const device::fixed_vector<T>& v) : v_(v) {}
    template <class Iter, class = std::enable_if_t<!std::is_arithmetic<Iter>::value>>
    CPPROB_HD dev_NDArray(Iter first, Iter last) : v_(first, last) {}
    CPPROB_HD std::size_t size() const { return v_.size(); }
    CPPROB_HD const T* begin() const { return v_.begin(); }
    CPPROB_HD const T* end() const { return v_.end(); }
    CPPROB_HD const T& operator[](std::size_t i) const { return v_[i]; }
    CPPROB_HD T& operator[](std::size_t i) { return v_[i]; }
    CPPROB_HD const device::fixed_vector<T>& values() const { return v_; }
    CPPROB_HD void push_back(T x) { v_.push_back(x); }
    CPPROB_HD dev_NDArray& operator+=(const dev_NDArray& o) { for (std::size_t i = 0; i < v_.size() && i < o.size(); ++i) v_[i] += o[i]; return *this; }
    CPPROB_HD dev_NDArray& operator-=(const dev_NDArray& o) { for (std::size_t i = 0; i < v_.size() && i < o.size(); ++i) v_[i] -= o[i]; return *this; }
    CPPROB_HD dev_NDArray& operator*=(T a) { for (std::size_t i = 0; i < v_.size(); ++i) v_[i] *= a; return *this; }
    CPPROB_HD friend dev_NDArray operator+(dev_NDArray a, const dev_NDArray& b) { return a += b; }
    CPPROB_HD friend dev_NDArray operator-(dev_NDArray a, const dev_NDArray& b) { return a -= b; }
    CPPROB_HD friend dev_NDArray operator*(dev_NDArray a, T b) { return a *= b; }
    CPPROB_HD friend dev_NDArray operator*(T b, dev_NDArray a) { return a *= b; }
private:
    device::fixed_vector<T> v_;
};

// cpprob::multivariate_normal_distribution (reference include/cpprob/distributions/multivariate_normal.hpp) over inline storage:
// independent normal components, the second constructor argument being the components' sigma (:41-50); generation in index order
// (:268-274).
template <class RealType = double>
class dev_multivariate_normal_distribution {
public:
    using input_type = device::fixed_vector<RealType>;
    using result_type = dev_NDArray<RealType>;
    dev_multivariate_normal_distribution() = default;
    template <class Iter, class = std::enable_if_t<!std::is_arithmetic<Iter>::value>>
    CPPROB_HD dev_multivariate_normal_distribution(Iter mean_first, Iter mean_last, RealType sigma) : mean_(mean_first, mean_last), sigma_(mean_.size(), sigma) {}
    template <class IterMean, class IterSigma, class = std::enable_if_t<!std::is_arithmetic<IterSigma>::value>>
    CPPROB_HD dev_multivariate_normal_distribution(IterMean mean_first, IterMean mean_last, IterSigma sigma_first, IterSigma sigma_last)
        : mean_(mean_first, mean_last), sigma_(sigma_first, sigma_last) {}
    CPPROB_HD dev_multivariate_normal_distribution(const std::initializer_list<RealType>& mean, RealType sigma) : mean_(mean), sigma_(mean.size(), sigma) {}
    CPPROB_HD dev_multivariate_normal_distribution(const std::initializer_list<RealType>& mean, const std::initializer_list<RealType>& sigma) : mean_(mean), sigma_(sigma) {}
    CPPROB_HD dev_multivariate_normal_distribution(const dev_NDArray<RealType>& mean, RealType sigma) : mean_(mean.values()), sigma_(mean.size(), sigma) {}
    CPPROB_HD dev_multivariate_normal_distribution(const dev_NDArray<RealType>& mean, const dev_NDArray<RealType>& sigma) : mean_(mean.values()), sigma_(sigma.values()) {}
    CPPROB_HD std::size_t size() const { return mean_.size(); }
    CPPROB_HD RealType mean_at(std::size_t i) const { return mean_[i]; }
    CPPROB_HD RealType sigma_at(std::size_t i) const { return i < sigma_.size() ? sigma_[i] : RealType(1); }
    CPPROB_HD dev_NDArray<RealType> mean() const { return dev_NDArray<RealType>(mean_); }
    void reset() {}
    template <class URNG>
    result_type operator()(URNG& rng)                              // host: the structural pass of this translation unit
    {
        result_type x;
        for (std::size_t i = 0; i < mean_.size(); ++i) { boost::random::normal_distribution<RealType> d(mean_[i], sigma_at(i)); x.push_back(d(rng)); }
        return x;
    }
private:
    device::fixed_vector<RealType> mean_, sigma_;
};

// logpdf = sum of the components' normal logpdfs (reference utils_multivariate_normal.hpp:20-33)
template <class RealType>
struct logpdf<dev_multivariate_normal_distribution<RealType>> {
    CPPROB_HD RealType operator()(const dev_multivariate_normal_distribution<RealType>& distr, const dev_NDArray<RealType>& x) const
    {
        RealType ret = 0;
        for (std::size_t i = 0; i < distr.size() && i < x.size(); ++i)
            ret += static_cast<RealType>(cph::normal_logpdf(static_cast<double>(x[i]), static_cast<double>(distr.mean_at(i)), static_cast<double>(distr.sigma_at(i))));
        return ret;
    }
};

template <class T> struct is_dev_ndarray : std::false_type {};
template <class T> struct is_dev_ndarray<dev_NDArray<T>> : std::true_type {};
template <class T> struct is_dev_mvn : std::false_type {};
template <class T> struct is_dev_mvn<dev_multivariate_normal_distribution<T>> : std::true_type {};

}  // namespace cpprob

// the name the device view's `std::vector` is mapped onto (cpprob/gpu.hpp)
namespace std {
template <class T> using cpprob_device_view_vector = ::cpprob::device::fixed_vector<T>;
}
#endif
