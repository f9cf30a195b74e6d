// Minimal cpprob::NDArray so that model headers which name it keep compiling.  The reference class
// (include/cpprob/ndarray.hpp) serves vector-valued predicts/observes, which are outside the sis/smc
// scope table; scalar models never instantiate it.
#ifndef CPPROB_COMPAT_NDARRAY_HPP
#define CPPROB_COMPAT_NDARRAY_HPP
#include <cstddef>
#include <vector>
namespace cpprob {
template <class T = double>
class NDArray {
public:
    NDArray() = default;
    NDArray(T x) : values_{x}, shape_{} {}
    template <class Iter> NDArray(Iter first, Iter last) : values_(first, last), shape_{values_.size()} {}
    const std::vector<T>& values() const { return values_; }
    const std::vector<std::size_t>& shape() const { return shape_; }
    typename std::vector<T>::const_iterator begin() const { return values_.begin(); }
    typename std::vector<T>::const_iterator end() const { return values_.end(); }
    bool is_scalar() const { return shape_.empty() && values_.size() == 1; }
private:
    std::vector<T> values_;
    std::vector<std::size_t> shape_;
};
}
#endif
