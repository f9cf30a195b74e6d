// cpprob::NDArray -- the value type of vector-valued statements (reference include/cpprob/ndarray.hpp): what
// a multivariate sample returns, what a vector observe takes, what predict() files under the real list
// (state.hpp:330-337) and what EmpiricalDistribution averages elementwise (stats_printer.hpp:113).
// Host-only (it owns std::vectors): on the device the components of a vector-valued statement are rows of
// the particle store (csrc/models.hpp, ModelGaussianND), never an object.
//
// Text form, as the reference prints and parses it (ndarray.hpp:273-288): a one-element array prints as the
// bare number, anything else as `[v0 v1 ...]`, followed by ` s[d0 d1 ...]` when it has more than one axis.
#ifndef CPPROB_COMPAT_NDARRAY_HPP
#define CPPROB_COMPAT_NDARRAY_HPP
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <functional>
#include <istream>
#include <iterator>
#include <numeric>
#include <ostream>
#include <stdexcept>
#include <type_traits>
#include <utility>
#include <vector>

namespace cpprob {

template <class T = double>
class NDArray {
public:
    using value_type = T;

    NDArray() = default;
    NDArray(T x) : values_{x}, shape_{1} {}
    NDArray(std::vector<T> values) : values_(std::move(values)), shape_{values_.size()} {}
    NDArray(std::vector<T> values, std::vector<std::size_t> shape) : values_(std::move(values)), shape_(std::move(shape)) {}
    // From a range of numbers (any arithmetic type) or of nested containers: ragged nesting is padded with zeros to the
    // smallest enclosing box, row-major (the behaviour pinned by the reference's tests/cpprob/ndarray.cpp).
    template <class Iter> NDArray(Iter first, Iter last)
    {
        using Elem = typename std::iterator_traits<Iter>::value_type;
        build(first, last, std::is_arithmetic<Elem>{});
    }
    template <class U> NDArray(const NDArray<U>& o) : values_(o.begin(), o.end()), shape_(o.shape()) {}

    const std::vector<T>& values() const { return values_; }
    const std::vector<std::size_t>& shape() const { return shape_; }
    std::size_t size() const { return values_.size(); }
    typename std::vector<T>::const_iterator begin() const { return values_.begin(); }
    typename std::vector<T>::const_iterator end() const { return values_.end(); }
    const T& operator[](std::size_t i) const { return values_[i]; }
    bool is_scalar() const { return shape_.size() == 1 && shape_[0] == 1; }
    explicit operator T() const
    {
        if (values_.size() != 1) throw std::runtime_error("NDArray: only a one-element array converts to a number");
        return values_[0];
    }

    // elementwise arithmetic; an operand of one element acts as a number (an empty array is the additive start value
    // of std::accumulate in the estimators)
    NDArray& operator+=(const NDArray& o) { return combine(o, std::plus<T>()); }
    NDArray& operator-=(const NDArray& o) { return combine(o, std::minus<T>()); }
    NDArray& operator*=(const NDArray& o) { return combine(o, std::multiplies<T>()); }
    NDArray& operator/=(const NDArray& o) { return combine(o, std::divides<T>()); }
    NDArray& operator*=(T a) { for (auto& v : values_) v *= a; return *this; }
    NDArray& operator/=(T a) { for (auto& v : values_) v /= a; return *this; }
    friend NDArray operator+(NDArray a, const NDArray& b) { return a += b; }
    friend NDArray operator-(NDArray a, const NDArray& b) { return a -= b; }
    friend NDArray operator*(NDArray a, const NDArray& b) { return a *= b; }
    friend NDArray operator/(NDArray a, const NDArray& b) { return a /= b; }
    friend NDArray operator*(NDArray a, T b) { return a *= b; }
    friend NDArray operator*(T b, NDArray a) { return a *= b; }
    friend NDArray operator/(NDArray a, T b) { return a /= b; }
    friend bool operator==(const NDArray& a, const NDArray& b) { return a.values_ == b.values_ && a.shape_ == b.shape_; }
    friend bool operator!=(const NDArray& a, const NDArray& b) { return !(a == b); }
    friend bool operator<(const NDArray& a, const NDArray& b) { return a.values_ < b.values_; }   // map key in distribution()

    template <class CharT, class Traits>
    friend std::basic_ostream<CharT, Traits>& operator<<(std::basic_ostream<CharT, Traits>& os, const NDArray& v)
    {
        if (v.is_scalar()) return os << v.values_[0];
        os << os.widen('[');
        for (std::size_t i = 0; i < v.values_.size(); ++i) { if (i) os << os.widen(' '); os << v.values_[i]; }
        os << os.widen(']');
        if (v.shape_.size() > 1) {
            os << os.widen(' ') << os.widen('s') << os.widen('[');
            for (std::size_t i = 0; i < v.shape_.size(); ++i) { if (i) os << os.widen(' '); os << v.shape_[i]; }
            os << os.widen(']');
        }
        return os;
    }

    template <class CharT, class Traits>
    friend std::basic_istream<CharT, Traits>& operator>>(std::basic_istream<CharT, Traits>& is, NDArray& v)
    {
        CharT ch;
        if (!(is >> std::ws)) return is;
        if (is.peek() != Traits::to_int_type(is.widen('['))) {        // bare number
            T x;
            if (is >> x) v = NDArray(x);
            return is;
        }
        is >> ch;
        std::vector<T> vals;
        if (!read_list(is, vals)) return is;
        std::vector<std::size_t> shape{vals.size()};
        // optional ` s[...]`
        const auto pos = is.tellg();
        CharT s = 0, b = 0;
        if ((is >> std::ws >> s) && s == is.widen('s') && (is >> b) && b == is.widen('[')) {
            std::vector<std::size_t> sh;
            if (!read_list(is, sh)) return is;
            shape = sh;
        } else {
            is.clear();
            is.seekg(pos);
        }
        v = NDArray(std::move(vals), std::move(shape));
        return is;
    }

private:
    std::vector<T> values_;
    std::vector<std::size_t> shape_;

    template <class Iter> void build(Iter first, Iter last, std::true_type /*numbers*/)
    {
        for (Iter it = first; it != last; ++it) values_.push_back(static_cast<T>(*it));
        shape_.assign(1, values_.size());
    }
    template <class Iter> void build(Iter first, Iter last, std::false_type /*containers*/)
    {
        shape_.assign(1, static_cast<std::size_t>(std::distance(first, last)));
        for (Iter it = first; it != last; ++it) extent(*it, 1);
        std::size_t total = 1;
        for (auto d : shape_) total *= d;
        values_.assign(total, T());
        std::size_t i = 0;
        for (Iter it = first; it != last; ++it, ++i) place(*it, 1, i * stride(0));
    }
    std::size_t stride(std::size_t axis) const
    {
        std::size_t st = 1;
        for (std::size_t d = axis + 1; d < shape_.size(); ++d) st *= shape_[d];
        return st;
    }
    // largest extent found at every nesting depth
    template <class C> std::enable_if_t<!std::is_arithmetic<C>::value> extent(const C& c, std::size_t depth)
    {
        if (shape_.size() <= depth) shape_.resize(depth + 1, 0);
        if (c.size() > shape_[depth]) shape_[depth] = c.size();
        for (const auto& e : c) extent(e, depth + 1);
    }
    template <class C> std::enable_if_t<std::is_arithmetic<C>::value> extent(const C&, std::size_t) {}
    template <class C> std::enable_if_t<!std::is_arithmetic<C>::value> place(const C& c, std::size_t depth, std::size_t offset)
    {
        std::size_t i = 0;
        for (const auto& e : c) { place(e, depth + 1, offset + i * stride(depth)); ++i; }
    }
    template <class C> std::enable_if_t<std::is_arithmetic<C>::value> place(const C& x, std::size_t, std::size_t offset) { values_[offset] = static_cast<T>(x); }

    template <class Op>
    NDArray& combine(const NDArray& o, Op op)
    {
        if (o.values_.empty()) return *this;
        if (values_.empty()) { values_.assign(o.values_.size(), T()); shape_ = o.shape_; }      // an empty array is a zero of any shape
        if (o.values_.size() == 1) { for (auto& v : values_) v = op(v, o.values_[0]); return *this; }
        if (values_.size() == 1) { const T a = values_[0]; *this = o; for (auto& v : values_) v = op(a, v); return *this; }
        if (o.values_.size() != values_.size()) throw std::runtime_error("NDArray: shapes do not match");
        for (std::size_t i = 0; i < values_.size(); ++i) values_[i] = op(values_[i], o.values_[i]);
        return *this;
    }

    // after the opening bracket: elements separated by blanks up to the closing bracket
    template <class CharT, class Traits, class U>
    static bool read_list(std::basic_istream<CharT, Traits>& is, std::vector<U>& out)
    {
        for (;;) {
            if (!(is >> std::ws)) return false;
            if (is.peek() == Traits::to_int_type(is.widen(']'))) { CharT c; is >> c; return true; }
            U x;
            if (!(is >> x)) return false;
            out.push_back(x);
        }
    }
};

template <class T> NDArray<T> sqrt(NDArray<T> a)
{
    std::vector<T> v(a.values());
    for (auto& x : v) x = std::sqrt(x);
    return NDArray<T>(std::move(v), a.shape());
}

}  // namespace cpprob
#endif
