// Clean-room stand-in for boost/random/discrete_distribution.hpp: interface only.  Boost keeps the
// weights in a std::vector and samples through an alias table; device code has no heap, so this
// stand-in holds up to kMaxWeights weights inline, as given; probabilities() normalises them when it
// is asked (a fixed-capacity range with operator[], begin(), end(), size()).  Construction is a copy and
// nothing else: on the device a distribution built in a replayed, dead iteration of the model's loop
// (cpprob/detail/device_trace.hpp) then costs the issue of its loads, not the wait for them and a
// division per weight.
#ifndef CPPROB_COMPAT_BOOST_RANDOM_DISCRETE_DISTRIBUTION_HPP
#define CPPROB_COMPAT_BOOST_RANDOM_DISCRETE_DISTRIBUTION_HPP
#include <cstddef>
#include <initializer_list>
#include "cpprob/detail/hd.hpp"

namespace boost { namespace random {

template <class IntType = int, class WeightType = double>
class discrete_distribution {
public:
    static constexpr std::size_t kMaxWeights = 16;
    using input_type = WeightType;
    using result_type = IntType;
    struct probabilities_type {
        WeightType p[kMaxWeights];
        std::size_t n;
        CPPROB_HD const WeightType& operator[](std::size_t i) const { return p[i]; }
        CPPROB_HD const WeightType* begin() const { return p; }
        CPPROB_HD const WeightType* end() const { return p + n; }
        CPPROB_HD std::size_t size() const { return n; }
    };
    CPPROB_HD discrete_distribution() { raw_.n = 1; raw_.p[0] = 1; }
    template <class Iter>
    CPPROB_HD discrete_distribution(Iter first, Iter last) { init(first, last); }
    discrete_distribution(std::initializer_list<WeightType> w) { init(w.begin(), w.end()); }
    CPPROB_HD result_type min() const { return 0; }
    CPPROB_HD result_type max() const { return static_cast<result_type>(raw_.n - 1); }
    CPPROB_HD probabilities_type probabilities() const
    {
        probabilities_type pr;
        WeightType tot = 0;
        for (std::size_t i = 0; i < raw_.n; ++i) tot += raw_.p[i];        // same order as the engine's tot
        for (std::size_t i = 0; i < raw_.n; ++i) pr.p[i] = raw_.p[i] / tot;
        pr.n = raw_.n;
        return pr;
    }
    // the weights as given (the engine's inverse-CDF draw works on these, in the caller's order)
    CPPROB_HD const probabilities_type& weights() const { return raw_; }
    CPPROB_HD void reset() {}
    template <class URNG>
    result_type operator()(URNG& g)
    {
        const double u = static_cast<double>(g() - URNG::min()) / (static_cast<double>(URNG::max() - URNG::min()) + 1.0);
        const probabilities_type pr = probabilities();
        double acc = 0;
        for (std::size_t i = 0; i + 1 < pr.n; ++i) { acc += pr.p[i]; if (u < acc) return static_cast<result_type>(i); }
        return static_cast<result_type>(pr.n - 1);
    }
private:
    template <class Iter>
    CPPROB_HD void init(Iter first, Iter last)
    {
        std::size_t n = 0;
        for (Iter it = first; it != last && n < kMaxWeights; ++it) { raw_.p[n] = *it; ++n; }
        if (n == 0) { n = 1; raw_.p[0] = 1; }
        raw_.n = n;
    }
    probabilities_type raw_;
};

}  // namespace random
using random::discrete_distribution;
}  // namespace boost
#endif
