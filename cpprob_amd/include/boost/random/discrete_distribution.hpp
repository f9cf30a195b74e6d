// Clean-room stand-in for boost/random/discrete_distribution.hpp: interface only.  Boost keeps the
// weights in a std::vector and samples through an alias table; device code has no heap, so this
// stand-in holds up to kMaxWeights normalised probabilities inline.  probabilities() returns a
// fixed-capacity range with operator[], begin(), end(), size().
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
    CPPROB_HD discrete_distribution() { probs_.n = 1; probs_.p[0] = 1; raw_ = probs_; }
    template <class Iter>
    CPPROB_HD discrete_distribution(Iter first, Iter last) { init(first, last); }
    discrete_distribution(std::initializer_list<WeightType> w) { init(w.begin(), w.end()); }
    CPPROB_HD result_type min() const { return 0; }
    CPPROB_HD result_type max() const { return static_cast<result_type>(probs_.n - 1); }
    CPPROB_HD probabilities_type probabilities() const { return probs_; }
    // the weights as given (the engine's inverse-CDF draw works on these, in the caller's order)
    CPPROB_HD const probabilities_type& weights() const { return raw_; }
    CPPROB_HD void reset() {}
    template <class URNG>
    result_type operator()(URNG& g)
    {
        const double u = static_cast<double>(g() - URNG::min()) / (static_cast<double>(URNG::max() - URNG::min()) + 1.0);
        double acc = 0;
        for (std::size_t i = 0; i + 1 < probs_.n; ++i) { acc += probs_.p[i]; if (u < acc) return static_cast<result_type>(i); }
        return static_cast<result_type>(probs_.n - 1);
    }
private:
    template <class Iter>
    CPPROB_HD void init(Iter first, Iter last)
    {
        std::size_t n = 0;
        WeightType tot = 0;
        for (Iter it = first; it != last && n < kMaxWeights; ++it) { probs_.p[n] = *it; raw_.p[n] = *it; ++n; }
        for (std::size_t i = 0; i < n; ++i) tot += probs_.p[i];        // same order as the engine's tot
        if (n == 0) { n = 1; probs_.p[0] = 1; raw_.p[0] = 1; tot = 1; }
        for (std::size_t i = 0; i < n; ++i) probs_.p[i] = probs_.p[i] / tot;
        probs_.n = n; raw_.n = n;
    }
    probabilities_type probs_, raw_;
};

}  // namespace random
using random::discrete_distribution;
}  // namespace boost
#endif
