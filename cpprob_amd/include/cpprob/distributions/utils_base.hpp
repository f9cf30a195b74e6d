// Trait declarations -- reference include/cpprob/distributions/utils_base.hpp:6-28.  Only logpdf is on
// the sis/smc path; buffer/proposal/flatbuffers traits belong to the NN protocol (out of scope).
#ifndef CPPROB_COMPAT_UTILS_BASE_HPP
#define CPPROB_COMPAT_UTILS_BASE_HPP
namespace cpprob {
template <class Distribution> struct logpdf;
template <class Distribution> struct proposal;
template <class Distribution> struct buffer;
template <class Distribution> struct normalise;
// Not in the reference: the largest value logpdf<Distribution>()(distr, x) takes over x (the density at the mode) -- what the device
// engine's fixed-point weights are taken against (cpprob/gpu.hpp).  NaN: unknown for this distribution.
template <class Distribution> struct logpdf_max {
    double operator()(const Distribution&) const { return __builtin_nan(""); }
};
}
#endif
