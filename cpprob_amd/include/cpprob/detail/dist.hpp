// logpdf arithmetic (host and device) -- the weight side of cpprob::observe
// (reference include/cpprob/cpprob.hpp:79-90 -> StateInfer::increment_log_prob,
// src/cpprob/state.cpp:212-223).  One function per reference functor, same branch order.
#pragma once
#include <math.h>
#include <stdint.h>

#include "cpprob/detail/hd.hpp"

namespace cph {

constexpr double kPi = 3.14159265358979323846;

// logpdf<boost::random::normal_distribution<R>>
// reference include/cpprob/distributions/utils_normal_distribution.hpp:20-45
CPPROB_HD inline double normal_logpdf(double x, double mean, double sigma)
{
    if (sigma == 0) return x == mean ? 0.0 : -INFINITY;     // :28-32 Dirac delta
    if (fabs(x) == INFINITY) return -INFINITY;              // :34-36
    double r = (x - mean) / sigma;                           // :38
    r *= r;                                                  // :39
    r += log(2 * kPi * sigma * sigma);                       // :40
    r *= -0.5;                                               // :41
    return r;
}

// Same value with the sigma-only term hoisted: log_norm = log(2*pi*sigma^2) computed once
// per model ("log term hoistable").  Bitwise equal to normal_logpdf
// when the same log_norm is passed, for finite x and sigma > 0.
CPPROB_HD inline double normal_logpdf_hoisted(double x, double mean, double sigma, double log_norm)
{
    if (fabs(x) == INFINITY) return -INFINITY;
    double r = (x - mean) / sigma;
    r *= r;
    r += log_norm;
    r *= -0.5;
    return r;
}

// The fused Gaussian kernels' form: the division by sigma as a multiplication by the host-computed 1 / sigma (an fp64 division is
// ~20 instructions on this GPU and the SIS kernel is VALU-bound).  Differs from normal_logpdf by at most 1 ulp of (x - mean) / sigma;
// the functor API (cpprob::logpdf, the building block, the generic model path) keeps the reference's division.
CPPROB_HD inline double normal_logpdf_scaled(double x, double mean, double inv_sigma, double log_norm)
{
    if (fabs(x) == INFINITY) return -INFINITY;
    double r = (x - mean) * inv_sigma;
    r *= r;
    r += log_norm;
    r *= -0.5;
    return r;
}

// logpdf<boost::random::uniform_smallint<I>>  utils_uniform_smallint.hpp:17-27
CPPROB_HD inline double uniform_smallint_logpdf(int64_t x, int64_t a, int64_t b)
{
    if (x < a || x > b) return -INFINITY;
    return -log((double)(b - a) + 1.0);
}

// logpdf<boost::random::discrete_distribution<I, W>>  utils_discrete.hpp:17-27
CPPROB_HD inline double discrete_logpdf(int64_t x, const double* w, int k)
{
    if (x < 0 || x > k - 1) return -INFINITY;
    double tot = 0.0;
    for (int i = 0; i < k; ++i) tot += w[i];
    return log(w[x] / tot);
}

// logpdf<boost::random::uniform_real_distribution<R>>  utils_uniform_real.hpp:21-31
CPPROB_HD inline double uniform_real_logpdf(double x, double a, double b)
{
    if (x < a || x > b) return -INFINITY;
    return -log(b - a);
}

// logpdf<boost::random::poisson_distribution<I, R>>  utils_poisson.hpp:17-36
CPPROB_HD inline double poisson_logpdf(int64_t x, double l)
{
    if (l == 0.0) return -INFINITY;
    double ret = (double)x * log(l) - l;
    for (int64_t i = 1; i <= x; ++i) ret -= log((double)i);
    return ret;
}

}  // namespace cph
