// The models of the sis/smc scope table written against the CPProb statement API, as a user would
// write them.  They restate, statement for statement, reference include/models/models.hpp:22-35
// (gaussian_unknown_mean), src/models/gaussian.cpp:6-17 (README variant), models.hpp:67-80
// (linear_gaussian_1d), models.hpp:114-141 (hmm); tests compile the reference's own header instead
// when /root/reference is available.  Nothing here is device-specific.
#ifndef CPPROB_EXAMPLES_TARGET_MODELS_HPP
#define CPPROB_EXAMPLES_TARGET_MODELS_HPP
#include <array>
#include <cmath>
#include <cstddef>
#include <vector>

#include <boost/math/distributions/normal.hpp>
#include <boost/random/discrete_distribution.hpp>
#include <boost/random/normal_distribution.hpp>
#include <boost/random/poisson_distribution.hpp>
#include <boost/random/uniform_real_distribution.hpp>
#include <boost/random/uniform_smallint.hpp>

#include "cpprob/cpprob.hpp"

namespace models {

// prior N(1, sd sqrt 5), two observes with sd sqrt 2, predict "Mu"
template <class Real = double>
void gaussian_unknown_mean(const Real y1, const Real y2)
{
    boost::random::normal_distribution<Real> prior{1, std::sqrt(5)};
    const Real mu = cpprob::sample(prior, true);
    boost::random::normal_distribution<Real> lik{mu, static_cast<Real>(std::sqrt(2))};
    cpprob::observe(lik, y1);
    cpprob::observe(lik, y2);
    cpprob::predict(mu, "Mu");
}

// the same model as a functor (restates reference include/models/models.hpp:51-65): found by TYPE, not by address
template <class Real = double>
struct GaussFunctor {
    void operator()(const Real y1, const Real y2) const
    {
        boost::random::normal_distribution<Real> prior{1, std::sqrt(5)};
        const Real mu = cpprob::sample(prior, true);
        boost::random::normal_distribution<Real> lik{mu, static_cast<Real>(std::sqrt(2))};
        cpprob::observe(lik, y1);
        cpprob::observe(lik, y2);
        cpprob::predict(mu, "Mu");
    }
};

// README variant: prior N(1, 1.5), likelihood sd 2, predict "Mean"
template <class Real = double>
void gaussian_readme(const Real x1, const Real x2)
{
    boost::normal_distribution<Real> prior{1, 1.5};
    const Real mu = cpprob::sample(prior, true);
    boost::normal_distribution<Real> lik{mu, 2};
    cpprob::observe(lik, x1);
    cpprob::observe(lik, x2);
    cpprob::predict(mu, "Mean");
}

// random walk observed with unit noise; predict "State" after every observe
template <std::size_t N>
void linear_gaussian_1d(const std::array<double, N>& ys)
{
    double x = 0;
    for (const auto y : ys) {
        boost::random::normal_distribution<> step{x, 1};
        x = cpprob::sample(step, true);
        boost::random::normal_distribution<> lik{x, 1};
        cpprob::observe(lik, y);
        cpprob::predict(x, "State");
    }
}

// 3-state HMM, emission N(mean[s], 1); predict "State" before every observe
template <std::size_t N>
void hmm(const std::array<double, N>& ys)
{
    constexpr int k = 3;
    static const std::array<double, k> mean{{-1, 0, 1}};
    static const std::array<std::array<double, k>, k> A{{{{0.1, 0.5, 0.4}}, {{0.2, 0.2, 0.6}}, {{0.15, 0.15, 0.7}}}};
    boost::random::uniform_smallint<std::size_t> init{0, 2};
    auto s = cpprob::sample(init, true);
    cpprob::predict(s, "State");
    boost::random::normal_distribution<> lik{mean[s], 1};
    cpprob::observe(lik, ys[0]);
    for (std::size_t t = 1; t < N; ++t) {
        boost::random::discrete_distribution<std::size_t> next{A[s].begin(), A[s].end()};
        s = cpprob::sample(next, true);
        cpprob::predict(s, "State");
        lik = boost::random::normal_distribution<>{mean[s], 1};
        cpprob::observe(lik, ys[t]);
    }
}

// The Gaussian model with its prior simulated by rejection sampling from the density alone (restates reference
// include/models/models.hpp:82-112): the number of sample statements differs from particle to particle.
template <class Real = double>
void gaussian_by_rejection(const Real y1, const Real y2)
{
    const Real mu0 = 1, s0 = std::sqrt(5), s = std::sqrt(2);
    const boost::math::normal_distribution<Real> prior_density(mu0, s0);
    const Real top = boost::math::pdf(prior_density, mu0);
    boost::random::uniform_real_distribution<Real> proposal{mu0 - 20 * s0, mu0 + 20 * s0};
    boost::random::uniform_real_distribution<Real> accept{0, top};
    Real mu;
    {
        cpprob::rejection_sampling guard{};
        do {
            mu = cpprob::sample(proposal, true);
        } while (cpprob::sample(accept, true) > boost::math::pdf(prior_density, mu));
    }
    boost::random::normal_distribution<Real> lik{mu, s};
    cpprob::observe(lik, y1);
    cpprob::observe(lik, y2);
    cpprob::predict(mu, "Mu");
}

// vector-valued statements (restates reference include/models/models.hpp:38-49): a 2-D mean with independent priors
// N(1, sd sqrt 5) and N(2, sd sqrt 3), ONE observe of a 2-vector with sd sqrt 2 per component, ONE NDArray predict "Mu"
template <class Real = double>
void gaussian_2d_unk_mean(const std::vector<Real> y)
{
    cpprob::multivariate_normal_distribution<Real> prior{{1, 2}, {static_cast<Real>(std::sqrt(5)), static_cast<Real>(std::sqrt(3))}};
    const auto mu = cpprob::sample(prior, true);
    cpprob::multivariate_normal_distribution<Real> lik{mu.begin(), mu.end(), static_cast<Real>(std::sqrt(2))};
    cpprob::observe(lik, y);
    cpprob::predict(mu, "Mu");
}

// Not in the reference: a model over the remaining scalar distributions of the scope table:
// uniform prior on a Poisson rate, two counts observed, predict "Rate".  Only runs through the generic
// device path (no hand-fused kernel exists for it).
template <class Real = double>
void poisson_rate(const int k1, const int k2)
{
    boost::random::uniform_real_distribution<Real> prior{0.5, 10};
    const Real rate = cpprob::sample(prior, true);
    boost::random::poisson_distribution<int, Real> lik{rate};
    cpprob::observe(lik, k1);
    cpprob::observe(lik, k2);
    cpprob::predict(rate, "Rate");
}

// Two test models for the replay machinery (not in the reference).  second_order: x_t depends on the TWO previous states;
// running_mean: x_t depends on the mean of ALL previous states (no finite window: the whole trace has to be replayed).
template <std::size_t N>
void second_order(const std::array<double, N>& y)
{
    double x1 = 0, x2 = 0;
    for (std::size_t t = 0; t < N; ++t) {
        boost::random::normal_distribution<> transition{0.5 * x1 + 0.3 * x2, 1};
        const double x = cpprob::sample(transition, true);
        boost::random::normal_distribution<> emission{x, 1};
        cpprob::observe(emission, y[t]);
        cpprob::predict(x, "State");
        x2 = x1; x1 = x;
    }
}
template <std::size_t N>
void running_mean(const std::array<double, N>& y)
{
    double sum = 0;
    for (std::size_t t = 0; t < N; ++t) {
        boost::random::normal_distribution<> transition{t ? sum / static_cast<double>(t) : 0.0, 1};
        const double x = cpprob::sample(transition, true);
        boost::random::normal_distribution<> emission{x, 1};
        cpprob::observe(emission, y[t]);
        cpprob::predict(x, "State");
        sum += x;
    }
}

// rare_memory: first order on almost every trace -- but a state beyond 3.6 drags the FIRST state back in.  A handful of host traces
// (the Markov probe) will not see that happen; thousands of particles on the device do, every run.
template <std::size_t N>
void rare_memory(const std::array<double, N>& y)
{
    double x0 = 0, x1 = 0;
    for (std::size_t t = 0; t < N; ++t) {
        boost::random::normal_distribution<> transition{0.6 * x1 + (std::fabs(x1) > 3.6 ? 0.8 * x0 : 0.0), 1};
        const double x = cpprob::sample(transition, true);
        boost::random::normal_distribution<> emission{x, 2};
        cpprob::observe(emission, y[t]);
        cpprob::predict(x, "State");
        if (t == 0) x0 = x;
        x1 = x;
    }
}

// random_scale: the emission's standard deviation is itself sampled, so the density at the observe statement's mode differs from
// particle to particle -- there is no host-known bound of a step's log-likelihood (the engine's fixed-point weights then take the
// generation's exact maximum as their reference).
template <std::size_t N>
void random_scale(const std::array<double, N>& y)
{
    double x = 0;
    for (std::size_t t = 0; t < N; ++t) {
        boost::random::uniform_real_distribution<> scale{0.5, 2.0};
        const double sd = cpprob::sample(scale, true);
        boost::random::normal_distribution<> transition{x, 1};
        x = cpprob::sample(transition, true);
        boost::random::normal_distribution<> emission{x, sd};
        cpprob::observe(emission, y[t]);
        cpprob::predict(x, "State");
    }
}

// One statement triple -- sample, address-less predict, observe of the sampled value -- per distribution of the library
// (restates the statement sequence of reference src/models/models.cpp:13-47; the two ints are unused there too; a template here so that
// host programs link the model library's instantiation: registered_models.hpp).  Every predict gets
// its address from its call site; the last triple is vector-valued (four independent normal components).
template <class Int = int>
void all_distr(Int, Int)
{
    boost::random::normal_distribution<> normal{1, 2};
    const auto normal_val = cpprob::sample(normal, true);
    cpprob::predict(normal_val);
    cpprob::observe(normal, normal_val);

    boost::random::uniform_smallint<> small{2, 7};
    const auto small_val = cpprob::sample(small, true);
    cpprob::predict(small_val);
    cpprob::observe(small, small_val);

    boost::random::uniform_real_distribution<> unif{2, 9.5};
    const auto unif_val = cpprob::sample(unif, true);
    cpprob::predict(unif_val);
    cpprob::observe(unif, unif_val);

    boost::random::poisson_distribution<> poisson(0.8);
    const auto poisson_val = cpprob::sample(poisson, true);
    cpprob::predict(poisson_val);
    cpprob::observe(poisson, poisson_val);

    cpprob::multivariate_normal_distribution<> multi{{1, 2, 3, 4}, {2, 1, 5, 3}};
    const auto multi_val = cpprob::sample(multi, true);
    cpprob::predict(multi_val);
    cpprob::observe(multi, multi_val);
}

}  // namespace models
#endif
