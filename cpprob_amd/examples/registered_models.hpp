// Host view of the model library (libcpprob_models.so, built from registered_models.hip).
// The `extern template` lines stop a host translation unit from instantiating its own copy of a model:
// cpprob::inference finds a model's device code by the ADDRESS of the function it is given, so host
// code must refer to the very instantiation the model library registered (the reference links its
// models from src/models/*.cpp the same way).  Alternative: link the host program with -rdynamic.
#ifndef CPPROB_EXAMPLES_REGISTERED_MODELS_HPP
#define CPPROB_EXAMPLES_REGISTERED_MODELS_HPP
#include "target_models.hpp"

namespace models {
extern template void gaussian_unknown_mean<double>(double, double);
extern template void gaussian_readme<double>(double, double);
extern template void linear_gaussian_1d<25>(const std::array<double, 25>&);
extern template void linear_gaussian_1d<100>(const std::array<double, 100>&);
extern template void hmm<16>(const std::array<double, 16>&);
extern template void hmm<128>(const std::array<double, 128>&);
extern template void poisson_rate<double>(int, int);
extern template void gaussian_2d_unk_mean<double>(std::vector<double>);
extern template void gaussian_by_rejection<double>(double, double);
extern template void all_distr<int>(int, int);
extern template void second_order<12>(const std::array<double, 12>&);
extern template void running_mean<12>(const std::array<double, 12>&);
extern template void rare_memory<12>(const std::array<double, 12>&);
}
#endif
