// Model translation unit (hipcc, C++17): compiles the model source for host AND device and registers
// the instantiations the tests / examples / bench use.  With CPPROB_USE_REFERENCE_MODELS the
// reference's own include/models/models.hpp is compiled UNCHANGED (only where /root/reference exists).
#include "cpprob/gpu.hpp"

#pragma clang force_cuda_host_device begin
#if defined(CPPROB_USE_REFERENCE_MODELS)
#include "models/models.hpp"
#else
#include "target_models.hpp"
#endif
#pragma clang force_cuda_host_device end

// The same source once more as its DEVICE VIEW (cpprob/device_view_begin.hpp): std::vector / NDArray /
// multivariate_normal_distribution spelled as their fixed-capacity counterparts, so that models with vector-valued statements run
// through the generic path too.
#if defined(CPPROB_USE_REFERENCE_MODELS)
#undef INCLUDE_MODELS_HPP_
#else
#undef CPPROB_EXAMPLES_TARGET_MODELS_HPP
#endif
namespace cpprob_device_view {
#include "cpprob/device_view_begin.hpp"
#if defined(CPPROB_USE_REFERENCE_MODELS)
#include "models/models.hpp"
#else
#include "target_models.hpp"
#endif
#include "cpprob/device_view_end.hpp"
}  // namespace cpprob_device_view

// generic path: the model body itself runs on the GPU
CPPROB_REGISTER_MODEL(models::gaussian_unknown_mean<double>);
CPPROB_REGISTER_MODEL(models::linear_gaussian_1d<25>);
CPPROB_REGISTER_MODEL(models::linear_gaussian_1d<100>);
CPPROB_REGISTER_MODEL(models::hmm<16>);
CPPROB_REGISTER_MODEL(models::hmm<128>);
#if defined(CPPROB_USE_REFERENCE_MODELS)
CPPROB_REGISTER_MODEL(models::normal_rejection_sampling<double>);   // reference models.hpp:82-112, untouched
CPPROB_REGISTER_FUNCTOR(models::Gauss<double>);                      // functor model, reference models.hpp:51-65
#endif
#if !defined(CPPROB_USE_REFERENCE_MODELS)
CPPROB_REGISTER_MODEL(models::gaussian_readme<double>);
CPPROB_REGISTER_MODEL(models::poisson_rate<double>);
CPPROB_REGISTER_FUNCTOR(models::GaussFunctor<double>);
CPPROB_REGISTER_MODEL(models::gaussian_by_rejection<double>);
CPPROB_REGISTER_MODEL(models::second_order<12>);
CPPROB_REGISTER_MODEL(models::running_mean<12>);
CPPROB_REGISTER_MODEL(models::rare_memory<12>);
CPPROB_REGISTER_MODEL(models::random_scale<12>);
CPPROB_REGISTER_BUILTIN(models::gaussian_readme<double>, CPPROB_HIP_MODEL_GAUSSIAN_README);
#endif
// vector-valued statements through the generic path: the device view of the same function
CPPROB_REGISTER_MODEL_VIEW(models::gaussian_2d_unk_mean<double>, cpprob_device_view::models::gaussian_2d_unk_mean<double>);
#if !defined(CPPROB_USE_REFERENCE_MODELS)
CPPROB_REGISTER_MODEL_VIEW(models::all_distr<int>, cpprob_device_view::models::all_distr<int>);
#endif
// fast path: the hand-fused kernels of libcpprob_hip for the same functions
CPPROB_REGISTER_BUILTIN(models::gaussian_unknown_mean<double>, CPPROB_HIP_MODEL_GAUSSIAN_UNKNOWN_MEAN);
CPPROB_REGISTER_BUILTIN(models::linear_gaussian_1d<25>, CPPROB_HIP_MODEL_LINEAR_GAUSSIAN_1D);
CPPROB_REGISTER_BUILTIN(models::linear_gaussian_1d<100>, CPPROB_HIP_MODEL_LINEAR_GAUSSIAN_1D);
CPPROB_REGISTER_BUILTIN(models::hmm<16>, CPPROB_HIP_MODEL_HMM3);
CPPROB_REGISTER_BUILTIN(models::hmm<128>, CPPROB_HIP_MODEL_HMM3);
// vector-valued statements: std::vector / NDArray cannot exist in device code, so this model has a built-in kernel only
CPPROB_REGISTER_BUILTIN(models::gaussian_2d_unk_mean<double>, CPPROB_HIP_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN);
