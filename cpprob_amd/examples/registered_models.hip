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
CPPROB_REGISTER_BUILTIN(models::gaussian_readme<double>, CPPROB_HIP_MODEL_GAUSSIAN_README);
#endif
// fast path: the hand-fused kernels of libcpprob_hip for the same functions
CPPROB_REGISTER_BUILTIN(models::gaussian_unknown_mean<double>, CPPROB_HIP_MODEL_GAUSSIAN_UNKNOWN_MEAN);
CPPROB_REGISTER_BUILTIN(models::linear_gaussian_1d<25>, CPPROB_HIP_MODEL_LINEAR_GAUSSIAN_1D);
CPPROB_REGISTER_BUILTIN(models::linear_gaussian_1d<100>, CPPROB_HIP_MODEL_LINEAR_GAUSSIAN_1D);
CPPROB_REGISTER_BUILTIN(models::hmm<16>, CPPROB_HIP_MODEL_HMM3);
CPPROB_REGISTER_BUILTIN(models::hmm<128>, CPPROB_HIP_MODEL_HMM3);
// vector-valued statements: std::vector / NDArray cannot exist in device code, so this model has a built-in kernel only
CPPROB_REGISTER_BUILTIN(models::gaussian_2d_unk_mean<double>, CPPROB_HIP_MODEL_GAUSSIAN_2D_UNKNOWN_MEAN);
