// Step kernels built per step for the registered state-space models (cpprob/gpu.hpp: model_step_kernel_at).  One SOURCE, compiled
// several times by cpprob_amd/build.py -- -DCPPROB_STEPS_MODEL=<k> -DCPPROB_STEPS_PART=<i> -DCPPROB_STEPS_PARTS=<n> -- with the flags
// that let the optimiser see through the unrolled model (the comment at CPPROB_REGISTER_MODEL_STEPS); every object registers its
// builds in the model's table when the library loads, and cpprob::inference launches them where a step's thresholds are theirs.
#include "cpprob/gpu.hpp"

#pragma clang force_cuda_host_device begin
#if defined(CPPROB_USE_REFERENCE_MODELS)
#include "models/models.hpp"
#else
#include "target_models.hpp"
#endif
#pragma clang force_cuda_host_device end

#ifndef CPPROB_STEPS_PART
#define CPPROB_STEPS_PART 0
#define CPPROB_STEPS_PARTS 1
#endif
#if CPPROB_STEPS_MODEL == 0
CPPROB_REGISTER_MODEL_STEPS(models::hmm<16>, 16, 1, CPPROB_STEPS_PART, CPPROB_STEPS_PARTS);
#elif CPPROB_STEPS_MODEL == 1
CPPROB_REGISTER_MODEL_STEPS(models::linear_gaussian_1d<25>, 25, 1, CPPROB_STEPS_PART, CPPROB_STEPS_PARTS);
#elif CPPROB_STEPS_MODEL == 2
CPPROB_REGISTER_MODEL_STEPS(models::hmm<128>, 128, 1, CPPROB_STEPS_PART, CPPROB_STEPS_PARTS);
#elif CPPROB_STEPS_MODEL == 3
CPPROB_REGISTER_MODEL_STEPS(models::linear_gaussian_1d<100>, 100, 1, CPPROB_STEPS_PART, CPPROB_STEPS_PARTS);
#else
#error "CPPROB_STEPS_MODEL: 0 hmm<16>, 1 linear_gaussian_1d<25>, 2 hmm<128>, 3 linear_gaussian_1d<100>"
#endif
