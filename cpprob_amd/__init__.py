"""cpprob_amd -- MI355X (gfx950) SIS / SMC inference engine behind the CPProb API.

The product is native: `lib/libcpprob_hip.so` (HIP kernels + C ABI, include/cpprob_hip.h) and the
C++14 compatibility headers in `include/`.  This Python package is the test / bench harness
around the C ABI (ctypes) and the multi-GPU driver (torch.distributed over RCCL).
There is no CPU fallback: without the built library or without a GPU, calls fail loudly.
"""
from . import capi  # noqa: F401
from .capi import (ALG_SIS, ALG_SMC, MODEL_GAUSSIAN_README, MODEL_GAUSSIAN_UNKNOWN_MEAN, MODEL_GAUSSIAN_2D_UNKNOWN_MEAN, MODEL_HMM3,  # noqa: F401
                   MODEL_LINEAR_GAUSSIAN_1D, MODEL_HMM_TABLE, RESAMPLE_MULTINOMIAL, RESAMPLE_STRATIFIED, RESAMPLE_SYSTEMATIC,
                   SCOPE_GLOBAL, SCOPE_ISLAND, SCOPE_EXCHANGE, Engine, Group, CpprobHipError, load_library)

__version__ = "0.1.0"
