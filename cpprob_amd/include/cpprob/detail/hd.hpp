// Host/device annotation used by the compatibility headers.  Under hipcc the statement API and the
// distribution shims are callable from device code (the model source is compiled for the GPU);
// under a plain C++14 host compiler the macros vanish.
#ifndef CPPROB_DETAIL_HD_HPP
#define CPPROB_DETAIL_HD_HPP
#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define CPPROB_HD __host__ __device__
#define CPPROB_DEVICE_COMPILE_AVAILABLE 1
#else
#define CPPROB_HD
#endif
#endif
