// Closes the device view opened by cpprob/device_view_begin.hpp.
#pragma clang force_cuda_host_device end
#undef vector
#undef multivariate_normal_distribution
#undef NDArray
