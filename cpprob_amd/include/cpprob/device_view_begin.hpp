// Opens the DEVICE VIEW of a model header (no include guard: a pair of these brackets every second inclusion).
//
//     #include "models/models.hpp"                       // the host view, as ever
//     #undef INCLUDE_MODELS_HPP_                         // the model header's own include guard
//     namespace cpprob_device_view {
//     #include "cpprob/device_view_begin.hpp"
//     #include "models/models.hpp"                       // the same source again
//     #include "cpprob/device_view_end.hpp"
//     }
//     CPPROB_REGISTER_MODEL_VIEW(models::gaussian_2d_unk_mean<double>, cpprob_device_view::models::gaussian_2d_unk_mean<double>);
//
// Between the brackets the three heap-backed names a model with vector-valued statements mentions are spelled as their
// fixed-capacity counterparts (cpprob/detail/device_vector.hpp), and every function is compiled for host and device.  The model
// source is not touched; everything it includes must have been included before (cpprob/gpu.hpp pulls in the library's headers).
#define vector cpprob_device_view_vector
#define multivariate_normal_distribution dev_multivariate_normal_distribution
#define NDArray dev_NDArray
#pragma clang force_cuda_host_device begin
