// (moved: the model translation units use the same wavefront primitives -- cpprob/gpu.hpp)
#pragma once
#include "cpprob/detail/wave.hpp"
