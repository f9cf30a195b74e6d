// Clean-room stand-in for boost/math/constants/constants.hpp: pi only.
#ifndef CPPROB_COMPAT_BOOST_MATH_CONSTANTS_HPP
#define CPPROB_COMPAT_BOOST_MATH_CONSTANTS_HPP
#include "cpprob/detail/hd.hpp"
namespace boost { namespace math { namespace constants {
template <class T> CPPROB_HD constexpr T pi() { return static_cast<T>(3.141592653589793238462643383279502884L); }
template <class T> CPPROB_HD constexpr T root_two_pi() { return static_cast<T>(2.506628274631000502415765284811045253L); }
}}}
#endif
