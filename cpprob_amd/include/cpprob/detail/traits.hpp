// Compile-time introspection of a model's signature -- what the reference gets from
// Boost.FunctionTypes in include/cpprob/traits.hpp:54-93 and metapriors.hpp:195-210
// (tuple_observes_t): the tuple of decayed parameter types a model takes its observes in.
#ifndef CPPROB_COMPAT_DETAIL_TRAITS_HPP
#define CPPROB_COMPAT_DETAIL_TRAITS_HPP
#include <cstddef>
#include <tuple>
#include <type_traits>

namespace cpprob {
namespace detail {

template <class F, class = void> struct fn_traits;
template <class R, class... P> struct fn_traits<R(P...), void> { using args = std::tuple<std::decay_t<P>...>; static constexpr bool is_function = true; };
template <class R, class... P> struct fn_traits<R (*)(P...), void> : fn_traits<R(P...)> {};
template <class R, class... P> struct fn_traits<R (&)(P...), void> : fn_traits<R(P...)> {};
template <class C, class R, class... P> struct fn_traits<R (C::*)(P...) const, void> { using args = std::tuple<std::decay_t<P>...>; static constexpr bool is_function = false; };
template <class C, class R, class... P> struct fn_traits<R (C::*)(P...), void> { using args = std::tuple<std::decay_t<P>...>; static constexpr bool is_function = false; };
template <class F> struct fn_traits<F, std::enable_if_t<std::is_class<F>::value>> : fn_traits<decltype(&F::operator())> {};

}  // namespace detail

template <class F> using tuple_observes_t = typename detail::fn_traits<std::remove_cv_t<std::remove_reference_t<F>>>::args;
template <class F> constexpr std::size_t num_args() { return std::tuple_size<tuple_observes_t<F>>::value; }

namespace detail {
template <class F, class Tuple, std::size_t... I>
CPPROB_HD inline void call_f_tuple_impl(const F& f, const Tuple& t, std::index_sequence<I...>) { f(std::get<I>(t)...); }
}
// call_f_tuple -- reference include/cpprob/call_function.hpp:75-80
template <class F, class... Args>
CPPROB_HD inline void call_f_tuple(const F& f, const std::tuple<Args...>& args)
{
    detail::call_f_tuple_impl(f, args, std::index_sequence_for<Args...>{});
}

}  // namespace cpprob
#endif
