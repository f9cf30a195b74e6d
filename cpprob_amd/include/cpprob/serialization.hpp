// Text grammar of the posterior files and of observes given on a command line -- an independent
// implementation of the format the reference defines in include/cpprob/serialization.hpp:
//   pair / tuple  (a b ...)      vector / array  [a b ...]      scalars separated by spaces
// Printing uses whatever flags the stream carries; StateInfer::dump_predicts sets
// std::scientific and precision digits10 (src/cpprob/state.cpp:262-267).  Output is checked byte for
// byte against the reference's own printer in tests (tests/golden/serialization.json).
#ifndef CPPROB_COMPAT_SERIALIZATION_HPP
#define CPPROB_COMPAT_SERIALIZATION_HPP

#include <array>
#include <cstddef>
#include <fstream>
#include <istream>
#include <ostream>
#include <sstream>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace cpprob {
namespace text {

// ---- writer ---------------------------------------------------------------------------------
template <class T> void put(std::ostream& os, const T& v) { os << v; }
template <class A, class B> void put(std::ostream& os, const std::pair<A, B>& p);
template <class T> void put(std::ostream& os, const std::vector<T>& v);
template <class T, std::size_t N> void put(std::ostream& os, const std::array<T, N>& v);

template <class It>
void put_range(std::ostream& os, It first, It last, char open, char close)
{
    os << open;
    for (It it = first; it != last; ++it) { if (it != first) os << ' '; put(os, *it); }
    os << close;
}
template <class A, class B> void put(std::ostream& os, const std::pair<A, B>& p) { os << '('; put(os, p.first); os << ' '; put(os, p.second); os << ')'; }
template <class T> void put(std::ostream& os, const std::vector<T>& v) { put_range(os, v.begin(), v.end(), '[', ']'); }
template <class T, std::size_t N> void put(std::ostream& os, const std::array<T, N>& v) { put_range(os, v.begin(), v.end(), '[', ']'); }

// ---- reader ---------------------------------------------------------------------------------
inline bool expect(std::istream& is, char c) { char ch; return (is >> std::ws >> ch) && ch == c ? true : (is.setstate(std::ios::failbit), false); }
template <class T> bool get(std::istream& is, T& v) { return static_cast<bool>(is >> v); }
template <class A, class B> bool get(std::istream& is, std::pair<A, B>& p);
template <class T> bool get(std::istream& is, std::vector<T>& v);
template <class T, std::size_t N> bool get(std::istream& is, std::array<T, N>& v);

template <class A, class B> bool get(std::istream& is, std::pair<A, B>& p) { return expect(is, '(') && get(is, p.first) && get(is, p.second) && expect(is, ')'); }
template <class T> bool get(std::istream& is, std::vector<T>& v)
{
    v.clear();
    if (!expect(is, '[')) return false;
    for (;;) {
        char ch;
        if (!(is >> std::ws)) return false;
        ch = static_cast<char>(is.peek());
        if (ch == ']') { is.get(); return true; }
        T x;
        if (!get(is, x)) return false;
        v.push_back(std::move(x));
    }
}
template <class T, std::size_t N> bool get(std::istream& is, std::array<T, N>& v)
{
    if (!expect(is, '[')) return false;
    for (std::size_t i = 0; i < N; ++i) if (!get(is, v[i])) return false;
    return expect(is, ']');
}

template <class Tuple, std::size_t... I>
bool get_tuple(std::istream& is, Tuple& t, std::index_sequence<I...>)
{
    bool ok = true;
    (void)std::initializer_list<int>{(ok = ok && get(is, std::get<I>(t)), 0)...};
    return ok;
}

}  // namespace text

// Observes on a command line / in a file: the elements of the tuple separated by spaces, aggregates in
// brackets -- reference serialization.hpp:259-284 (parse_string / parse_file), src/main.cpp:77-85.
template <class... T>
bool parse_string(const std::string& s, std::tuple<T...>& out)
{
    std::istringstream iss(s);
    if (!text::get_tuple(iss, out, std::index_sequence_for<T...>{})) return false;
    iss >> std::ws;
    return iss.eof();
}

template <class... T>
bool parse_file(const std::string& path, std::tuple<T...>& out)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    return parse_string(ss.str(), out);
}

}  // namespace cpprob
#endif
