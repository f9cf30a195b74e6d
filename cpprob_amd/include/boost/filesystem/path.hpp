// Clean-room stand-in for boost/filesystem/path.hpp: the subset cpprob::inference's signature and
// callers use (construction from strings, string(), c_str(), operator/, operator+=).
#ifndef CPPROB_COMPAT_BOOST_FILESYSTEM_PATH_HPP
#define CPPROB_COMPAT_BOOST_FILESYSTEM_PATH_HPP
#include <string>
#include <ostream>
namespace boost { namespace filesystem {
class path {
public:
    path() = default;
    path(const char* s) : s_(s) {}
    path(const std::string& s) : s_(s) {}
    const std::string& string() const { return s_; }
    const char* c_str() const { return s_.c_str(); }
    bool empty() const { return s_.empty(); }
    path& operator/=(const path& o) { if (!s_.empty() && s_.back() != '/') s_ += '/'; s_ += o.s_; return *this; }
    path& operator+=(const std::string& o) { s_ += o; return *this; }
    friend path operator/(path a, const path& b) { a /= b; return a; }
    friend std::ostream& operator<<(std::ostream& os, const path& p) { return os << '"' << p.s_ << '"'; }
private:
    std::string s_;
};
}}
#endif
