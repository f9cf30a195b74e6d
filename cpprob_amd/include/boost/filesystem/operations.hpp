// Clean-room stand-in for boost/filesystem/operations.hpp: exists(), create_directory().
#ifndef CPPROB_COMPAT_BOOST_FILESYSTEM_OPERATIONS_HPP
#define CPPROB_COMPAT_BOOST_FILESYSTEM_OPERATIONS_HPP
#include <sys/stat.h>
#include "boost/filesystem/path.hpp"
namespace boost { namespace filesystem {
inline bool exists(const path& p) { struct stat st; return ::stat(p.c_str(), &st) == 0; }
inline bool create_directory(const path& p) { return ::mkdir(p.c_str(), 0777) == 0; }
}}
#endif
