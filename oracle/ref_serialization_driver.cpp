// Driver around the REFERENCE's own text grammar (include/cpprob/serialization.hpp,
// compiled from /root/reference where it lies; nothing is copied).  Test
// infrastructure only.  Modes:
//   print-real  : stdin "n_pred v0 v1 ... logw" per line  -> reference-formatted line
//   print-int   : same with integer values
//   parse-real  : stdin reference-grammar lines -> "n id v id v ... logw" (round trip check)
//   parse-int
// Formatting flags are the ones StateInfer::dump_predicts sets (state.cpp:262-267):
// std::scientific, precision = numeric_limits<double>::digits10.
#include <iostream>
#include <iomanip>
#include <limits>
#include <sstream>
#include <string>
#include <utility>
#include <vector>
#include "cpprob/serialization.hpp"

using namespace cpprob;

template <class T>
int do_print()
{
    std::cout.precision(std::numeric_limits<double>::digits10);
    std::cout << std::scientific;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream iss(line);
        std::size_t n;
        iss >> n;
        std::vector<std::pair<std::size_t, T>> preds;
        for (std::size_t k = 0; k < n; ++k) { T v; iss >> v; preds.emplace_back(0, v); }
        double logw;
        iss >> logw;
        std::cout << std::make_pair(preds, logw) << std::endl;
    }
    return 0;
}

template <class T>
int do_parse()
{
    std::cout.precision(17);
    std::string line;
    while (std::getline(std::cin, line)) {
        std::pair<std::vector<std::pair<std::size_t, T>>, double> predicts;
        std::istringstream iss(line);
        if (!(iss >> predicts)) { std::cout << "BAD" << std::endl; continue; }
        std::cout << predicts.first.size();
        for (const auto & e : predicts.first) std::cout << ' ' << e.first << ' ' << e.second;
        std::cout << ' ' << predicts.second << std::endl;
    }
    return 0;
}

int main(int argc, char ** argv)
{
    if (argc < 2) return 2;
    std::string mode = argv[1];
    if (mode == "print-real") return do_print<double>();
    if (mode == "print-int") return do_print<int>();
    if (mode == "parse-real") return do_parse<double>();
    if (mode == "parse-int") return do_parse<int>();
    return 2;
}
