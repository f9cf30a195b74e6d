// cpprob::StatsPrinter -- reads `<file>.ids/.int/.real` back and prints the posterior estimators, same
// output layout as reference include/cpprob/postprocess/stats_printer.hpp:25-120 (real predicts are NDArrays:
// a scalar prints as a bare number, a vector-valued predict as `[m0 m1 ...]`, elementwise).
// The k-th hit of a predict address inside one trace goes to the k-th distribution (:106-118).
#ifndef CPPROB_COMPAT_STATS_PRINTER_HPP
#define CPPROB_COMPAT_STATS_PRINTER_HPP

#include <cstdlib>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>
#include <utility>
#include <vector>

#include "cpprob/postprocess/empirical_distribution.hpp"
#include "cpprob/serialization.hpp"

namespace cpprob {

class StatsPrinter {
public:
    explicit StatsPrinter(const std::string& file_path) : file_name_{file_path}
    {
        std::ifstream ids_file((file_path + ".ids").c_str());
        if (!ids_file.is_open()) {
            std::cerr << file_path + ".ids" << " not found." << std::endl;     // stats_printer.hpp:29-32
            return;
        }
        for (std::string line; std::getline(ids_file, line);) ids_.emplace_back(std::move(line));
        load_distr(file_path + ".int", int_distr_);
        load_distr(file_path + ".real", real_distr_);
    }

    friend std::ostream& operator<<(std::ostream& out, const StatsPrinter& sp)
    {
        for (const auto& kv : sp.real_distr_) {
            out << "Estimators for " << sp.file_name_ << ".real" << std::endl;
            std::size_t i = 0;
            for (const auto& emp : kv.second) {
                out << sp.ids_[kv.first];
                if (kv.second.size() > 1) out << ' ' << i;
                out << ':' << std::endl;
                const auto mean = emp.mean();
                out << "  Mean: " << mean << std::endl << "  Variance: " << emp.variance(mean) << std::endl;
                ++i;
            }
        }
        for (const auto& kv : sp.int_distr_) {
            out << "Estimators for " << sp.file_name_ << ".int" << std::endl;
            std::size_t i = 0;
            for (const auto& emp : kv.second) {
                out << sp.ids_[kv.first];
                if (kv.second.size() > 1) out << ' ' << i;
                out << ':' << std::endl << "  Distribution:\n";
                const auto distr = emp.distribution();
                for (const auto& x_w : distr) out << "    " << x_w.first << ": " << x_w.second << std::endl;
                out << "  MAP: " << emp.max_a_posteriori(distr) << std::endl;
                out << "  Num points: " << emp.num_points() << std::endl;
                ++i;
            }
        }
        return out;
    }

    // programmatic access (not in the reference): distributions of predict address `id`
    const std::vector<EmpiricalDistribution<NDArray<double>>>& real(std::size_t id = 0) const { return real_distr_.at(id); }
    const std::vector<EmpiricalDistribution<int>>& integer(std::size_t id = 0) const { return int_distr_.at(id); }
    const std::vector<std::string>& ids() const { return ids_; }

private:
    std::map<std::size_t, std::vector<EmpiricalDistribution<int>>> int_distr_;
    std::map<std::size_t, std::vector<EmpiricalDistribution<NDArray<double>>>> real_distr_;
    std::vector<std::string> ids_;
    std::string file_name_;

    template <class T>
    void load_distr(const std::string& file_name, std::map<std::size_t, std::vector<EmpiricalDistribution<T>>>& distributions)
    {
        std::ifstream file(file_name.c_str());
        if (!file.is_open()) return;
        for (std::string line; std::getline(file, line);) {
            std::map<std::size_t, std::size_t> hits;
            std::pair<std::vector<std::pair<std::size_t, T>>, double> predicts;
            std::istringstream iss(line);
            if (!text::get(iss, predicts)) {
                std::cerr << "Bad format in line:\n" << line << std::endl;       // stats_printer.hpp:100-103
                std::exit(EXIT_FAILURE);
            }
            for (const auto& elem : predicts.first) {
                auto& vec = distributions[elem.first];
                auto& k = hits[elem.first];
                if (k == vec.size()) vec.emplace_back();
                vec[k].add_point(elem.second, predicts.second);
                ++k;
            }
        }
    }
};

}  // namespace cpprob
#endif
