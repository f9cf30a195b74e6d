// cpprob::StatsPrinter -- reads `<file>.ids/.int/.real` back and prints the posterior estimators with the output layout of
// reference include/cpprob/postprocess/stats_printer.hpp:25-120 (real predicts are NDArrays: a scalar prints as a bare
// number, a vector-valued predict as `[m0 m1 ...]`, elementwise).  The k-th hit of a predict address inside one trace
// belongs to the k-th distribution of that address (:106-118).
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
    using RealDistr = EmpiricalDistribution<NDArray<double>>;
    using IntDistr = EmpiricalDistribution<int>;

public:
    explicit StatsPrinter(const std::string& file_path) : base_{file_path}
    {
        std::ifstream ids((base_ + ".ids").c_str());
        if (!ids) {
            std::cerr << base_ + ".ids" << " not found." << std::endl;         // stats_printer.hpp:29-32
            return;
        }
        std::string address;
        while (std::getline(ids, address)) addresses_.push_back(address);
        read_points(base_ + ".int", ints_);
        read_points(base_ + ".real", reals_);
    }

    friend std::ostream& operator<<(std::ostream& os, const StatsPrinter& sp)
    {
        sp.print_table(os, ".real", sp.reals_, [](std::ostream& o, const RealDistr& d) {
            const auto mean = d.mean();
            o << "  Mean: " << mean << std::endl << "  Variance: " << d.variance(mean) << std::endl;
        });
        sp.print_table(os, ".int", sp.ints_, [](std::ostream& o, const IntDistr& d) {
            o << "  Distribution:\n";
            const auto pmf = d.distribution();
            for (const auto& value_prob : pmf) o << "    " << value_prob.first << ": " << value_prob.second << std::endl;
            o << "  MAP: " << d.max_a_posteriori(pmf) << std::endl;
            o << "  Num points: " << d.num_points() << std::endl;
        });
        return os;
    }

    // programmatic access (not in the reference): distributions of predict address `id`
    const std::vector<RealDistr>& real(std::size_t id = 0) const { return reals_.at(id); }
    const std::vector<IntDistr>& integer(std::size_t id = 0) const { return ints_.at(id); }
    const std::vector<std::string>& ids() const { return addresses_; }

private:
    std::string base_;
    std::vector<std::string> addresses_;
    std::map<std::size_t, std::vector<IntDistr>> ints_;
    std::map<std::size_t, std::vector<RealDistr>> reals_;

    // one block per file kind: "Estimators for <file><ext>", then `<address>[ k]:` and the body for every distribution
    template <class Table, class Body>
    void print_table(std::ostream& os, const char* ext, const Table& table, Body body) const
    {
        for (const auto& entry : table) {
            os << "Estimators for " << base_ << ext << std::endl;
            const auto& hits = entry.second;
            for (std::size_t k = 0; k < hits.size(); ++k) {
                os << addresses_[entry.first];
                if (hits.size() > 1) os << ' ' << k;
                os << ':' << std::endl;
                body(os, hits[k]);
            }
        }
    }

    // every line is one trace: `([(id value) ...] logw)`
    template <class Distr>
    static void read_points(const std::string& path, std::map<std::size_t, std::vector<Distr>>& table)
    {
        using Value = decltype(std::declval<Distr>().max_a_posteriori());
        std::ifstream in(path.c_str());
        std::string line;
        while (in && std::getline(in, line)) {
            std::pair<std::vector<std::pair<std::size_t, Value>>, double> trace;
            std::istringstream fields(line);
            if (!text::get(fields, trace)) {
                std::cerr << "Bad format in line:\n" << line << std::endl;       // stats_printer.hpp:100-103
                std::exit(EXIT_FAILURE);
            }
            std::map<std::size_t, std::size_t> seen;                             // hits of each address in THIS trace so far
            for (const auto& id_value : trace.first) {
                std::vector<Distr>& per_hit = table[id_value.first];
                const std::size_t k = seen[id_value.first]++;
                if (per_hit.size() <= k) per_hit.resize(k + 1);
                per_hit[k].add_point(id_value.second, trace.second);
            }
        }
    }
};

}  // namespace cpprob
#endif
