// TEST INFRASTRUCTURE ONLY -- oracle restatement of `platanus_b assemble` (assemble.cpp:116-782).
#pragma once
#include "counter.hpp"
#include <string>
#include <vector>

namespace pbo {

struct AssembleOptions {                 // defaults: assemble.cpp:55-70
    std::string out_prefix = "out";
    std::vector<std::string> files;
    unsigned k0 = 32;
    double max_k_ratio = 0.5;            // -K
    unsigned step = 10;                  // -s
    unsigned n_cutoff = 0;               // -n (0 = auto)
    unsigned min_cov = 1;                // -c
    double safety = 10.0;                // -a
    std::string bubble = "0";            // -u (kept as text: atoi() decides whether the final crush runs, Q1)
    double branch = 0.5;                 // -d
    std::string expected_cov = "";       // -e
    unsigned long long memory_gb = 16;   // -m
    bool kmer_occ_only = false;
    bool repeat = false;
    bool verbose = true;
};

struct KmerSchedule {
    std::vector<unsigned> k;
    std::vector<u64> cutoff;
};

// assemble.cpp:657-736
double calc_log_probability_join(u64 coverage_cutoff, double avg_cov, double avg_len, u64 large_k, u64 small_k);
u64 decrease_coverage_cutoff(u64 coverage_cutoff, double avg_cov, double avg_len, double min_log_p_join, u64 large_k, u64 small_k);
u64 extend_kmer(double min_log_p_join, double avg_cov, double avg_len, u64 min_cov, double max_k_ratio,
                std::vector<unsigned> &k, std::vector<u64> &cutoff, u64 step, bool verbose);

int run_assemble(const AssembleOptions &opt, const ReadSet &reads);

} // namespace pbo
