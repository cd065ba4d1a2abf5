// Host arithmetic of `platanus_b assemble` that decides the k schedule and the per-k coverage cutoffs
// (assemble.cpp:657-736) and the histogram statistics feeding it (counter.h:220-267).  Floating-point expressions
// follow the reference literally (operation order matters for the truncations).  Shared by the product host
// and the oracle.
#pragma once
#include "kmer.hpp"
#include "errors.hpp"
#include <cmath>
#include <iostream>
#include <stdexcept>
#include <vector>

namespace pbh {

// assemble.cpp:700-715
inline double calc_log_probability_join(u64 coverage_cutoff, double avg_cov, double avg_len, u64 large_k, u64 small_k)
{
    double p, s = 0;
    const double c = avg_cov * (avg_len - large_k + 1.0) / avg_len;
    for (u64 i = 0; i < coverage_cutoff; ++i) {
        p = 0;
        for (u64 j = 1; j <= i; ++j) p += log(c) - log(static_cast<double>(j));
        s += exp(p);
    }
    s = exp(-c + log(s));
    return ((large_k - small_k) + 1.0) * (-s);
}

// assemble.cpp:723-736
inline u64 decrease_coverage_cutoff(u64 coverage_cutoff, double avg_cov, double avg_len, double min_log_p_join, u64 large_k, u64 small_k)
{
    if (coverage_cutoff <= 1) return 1;
    u64 i;
    for (i = coverage_cutoff; i > 1; --i)
        if (calc_log_probability_join(i, avg_cov, avg_len, large_k, small_k) > min_log_p_join) break;
    return i;
}

// assemble.cpp:657-694
inline u64 extend_kmer(double min_log_p_join, double avg_cov, double avg_len, u64 min_cov, double max_k_ratio,
                       std::vector<unsigned> &k, std::vector<u64> &cutoff, u64 step, bool verbose)
{
    u64 i, j;
    const u64 min_max_k = static_cast<long>(avg_len * max_k_ratio + 0.5);
    if (verbose) {
        std::cerr << "\nKMER_EXTENSION:" << std::endl;
        std::cerr << "K=" << k[0] << ", KMER_COVERAGE=" << avg_cov * (avg_len - k[0] + 1.0) / avg_len;
        std::cerr << " (>= " << cutoff[0] << "), COVERAGE_CUTOFF=" << cutoff[0] << std::endl;
    }
    for (i = 1; k[i - 1] <= avg_len; ++i) {
        k.push_back(0);
        cutoff.push_back(0);
        for (j = 1; j <= step + 1; ++j) {
            k[i] = k[i - 1] + j;
            cutoff[i] = decrease_coverage_cutoff(cutoff[i - 1], avg_cov, avg_len, min_log_p_join, k[i], k[i - 1]);
            if (cutoff[i] < min_cov) cutoff[i] = min_cov;
            if (k[i - 1] + j > min_max_k && calc_log_probability_join(cutoff[i], avg_cov, avg_len, k[i], k[i - 1]) < min_log_p_join)
                break;
        }
        --(k[i]);
        cutoff[i] = decrease_coverage_cutoff(cutoff[i - 1], avg_cov, avg_len, min_log_p_join, k[i], k[i - 1]);
        if (cutoff[i] < min_cov) cutoff[i] = min_cov;
        if (k[i] == k[i - 1]) break;
        if (verbose) {
            std::cerr << "K=" << k[i] << ", KMER_COVERAGE=" << avg_cov * (avg_len - k[i] + 1.0) / avg_len;
            std::cerr << ", COVERAGE_CUTOFF=" << cutoff[i] << ", PROB_SPLIT=10e"
                      << log10(1.0 - exp(calc_log_probability_join(cutoff[i], avg_cov, avg_len, k[i], k[i - 1]))) << std::endl;
        }
    }
    return i;
}

// counter.h:244-267 on a histogram hist[0..65534], max_occ = highest non-empty bin
inline u64 left_local_minimal_value(const std::vector<u64> &hist, u64 max_occ, u64 window)
{
    if (max_occ <= window) return 0;
    const u64 n = max_occ - window + 2;
    std::vector<u64> w(n, 0);
    u64 i;
    for (i = 0; i < window; ++i) w[1] += hist[1 + i];
    for (i = 2; i < n; ++i) {
        w[i] = w[i - 1] - hist[i - 1] + hist[i + window - 1];
        if (w[i] >= w[i - 1]) break;
    }
    return i <= max_occ ? (i - 1 + window / 2) : (1 + window / 2);
}

// counter.h:220-238
inline double dist_average(const std::vector<u64> &d, u64 start, u64 end)
{
    if (end > d.size() || start > end) throw kmer_dist_error();
    u64 sum = 0, num = 0;
    for (u64 i = start; i <= end && i < d.size(); ++i) { sum += i * d[i]; num += d[i]; }
    if (num == 0) throw kmer_dist_error();
    return (double)sum / num;
}

} // namespace pbh
