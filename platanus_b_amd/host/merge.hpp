// `platanus_b merge` (merge.cpp:26-147): re-assemble contigs at a larger k -- SURVEY 8(f) "next" row 2.  Its graph path is the
// assemble path (makeInitialBruijnGraph, cutBranchIterative, crushBubbleIterative, saveContigSimple); the only new piece
// is Counter::makeKmerReadDistributionFromContig (counter.h:510-593), which is the contig-string -> k-mer table kernel the
// multi-k rounds already use (value = max(contig coverage, -c)).  This header holds the host parts shared with the oracle.
#pragma once
#include "kmer.hpp"
#include "errors.hpp"
#include <algorithm>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "kmer_divide.hpp"      // find_number_after (common.h:647-663)

namespace pbh {

struct MergeOptions {                         // defaults: merge.cpp:29-39
    std::string out_prefix = "out";
    std::vector<std::string> files;
    unsigned min_cov = 1;                     // -c
    double k_factor = 1.0;                    // -k
    double len_factor = 2.0;                  // -l
    std::string bubble = "0";                 // -u
    double branch = 0.5;                      // -d
    unsigned long long memory_gb = 16;        // -m
};

struct MergeContigs {
    std::vector<std::string> seq;             // codes, pieces cut at 'N'
    std::vector<u16> coverage;
};

// Contig::readFasta in CUTN mode (common.h:540-575,595-645): a record shorter than `min_len` is dropped whole, the rest is cut
// at every 'N'; each piece inherits the record's _cov value
inline void read_contigs_cut_n(const std::string &path, u64 min_len, MergeContigs &mc)
{
    std::ifstream ifs(path);
    if (!ifs) throw file_error(path);
    std::string line, contig;
    bool have = false, first = true;
    u16 cov = 0;
    auto header_cov = [&](const std::string &h) -> u16 {
        const u64 c = find_number_after(h.c_str(), "cov");
        if (first) { first = false; return (u16)std::min<u64>(c, 65535); }   // common.h:622-623 vs 636: only the first record saturates
        return (u16)c;
    };
    auto push = [&]() {
        if (contig.size() < min_len) return;
        std::string piece;
        for (size_t i = 0; i < contig.size(); ++i) {
            if (contig[i] != 'N') piece.push_back((char)char2bin(contig[i]));
            else if (!piece.empty()) { mc.seq.push_back(piece); mc.coverage.push_back(cov); piece.clear(); }
        }
        if (!piece.empty()) { mc.seq.push_back(piece); mc.coverage.push_back(cov); }
    };
    while (std::getline(ifs, line)) {
        if (!have) {
            if (!line.empty() && line[0] == '>') { have = true; cov = header_cov(line); }
            continue;
        }
        if (line.empty() || line[0] != '>') contig += line;
        else { push(); contig.clear(); cov = header_cov(line); }
    }
    if (have) push();
}

// Contig::getSeqLengthMedian + calculateAverageCoverageExcludingOutlier (common.h:747-772,803-808)
inline double merge_average_coverage(const MergeContigs &mc)
{
    std::vector<long> lens;
    for (const std::string &s : mc.seq) lens.push_back((long)s.size());
    std::sort(lens.begin(), lens.end());
    const long min_len = lens[lens.size() / 2];
    long sum = 0, num = 0;
    for (size_t i = 0; i < mc.seq.size(); ++i)
        if ((long)mc.seq[i].size() >= min_len) { sum += mc.coverage[i] * (long)mc.seq[i].size(); num += (long)mc.seq[i].size(); }
    const double mean = static_cast<double>(sum) / num;
    sum = num = 0;
    for (size_t i = 0; i < mc.seq.size(); ++i)
        if ((long)mc.seq[i].size() >= min_len && mc.coverage[i] >= mean / 100.0 && mc.coverage[i] <= mean * 100.0) {
            sum += mc.coverage[i] * (long)mc.seq[i].size();
            num += (long)mc.seq[i].size();
        }
    return static_cast<double>(sum) / num;
}

// Contig::readContigFP (common.h:774-800): the binary contig records of assemble's contigFP as pieces (-repeat's mergeContig).
// RECORD has .seq (packed words, position p counts from the 3' end), .length, .cov.
template <typename RECORD>
inline void contig_records_to_pieces(const std::vector<RECORD> &recs, MergeContigs &mc)
{
    for (const RECORD &r : recs) {
        std::string codes(r.length, 0);
        for (u64 j = 0; j < r.length; ++j) {
            const u64 p = r.length - 1 - j;
            codes[j] = (char)((r.seq[p / 32] >> ((p % 32) * 2)) & 3);
        }
        mc.seq.push_back(codes);
        mc.coverage.push_back((u16)r.cov);
    }
}

inline u64 first_header_read_length(const std::string &path)        // common.h:686-701
{
    std::ifstream ifs(path);
    std::string line;
    while (std::getline(ifs, line)) if (!line.empty() && line[0] == '>') break;
    const u64 r = find_number_after(line.c_str(), "read");
    return r > 1 ? r : 100;
}

} // namespace pbh
