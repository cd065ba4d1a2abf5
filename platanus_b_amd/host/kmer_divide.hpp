// `platanus_b kmer_divide` (kmer_divide.cpp:26-388, kmer_divide.h:34-128): split or mask contigs by the k-mer occurrence
// of every contig position, looked up in a PREFIX_kmer_occ.bin written by `assemble -kmer_occ_only`.
// SURVEY 8(f) ranks this consumer of the hot path's output first among the "next" rows.  The per-position table
// look-ups (ContigDivider::getOccurrenceArray, kmer_divide.cpp:150-197) are the device part (pb_occ_lookup); everything
// in this header is the host logic around them, shared with the oracle, which fills the same arrays with CPU look-ups.
#pragma once
#include "kmer.hpp"
#include "errors.hpp"
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstring>
#include <deque>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace pbh {

struct DivideOptions {                       // defaults: kmer_divide.cpp:28-38
    std::string out_prefix = "out";
    std::string occ_file;                    // -k
    std::vector<std::string> files;          // -f
    double rate = 0.1;                       // -r
    double mask_rate = 2.0;                  // -R
    bool recalc_cov = false;
    bool mask = false;
};

struct ContigSet {                           // the part of platanus::Contig that kmer_divide uses (common.h:495-560)
    std::vector<std::string> name;           // first whitespace-delimited token of the header
    std::vector<std::string> seq;            // codes 0..3, 4 = N (Char2Bin), one char per base
    size_t size() const { return seq.size(); }
};

// common.h:647-663
inline u64 find_number_after(const char *line, const std::string &tag)
{
    u64 v = 0;
    const char *p = strstr(line, tag.c_str());
    if (p == nullptr) return 1;
    p += tag.length();
    while (*p != '\0' && isdigit((unsigned char)*p)) { v = v * 10 + (u64)(*p - 48); ++p; }
    return v;
}

// Contig::readFasta in NORMAL mode (common.h:595-645) + getReadLength (common.h:676-702) for "read" and "maxK"
inline void read_contig_fasta(const std::string &path, ContigSet &cs, bool first_file, u64 &read_length, u64 &max_k)
{
    std::ifstream ifs(path);
    if (!ifs) throw file_error(path);
    std::string line, contig, header;
    bool have = false;
    auto push = [&]() {
        std::string codes(contig.size(), 0);
        for (size_t i = 0; i < contig.size(); ++i) codes[i] = (char)char2bin(contig[i]);
        cs.seq.push_back(codes);
        std::istringstream ss(header);
        std::string nm;
        ss >> nm;
        cs.name.push_back(nm);
    };
    while (std::getline(ifs, line)) {
        if (!have) {
            if (!line.empty() && line[0] == '>') {
                have = true;
                header = line.substr(1);
                if (first_file) {
                    const u64 r = find_number_after(line.c_str(), "read"), m = find_number_after(line.c_str(), "maxK");
                    read_length = r > 1 ? r : 100;
                    max_k = m > 1 ? m : 100;
                }
            }
            continue;
        }
        if (line.empty() || line[0] != '>') contig += line;
        else { push(); contig.clear(); header = line.substr(1); }
    }
    if (have) push();
}

// PREFIX_kmer_occ.bin -> (k, records); doubleHash.h:280-293 / counter.h:966-972.  Slot numbers are not needed: the table is
// only probed for presence (findValue), so any table with the same (key,count) content answers identically.
inline void read_kmer_occ(const std::string &path, unsigned &k, std::vector<u64> &keys, std::vector<u16> &counts)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw file_error(path);
    u64 hdr[2];
    if (fread(hdr, 8, 2, f) != 2) { fclose(f); throw std::runtime_error("truncated " + path); }
    k = (unsigned)hdr[0];
    const int W = words_for_k(k);
    const size_t skip = W > 1 ? 24 : 0;            // raw object header of BinstrNN records
    std::vector<unsigned char> rec(8 + skip + 8 * W + 2);
    while (fread(rec.data(), 1, rec.size(), f) == rec.size()) {
        for (int w = 0; w < W; ++w) { u64 v; memcpy(&v, &rec[8 + skip + 8 * w], 8); keys.push_back(v); }
        u16 c; memcpy(&c, &rec[8 + skip + 8 * W], 2);
        counts.push_back(c);
    }
    fclose(f);
}

// kmer_divide.cpp:223-228
inline u64 find_median(std::vector<u64> v)
{
    std::partial_sort(v.begin(), v.begin() + v.size() / 2 + 1, v.end());
    return v[v.size() / 2];
}

// Everything after getOccurrenceArray (kmer_divide.cpp:116-143,199-388).  occ[i][j] = occurrence of the k-mer starting at
// position j of contig i (0 when the window covers an N), j < len_i - k + 1.
inline void divide_and_write(const DivideOptions &opt, unsigned k, ContigSet &cs, const std::vector<std::vector<u64>> &occ, u64 read_length,
                             u64 max_k, bool verbose)
{
    const size_t n = cs.size();
    const double rate = opt.recalc_cov ? DBL_MAX : opt.rate;
    auto put_seq = [](std::ofstream &ofs, const std::string &codes, u64 from, u64 to) {
        u64 j = 0;
        for (u64 pos = from; pos < to; ++pos, ++j) {
            ofs.put(bin2char((unsigned char)codes[pos]));
            if ((j + 1) % 80 == 0) ofs.put('\n');
        }
        if (j % 80 != 0) ofs.put('\n');
    };
    if (!opt.mask) {
        std::vector<u64> median(n, 0);
        for (size_t i = 0; i < n; ++i) median[i] = find_median(occ[i]);
        std::vector<std::deque<u64>> bp(n);
        for (size_t i = 0; i < n; ++i) {                                  // decideContigBreakPoint
            bp[i].emplace_back(0);
            bp[i].emplace_back(cs.seq[i].size() - k + 1);
            const u64 cutoff = median[i] * rate;
            if (cutoff > 0)
                for (u64 j = 0; j < cs.seq[i].size() - k + 1; ++j)
                    if (occ[i][j] < cutoff) { bp[i].emplace_back(j); bp[i].emplace_back(j + 1); }
        }
        const std::string fn = opt.out_prefix + (opt.recalc_cov ? "_recalc.fa" : "_kmerDivided.fa");
        std::ofstream ofs(fn.c_str());
        u64 seq_id = 1;
        for (size_t c = 0; c < n; ++c) {                                  // divideAndPrintContig
            std::sort(bp[c].begin(), bp[c].end());
            for (auto it = bp[c].begin(), end = bp[c].end(); it != end; ++it) {
                auto next = it;
                ++next;
                if (next == end) break;
                if (*next - *it <= 0) continue;
                const u64 max_cut = std::max(median[c] * rate, 1.0);
                u64 hits = 0;                                              // judgeMajorityGreaterOrEqualCoverage
                bool ok = false;
                for (u64 i = *it; i < *next; ++i)
                    if (max_cut <= occ[c][i]) { ++hits; if (hits >= (*next - *it) / 2) { ok = true; break; } }
                if (!ok) continue;
                double cov = 0;                                            // calcAverageCoverage
                for (u64 i = *it; i < *next; ++i) cov += occ[c][i];
                cov /= (*next - *it);
                const u64 avg = static_cast<unsigned long>(cov + 0.5);
                const unsigned long read_cov = static_cast<u64>((avg * read_length) / (read_length - k + 1) + 0.5);
                ofs << ">seq" << seq_id << "_len" << *next - *it + k - 1 << "_cov" << read_cov << "_read" << read_length << "_maxK" << max_k << "\n";
                put_seq(ofs, cs.seq[c], *it, *next + k - 1);
            }
            ++seq_id;
        }
    } else {
        std::vector<u64> all;                                              // setMedianCoverageAll
        for (size_t i = 0; i < n; ++i) all.insert(all.end(), occ[i].begin(), occ[i].end());
        const u64 median_all = find_median(all);
        const u64 threshold = median_all * opt.mask_rate;
        if (verbose) {
            std::cerr << "masking high-coverage k-mers..." << std::endl;
            std::cerr << "OVERALL_MEDIAN_COVERAGE = " << median_all << std::endl;
            std::cerr << "COVERAGE_THRESHOLD = " << threshold << std::endl;
        }
        for (size_t i = 0; i < n; ++i)                                     // maskHighCoverageKmer
            for (u64 j = 0; j < cs.seq[i].size() - k + 1; ++j)
                if (occ[i][j] > threshold) std::fill(cs.seq[i].begin() + j, cs.seq[i].begin() + j + k, (char)4);
        std::ofstream ofs((opt.out_prefix + "_kmerMasked.fa").c_str());
        for (size_t i = 0; i < n; ++i) {                                   // printContig
            ofs << '>' << cs.name[i] << '\n';
            put_seq(ofs, cs.seq[i], 0, cs.seq[i].size());
        }
    }
    if (verbose) std::cerr << "divide completed" << std::endl;
}

} // namespace pbh
