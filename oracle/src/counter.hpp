// TEST INFRASTRUCTURE ONLY -- oracle restatement of the reference's Counter<KMER> (counter.h:36-1056)
// for the single-thread (-t 1) schedule, which is the one whose results are fully deterministic
// (SURVEY section 4).  Temp files of the reference become in-memory vectors kept in file order.
#pragma once
#include "dhash.hpp"
#include "../../platanus_b_amd/host/errors.hpp"
#include <algorithm>
#include <cstdio>
#include <string>
#include <vector>

namespace pbo {

static const unsigned HASH_DIVID = 1024;          // counter.h:210-212
static const u64 MAX_READ_LEN = 500000;           // common.cpp:28
static const double MAX_LOAD_FACTOR = 0.9;        // common.cpp:34

// reads in the order the reference's temp file 0 holds them with -t 1 (assemble.cpp:790-986):
// all reads of -f file 1, then file 2, ...   codes: 0..3, 4 = N
struct ReadSet {
    std::vector<uint8_t> bases;
    std::vector<u64> off{0};
    size_t size() const { return off.size() - 1; }
    const uint8_t *read(size_t i) const { return bases.data() + off[i]; }
    unsigned len(size_t i) const { return (unsigned)(off[i + 1] - off[i]); }
    void add(const uint8_t *b, size_t n) { bases.insert(bases.end(), b, b + n); off.push_back(bases.size()); }
};

// calls f(canonical key) for every N-free k-mer window of the read, left to right (counter.h:405-431)
template <int W, typename F>
inline void for_each_kmer(const uint8_t *b, unsigned len, unsigned k, F &&f)
{
    if (len < k) return;
    Kmer<W> km(k);
    bool have_n = false;
    unsigned n_pos = 0;
    for (unsigned i = 0; i < k - 1; ++i) {
        const unsigned c = b[i];
        if (c > 3) { have_n = true; n_pos = i; }
        km.push_right(c & 3);
    }
    for (unsigned i = 0; i + k <= len; ++i) {
        const unsigned c = b[i + k - 1];
        if (c > 3) { have_n = true; n_pos = i + k - 1; }
        km.push_right(c & 3);
        if (have_n && n_pos >= i) continue;       // window [i, i+k) covers an unknown base
        f(km.canon());
    }
}

template <int W>
struct Counter {
    unsigned k = 0;
    std::vector<u64> length_hist;                 // lengthDistribution[MAX_READ_LEN + 1]
    std::vector<u64> occ_hist;                    // occurrenceDistribution[65535]
    u64 max_occ = 0;
    DHash<W> table;                               // occurrenceTable
    // kmerFP: (key, count) records in the order the reference appends them
    std::vector<Key<W>> kmer_keys;
    std::vector<u16> kmer_vals;

    void clear_kmer_file() { kmer_keys.clear(); kmer_vals.clear(); }

    // counter.h:458-476 on one of the 1024 stripe tables
    static inline bool count_or_spill(DHash<W> &t, const Key<W> &key, unsigned iterate_times)
    {
        const u64 s = t.find_times_any(key, iterate_times);
        if (t.vals[s] == 0) { t.keys[s] = key; t.vals[s] = 1; return true; }
        if (t.keys[s] == key) { if (t.vals[s] != 65534) ++t.vals[s]; return true; }
        return false;
    }

    // counter.h:482-507 (stripe form) -- returns the iterateTimes increment
    unsigned dump_stripes(std::vector<DHash<W>> &stripes)
    {
        u64 num = 0;
        for (unsigned h = 0; h < HASH_DIVID; ++h) {
            DHash<W> &t = stripes[h];
            for (u64 s = 0, n = t.size(); s < n; ++s) {
                if (t.vals[s] != 0) {
                    ++num;
                    kmer_keys.push_back(t.keys[s]);
                    kmer_vals.push_back(t.vals[s]);
                    ++occ_hist[t.vals[s]];
                    t.vals[s] = 0;
                }
            }
        }
        return (num / HASH_DIVID < stripes[0].size() * MAX_LOAD_FACTOR / 2) ? 1 : 0;
    }

    void find_max_occ()
    {
        for (unsigned i = 65534; i > 0; --i) if (occ_hist[i] > 0) { max_occ = i; break; }
    }

    // counter.h:275-383 makeKmerReadDistributionMT with numThread == 1; returns doubleHashSize
    u64 count_reads(unsigned k_, const ReadSet &reads, u64 memory)
    {
        k = k_;
        clear_kmer_file();
        const u64 dh_size = table_size_for_memory(memory, W);
        std::vector<DHash<W>> stripes(HASH_DIVID);
        for (auto &t : stripes) t.resize(dh_size / HASH_DIVID);

        length_hist.assign(MAX_READ_LEN + 1, 0);
        std::vector<Key<W>> spill, next_spill;
        for (size_t r = 0; r < reads.size(); ++r) {
            const unsigned len = reads.len(r);
            ++length_hist[len];
            for_each_kmer<W>(reads.read(r), len, k, [&](const Key<W> &key) {
                if (!count_or_spill(stripes[key.w[0] & (HASH_DIVID - 1)], key, 32)) spill.push_back(key);
            });
        }
        occ_hist.assign(65535, 0);
        unsigned iterate_times = 32;
        while (true) {
            iterate_times += dump_stripes(stripes);
            if (spill.empty()) break;
            next_spill.clear();
            for (const Key<W> &key : spill)
                if (!count_or_spill(stripes[key.w[0] & (HASH_DIVID - 1)], key, iterate_times)) next_spill.push_back(key);
            spill.swap(next_spill);
        }
        max_occ = 0;
        find_max_occ();
        return dh_size;
    }

    // counter.h:244-267
    u64 left_local_minimal_value(u64 window) const
    {
        if (max_occ <= window) return 0;
        const u64 n = max_occ - window + 2;
        std::vector<u64> w(n, 0);
        u64 i;
        for (i = 0; i < window; ++i) w[1] += occ_hist[1 + i];
        for (i = 2; i < n; ++i) {
            w[i] = w[i - 1] - occ_hist[i - 1] + occ_hist[i + window - 1];
            if (w[i] >= w[i - 1]) break;
        }
        return i <= max_occ ? (i - 1 + window / 2) : (1 + window / 2);
    }
    // counter.h:220-238
    static double dist_average(const std::vector<u64> &d, u64 start, u64 end)
    {
        if (end > d.size() || start > end) throw pbh::kmer_dist_error();       // counter.h:226
        u64 sum = 0, num = 0;
        for (u64 i = start; i <= end; ++i) { sum += i * d[i]; num += d[i]; }
        if (num == 0) throw pbh::kmer_dist_error();                            // counter.h:236
        return (double)sum / num;
    }
    double occ_average(u64 s, u64 e) const { return dist_average(occ_hist, s, e); }
    double length_average(u64 s, u64 e) const { return dist_average(length_hist, s, e); }

    // counter.h:999-1007
    void write_occ_distribution(const std::string &path) const
    {
        FILE *f = fopen(path.c_str(), "w");
        if (!f) throw std::runtime_error("cannot open " + path);
        for (u64 i = 1; i <= max_occ; ++i) fprintf(f, "%llu\t%llu\n", (unsigned long long)i, (unsigned long long)occ_hist[i]);
        fclose(f);
    }

    // counter.h:916-951: keys with count >= cutoff, ascending
    std::vector<Key<W>> sorted_keys(u64 min_occ)
    {
        std::vector<Key<W>> out;
        for (size_t i = 0; i < kmer_keys.size(); ++i) if (kmer_vals[i] >= min_occ) out.push_back(kmer_keys[i]);
        std::sort(out.begin(), out.end());
        std::vector<u64>().swap(occ_hist);
        return out;
    }

    // counter.h:599-640; returns `size` (need), table gets max(size, dh_size) slots
    u64 load_kmers(u64 min_occ, u64 dh_size)
    {
        u64 total = 0;
        for (size_t i = 0; i < kmer_vals.size(); ++i) if (kmer_vals[i] >= min_occ) ++total;
        const u64 size = table_size_for_keys(total);
        table.resize(std::max(size, dh_size));
        for (size_t i = 0; i < kmer_keys.size(); ++i) if (kmer_vals[i] >= min_occ) table.put(kmer_keys[i], kmer_vals[i]);
        clear_kmer_file();
        return size;
    }

    // counter.h:954-963 + doubleHash.h:266-278.  For W > 1 the reference dumps the raw C++ object
    // (vptr, heap pointer, len, words); the two pointers are not reproducible even by the reference
    // itself, they are written as zero here.
    void write_table_binary(const std::string &path) const
    {
        FILE *f = fopen(path.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot open " + path);
        u64 kk = k, m = table.nmask;
        fwrite(&kk, 8, 1, f);
        fwrite(&m, 8, 1, f);
        for (u64 s = 0; s <= table.nmask; ++s) {
            if (!table.vals[s]) continue;
            fwrite(&s, 8, 1, f);
            if (W > 1) { u64 z[3] = {0, 0, kk}; fwrite(z, 8, 3, f); }
            fwrite(table.keys[s].w, 8, W, f);
            fwrite(&table.vals[s], 2, 1, f);
        }
        fclose(f);
    }

    // ---- second form: the single big table already holds contig-derived k-mers (counter.h:662-861) ----
    // reads' k-mers absent from the table are spilled, the table is dumped + cleared, the spilled keys are
    // then counted into the same table with bounded probing.
    void count_reads_considering_table(unsigned k_, const ReadSet &reads)
    {
        k = k_;
        clear_kmer_file();
        std::vector<Key<W>> spill, next_spill;
        for (size_t r = 0; r < reads.size(); ++r) {
            for_each_kmer<W>(reads.read(r), reads.len(r), k, [&](const Key<W> &key) {
                if (table.value(key) == 0) spill.push_back(key);
            });
        }
        occ_hist.assign(65535, 0);
        for (u64 s = 0, n = table.size(); s < n; ++s) {
            if (table.vals[s] != 0) {
                kmer_keys.push_back(table.keys[s]);
                kmer_vals.push_back(table.vals[s]);
                ++occ_hist[table.vals[s]];
                table.vals[s] = 0;
            }
        }
        bool first = true;
        unsigned iterate_times = 32;
        while (true) {
            if (!first) {
                u64 num = 0;
                for (u64 s = 0, n = table.size(); s < n; ++s) {
                    if (table.vals[s] != 0) {
                        ++num;
                        kmer_keys.push_back(table.keys[s]);
                        kmer_vals.push_back(table.vals[s]);
                        ++occ_hist[table.vals[s]];
                        table.vals[s] = 0;
                    }
                }
                if (num / HASH_DIVID < table.size() * MAX_LOAD_FACTOR / 2) ++iterate_times;
            }
            first = false;
            if (spill.empty()) break;
            next_spill.clear();
            for (const Key<W> &key : spill)
                if (!count_or_spill(table, key, iterate_times)) next_spill.push_back(key);
            spill.swap(next_spill);
        }
        max_occ = 0;
        find_max_occ();
    }

    // counter.h:869-910: keep a read iff one of its N-free k-mers is in the table
    ReadSet pickup_reads_matching_table(const ReadSet &reads) const
    {
        ReadSet out;
        for (size_t r = 0; r < reads.size(); ++r) {
            bool hit = false;
            const uint8_t *b = reads.read(r);
            const unsigned len = reads.len(r);
            if (len < k) continue;
            Kmer<W> km(k);
            bool have_n = false; unsigned n_pos = 0;
            for (unsigned i = 0; i < k - 1; ++i) { if (b[i] > 3) { have_n = true; n_pos = i; } km.push_right(b[i] & 3); }
            for (unsigned i = 0; i + k <= len && !hit; ++i) {
                const unsigned c = b[i + k - 1];
                if (c > 3) { have_n = true; n_pos = i + k - 1; }
                km.push_right(c & 3);
                if (have_n && n_pos >= i) continue;
                if (table.value(km.canon()) > 0) hit = true;
            }
            if (hit) out.add(b, len);
        }
        return out;
    }
};

} // namespace pbo
