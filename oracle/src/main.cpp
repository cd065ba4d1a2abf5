// TEST INFRASTRUCTURE ONLY -- `pb_oracle assemble ...`: CLI of the CPU restatement, same options and
// output files as `platanus_b assemble` (assemble.cpp:55-110), uncompressed input only.
#include "assemble.hpp"
#include "fastx.hpp"
#include "../../platanus_b_amd/host/kmer_divide.hpp"
#include "../../platanus_b_amd/host/merge.hpp"
#include "merge_run.hpp"
#include "graph.hpp"
#include <unordered_map>
#include <cstdlib>
#include <cstring>
#include <iostream>

// kmer_divide with CPU look-ups (ContigDivider::getOccurrenceArray, kmer_divide.cpp:150-197, restated per position)
template <int W>
static void cpu_occurrence(unsigned k, const std::vector<pbo::u64> &keys, const std::vector<pbo::u16> &counts, const pbh::ContigSet &cs,
                           std::vector<std::vector<pbo::u64>> &occ)
{
    using namespace pbo;
    std::unordered_map<Key<W>, u16, KeyHash<W>> table;
    for (size_t i = 0; i < counts.size(); ++i) { Key<W> key; for (int w = 0; w < W; ++w) key.w[w] = keys[i * W + w]; table[key] = counts[i]; }
    occ.resize(cs.size());
    for (size_t i = 0; i < cs.size(); ++i) {
        const std::string &s = cs.seq[i];
        occ[i].assign(s.size() - k + 1, 0);
        for (u64 j = 0; j + k <= s.size(); ++j) {
            bool has_n = false;
            Kmer<W> km(k);
            for (unsigned b = 0; b < k; ++b) { const unsigned c = (unsigned char)s[j + b]; if (c > 3) { has_n = true; break; } km.push_right(c); }
            if (has_n) continue;
            auto it = table.find(km.canon());
            if (it != table.end()) occ[i][j] = it->second;
        }
    }
}

// merge.cpp:86-147 with the oracle's own table (DoubleHash) and k-mer-level DFS
template <int W>
static void oracle_merge_run(const pbh::MergeOptions &opt, const pbh::MergeContigs &mc, unsigned k, double avg_cov, pbo::u64 read_length)
{
    using namespace pbo;
    Graph<W> g;
    g.set_thresholds(atof(opt.bubble.c_str()), opt.branch);
    std::vector<ContigRecord> contigs, junctions;
    merge_contigs_into_graph<W>(g, mc, k, opt.min_cov, opt.memory_gb * 1000000000ull, avg_cov, 1.0, true, contigs, junctions);
    print_contigs(opt.out_prefix + "_merged.fa", contigs, 1.0, (double)read_length, (long)k, "seq", false);
    print_contigs(opt.out_prefix + "_mergedJunctionKmer.fa", junctions, 1.0, (double)read_length, (long)k, "junction", false);
}

static int oracle_merge(int argc, char **argv)
{
    pbh::MergeOptions o;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-o") o.out_prefix = val();
        else if (a == "-c") o.min_cov = atoi(val().c_str());
        else if (a == "-k") o.k_factor = atof(val().c_str());
        else if (a == "-l") o.len_factor = atof(val().c_str());
        else if (a == "-u") o.bubble = val();
        else if (a == "-d") o.branch = atof(val().c_str());
        else if (a == "-m") o.memory_gb = atoi(val().c_str());
        else if (a == "-tmp") val();
        else if (a == "-f") { while (i + 1 < argc && argv[i + 1][0] != '-') o.files.push_back(argv[++i]); }
        else return 1;
    }
    const pbo::u64 read_length = pbh::first_header_read_length(o.files.at(0));
    const unsigned k = (unsigned)(o.k_factor * read_length + 0.5);
    const pbo::u64 min_len = (pbo::u64)(o.len_factor * read_length + 0.5);
    pbh::MergeContigs mc;
    for (const std::string &f : o.files) pbh::read_contigs_cut_n(f, min_len, mc);
    const double avg_cov = pbh::merge_average_coverage(mc);
    switch (pbh::words_for_k(k)) {
    case 1: oracle_merge_run<1>(o, mc, k, avg_cov, read_length); break;
    case 2: oracle_merge_run<2>(o, mc, k, avg_cov, read_length); break;
    case 3: oracle_merge_run<3>(o, mc, k, avg_cov, read_length); break;
    case 4: oracle_merge_run<4>(o, mc, k, avg_cov, read_length); break;
    default: oracle_merge_run<5>(o, mc, k, avg_cov, read_length); break;
    }
    return 0;
}

static int oracle_kmer_divide(int argc, char **argv)
{
    pbh::DivideOptions o;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "-o") o.out_prefix = val();
        else if (a == "-k") o.occ_file = val();
        else if (a == "-r") o.rate = atof(val().c_str());
        else if (a == "-R") o.mask_rate = atof(val().c_str());
        else if (a == "-t" || a == "-tmp") val();
        else if (a == "-recalc_cov") o.recalc_cov = true;
        else if (a == "-mask") o.mask = true;
        else if (a == "-f") { while (i + 1 < argc && argv[i + 1][0] != '-') o.files.push_back(argv[++i]); }
        else return 1;
    }
    unsigned k = 0;
    std::vector<pbo::u64> keys;
    std::vector<pbo::u16> counts;
    pbh::read_kmer_occ(o.occ_file, k, keys, counts);
    pbh::ContigSet cs;
    pbo::u64 read_length = 100, max_k = 100;
    for (size_t i = 0; i < o.files.size(); ++i) pbh::read_contig_fasta(o.files[i], cs, i == 0, read_length, max_k);
    std::vector<std::vector<pbo::u64>> occ;
    switch (pbh::words_for_k(k)) {
    case 1: cpu_occurrence<1>(k, keys, counts, cs, occ); break;
    case 2: cpu_occurrence<2>(k, keys, counts, cs, occ); break;
    case 3: cpu_occurrence<3>(k, keys, counts, cs, occ); break;
    case 4: cpu_occurrence<4>(k, keys, counts, cs, occ); break;
    default: cpu_occurrence<5>(k, keys, counts, cs, occ); break;
    }
    pbh::divide_and_write(o, k, cs, occ, read_length, max_k, true);
    return 0;
}

int main(int argc, char **argv)
{
    using namespace pbo;
    if (argc >= 2 && strcmp(argv[1], "merge") == 0) {
        try { return oracle_merge(argc, argv); }
        catch (const pbh::RefError &e) { std::cerr << "Error(" << e.id << "): " << e.what() << std::endl; return e.id; }
        catch (const std::exception &e) { std::cerr << "Error: " << e.what() << std::endl; return 2; }
    }
    if (argc >= 2 && strcmp(argv[1], "kmer_divide") == 0) {
        try { return oracle_kmer_divide(argc, argv); }
        catch (const pbh::RefError &e) { std::cerr << "Error(" << e.id << "): " << e.what() << std::endl; return e.id; }
        catch (const std::exception &e) { std::cerr << "Error: " << e.what() << std::endl; return 2; }
    }
    if (argc < 2 || strcmp(argv[1], "assemble") != 0) {
        std::cerr << "usage: pb_oracle assemble [platanus_b assemble options]" << std::endl;
        return 1;
    }
    AssembleOptions o;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) { std::cerr << "missing value for " << a << std::endl; exit(1); } return argv[++i]; };
        if (a == "-o") o.out_prefix = val();
        else if (a == "-k") o.k0 = atoi(val().c_str());
        else if (a == "-K") o.max_k_ratio = atof(val().c_str());
        else if (a == "-s") o.step = atoi(val().c_str());
        else if (a == "-n") o.n_cutoff = atoi(val().c_str());
        else if (a == "-c") o.min_cov = atoi(val().c_str());
        else if (a == "-a") o.safety = atof(val().c_str());
        else if (a == "-u") o.bubble = val();
        else if (a == "-d") o.branch = atof(val().c_str());
        else if (a == "-e") o.expected_cov = val();
        else if (a == "-t") val();                       // single-thread schedule only
        else if (a == "-m") o.memory_gb = atoi(val().c_str());
        else if (a == "-tmp") val();
        else if (a == "-kmer_occ_only") o.kmer_occ_only = true;
        else if (a == "-repeat") o.repeat = true;
        else if (a == "-f") { while (i + 1 < argc && argv[i + 1][0] != '-') o.files.push_back(argv[++i]); }
        else { std::cerr << "unknown option " << a << std::endl; return 1; }
    }
    try {
        ReadSet reads;
        for (const std::string &f : o.files) read_fastx(f, reads);
        return run_assemble(o, reads);
    } catch (const pbh::RefError &e) {           // main.cpp:118-127: message, exit status = error id
        std::cerr << "Error(" << e.id << "): " << e.what() << std::endl;
        return e.id;
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 2;
    }
}
