// TEST INFRASTRUCTURE ONLY -- the graph path shared by `merge` (merge.cpp:106-147) and by assemble's -repeat mergeContig
// (assemble.cpp:270-295): k-mers of contig pieces with value max(coverage, min_cov) (Counter::makeKmerReadDistributionFromContig,
// counter.h:510-593), sorted keys, table, makeInitialBruijnGraph, cutBranchIterative, crushBubbleIterative, saveContigSimple /
// saveJunction.  The graph object is the caller's: the reference reuses its per-width BruijnGraph, whose unordered_maps keep
// their bucket counts, and the container order decides the output order.
#pragma once
#include "assemble.hpp"
#include "graph.hpp"
#include "../../platanus_b_amd/host/merge.hpp"

namespace pbo {

template <int W>
inline void merge_contigs_into_graph(Graph<W> &g, const pbh::MergeContigs &mc, unsigned k, unsigned min_cov, u64 memory, double avg_cov, double ratio,
                                     bool verbose, std::vector<ContigRecord> &contigs, std::vector<ContigRecord> &junctions)
{
    Counter<W> c;
    c.k = k;
    u64 total = 0;
    for (const std::string &s : mc.seq) if (s.size() >= k) total += s.size() - k + 1;
    const u64 upper = std::max(table_size_for_keys(total), table_size_for_memory(memory, W));
    c.table.resize(upper);
    for (size_t i = 0; i < mc.seq.size(); ++i) {                                    // counter.h:545-578
        const u64 v = std::max<u64>(mc.coverage[i], min_cov);
        for_each_kmer<W>((const uint8_t *)mc.seq[i].data(), (unsigned)mc.seq[i].size(), k, [&](const Key<W> &key) {
            const u64 s = c.table.find_any(key);
            c.table.keys[s] = key;
            if (c.table.vals[s] < v) c.table.vals[s] = (u16)v;
        });
    }
    c.occ_hist.assign(65535, 0);
    for (u64 s = 0; s < c.table.size(); ++s)                                        // writeKmerDistribution (counter.h:783-805)
        if (c.table.vals[s] != 0) { c.kmer_keys.push_back(c.table.keys[s]); c.kmer_vals.push_back(c.table.vals[s]); ++c.occ_hist[c.table.vals[s]]; c.table.vals[s] = 0; }
    c.find_max_occ();
    std::vector<Key<W>> sorted = c.sorted_keys(0);
    c.load_kmers(0, upper);
    g.set_k(k);
    g.make_initial(c.table, sorted);
    c.table.release();
    g.cut_branch_iterative(verbose);
    g.crush_bubble_iterative(avg_cov, verbose);
    g.save_contig_simple(contigs, ratio);
    g.save_junction(junctions, ratio);
}

} // namespace pbo
