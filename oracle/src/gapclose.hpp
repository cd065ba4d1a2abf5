// TEST INFRASTRUCTURE ONLY -- CPU restatement of the local de Bruijn graphs of the reference's gap closing:
//   Counter<KMER>::countKmerForGapClose        counter.h:1015-1056   every N-free window's FORWARD k-mer of a gap's candidate sequences
//   BruijnGraph<KMER>::makeBruijnGraphForGapClose  graph.h:2116-2345 strand-specific nodes over the k-mers with value >= minCoverage
// What the reference's loop does, read closely: `kmerStack` is declared outside the loop over the sorted keys and every iteration pushes ONE key
// and works on the stack's top ONCE (`if (kmerStack.size() > 0)`, graph.h:2143 -- not `while`), so the neighbours it pushes are never looked at
// again: the function is a plain loop over the keys in ascending order, each unvisited key with value >= minCoverage starting one node.
// Nodes are reported in the order in which the reference puts them into its tables (setJunctionTable / insertStraight); with that order and the
// reference's hashers the product's Graph<W> reproduces the tables' iteration order (tests: golden dumps of the reference's own tables).
#pragma once
#include "../../platanus_b_amd/host/node_graph.hpp"
#include <algorithm>
#include <unordered_map>
#include <vector>

namespace pbo {
using namespace pbh;

template <int W>
struct GapNode {
    bool straight = false;
    Key<W> key;                      // junction: its k-mer
    Straight st;                     // straight: length, coverage, out, seq (graph.h:2320-2338)
    Junction jn;
};

template <int W>
struct GapGraph {
    typedef Key<W> K;
    unsigned k = 0;
    std::unordered_map<K, u16, KeyHash<W>> table;     // occurrenceGapTable

    // counter.h:1015-1056; `seqs`: the sequences that are counted (the caller leaves out the gap's first GAP_CANDIDATE_SEQ_1ST = 2 entries)
    void count(const std::vector<std::vector<unsigned char>> &seqs)
    {
        for (const auto &s : seqs) {
            if (s.size() < k) continue;
            K f;
            unsigned run = 0;                      // bases since the last unknown one
            for (size_t i = 0; i < s.size(); ++i) {
                if (s[i] >= 4) { run = 0; continue; }
                f.shl2(k);
                f.set(0, s[i]);
                if (++run >= k) { u16 &v = table[f]; if (v < 65534) ++v; }       // counter.h:1051-1052
            }
        }
    }
    u16 val(const K &key) const { auto it = table.find(key); return it == table.end() ? (u16)0 : it->second; }
    u16 overwrite(const K &key) { auto it = table.find(key); const u16 v = it->second; it->second = 65535; return v; }

    // graph.h:2116-2345
    std::vector<GapNode<W>> make_graph(long min_cov)
    {
        std::vector<GapNode<W>> out;
        std::vector<K> sorter;
        for (auto &e : table) if (e.second != 0) sorter.push_back(e.first);       // counter.h:168-175
        std::sort(sorter.begin(), sorter.end());
        auto present = [&](const K &x) { return (long)val(x) >= min_cov; };
        std::vector<unsigned char> lseq, rseq;
        for (const K &start : sorter) {
            if ((long)val(start) < min_cov) continue;
            const u16 v0 = overwrite(start);
            if (v0 == 65535) continue;
            u64 sum = v0;
            unsigned lfl = 0, rfl = 0;
            K lk = start; lk.shr2();
            for (unsigned b = 0; b < 4; ++b) { lk.set(k - 1, b); if (present(lk)) lfl |= 1u << b; }
            K rk = start; rk.shl2(k);
            for (unsigned b = 0; b < 4; ++b) { rk.set(0, b); if (present(rk)) rfl |= 1u << b; }
            lseq.clear(); rseq.clear();
            if (lfl) lseq.push_back((unsigned char)flag2base(lfl));
            if (rfl) rseq.push_back((unsigned char)flag2base(rfl));
            if ((!lseq.empty() && lseq[0] == 4) || (!rseq.empty() && rseq[0] == 4)) {
                GapNode<W> n;
                n.key = start;
                n.jn.coverage = (u16)sum;
                n.jn.out = (uint8_t)((lfl << 4) | rfl);
                out.push_back(n);
                continue;
            }
            if (!rseq.empty()) {                                   // graph.h:2203-2258
                K key = start; key.shl2(k); key.set(0, rseq[0]);
                while (true) {
                    K p = key; p.shr2();
                    unsigned in = 0;
                    for (unsigned b = 0; b < 4; ++b) { p.set(k - 1, b); if (present(p)) in |= 1u << b; }
                    if (flag_count(in) > 1 || val(key) == 65535) { rfl = 1u << rseq.back(); rseq.pop_back(); break; }
                    K s = key; s.shl2(k);
                    K next;
                    rfl = 0;
                    for (unsigned b = 0; b < 4; ++b) { s.set(0, b); if (present(s)) { rfl |= 1u << b; next = s; } }
                    if (rfl == 0) { sum += overwrite(key); break; }
                    else if (flag_count(rfl) > 1) { rfl = 1u << rseq.back(); rseq.pop_back(); break; }
                    rseq.push_back((unsigned char)flag2base(rfl));
                    sum += overwrite(key);
                    key = next;
                }
            }
            if (!lseq.empty()) {                                   // graph.h:2261-2309
                K key = start; key.shr2(); key.set(k - 1, lseq[0]);
                while (true) {
                    K s = key; s.shl2(k);
                    unsigned outd = 0;
                    for (unsigned b = 0; b < 4; ++b) { s.set(0, b); if (present(s)) outd |= 1u << b; }
                    if (flag_count(outd) > 1 || val(key) == 65535) { lfl = 1u << lseq.back(); lseq.pop_back(); break; }
                    K p = key; p.shr2();
                    K next;
                    lfl = 0;
                    for (unsigned b = 0; b < 4; ++b) { p.set(k - 1, b); if (present(p)) { lfl |= 1u << b; next = p; } }
                    if (lfl == 0) { sum += overwrite(key); break; }
                    else if (flag_count(lfl) > 1) { lfl = 1u << lseq.back(); lseq.pop_back(); break; }
                    lseq.push_back((unsigned char)flag2base(lfl));
                    sum += overwrite(key);
                    key = next;
                }
            }
            GapNode<W> n;                                          // graph.h:2320-2338
            n.straight = true;
            Straight &st = n.st;
            st.length = lseq.size() + rseq.size() + 1;
            st.seq.assign((st.length + k + 31) / 32, 0);
            const u64 rn = rseq.size();
            for (u64 j = 0; j < rn; ++j) st.set(rn - 1 - j, rseq[j]);
            for (unsigned j = 0; j < k; ++j) st.set(rn + j, start.get(j));
            for (u64 j = 0; j < lseq.size(); ++j) st.set(rn + k + j, lseq[j]);
            st.coverage = (u16)(static_cast<double>(sum) / static_cast<double>(st.length) + 0.5);
            st.out = (uint8_t)((lfl << 4) | rfl);
            out.push_back(n);
        }
        return out;
    }
};

// BruijnGraph<KMER>::saveLargeKmerForGapClose (graph.h:2347-2370): the next-k-mers of every live straight node of `g` (walked in its table's
// order), each with the node's coverage, overwriting (counter.h:194)
template <int W, int WN>
void save_large_kmer(Graph<W> &g, GapGraph<WN> &next)
{
    const unsigned nk = next.k;
    for (auto &e : g.left) {
        const Straight &st = e.second;
        if (st.coverage == 65535 || st.length + g.k - 1 < nk) continue;
        for (u64 i = 0; i + nk <= st.length + g.k - 1; ++i) {
            Key<WN> key;
            for (unsigned j = 0; j < nk; ++j) key.set(j, st.get(i + j));
            next.table[key] = st.coverage;
        }
    }
}

// GapCloseDBG<M, N>::gapAssemble (gapCloseDBG.h:416-436) up to its last graph: low k and high k each counted, built and branch-cut
// (assembleLowKmer / assembleHighKmer, :444-470), the high table refilled from both graphs (mergeGraph, :166-258), the high graph built again
// from it, branch-cut and bubble-crushed.  `low` / `high` are what kmerNNGraph of either width would hold afterwards.
template <int WL, int WH>
void gap_assemble(const std::vector<std::vector<unsigned char>> &seqs, unsigned k_low, unsigned k_high, long min_cov, Graph<WL> &low, Graph<WH> &high)
{
    auto build = [&](auto &graph, auto &gg) {
        const auto nodes = gg.make_graph(min_cov);
        graph.delete_all();
        for (const auto &n : nodes) { if (n.straight) graph.insert_straight(n.st); else graph.junctions[n.key] = n.jn; }
    };
    auto cut = [](auto &graph) { u64 n; do { n = graph.cut_branch(); graph.concatinate_nodes(); } while (n > 0); };
    low.set_k(k_low); low.set_thresholds(0.1, 0.5);
    high.set_k(k_high); high.set_thresholds(0.1, 0.5);
    GapGraph<WL> cl; cl.k = k_low; cl.count(seqs);
    build(low, cl); cut(low);
    GapGraph<WH> ch; ch.k = k_high; ch.count(seqs);
    build(high, ch); cut(high);
    ch.table.clear();
    save_large_kmer(low, ch);
    save_large_kmer(high, ch);
    build(high, ch); cut(high);
    { u64 n; do { n = high.crush_bubble(65535.0); high.concatinate_nodes(); } while (n > 0); }
}

} // namespace pbo
