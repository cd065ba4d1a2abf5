// Host side of the batched gap-closing graphs: what gapCloseDBG.h does per gap with
//   kmerNNCounter.countKmerForGapClose(gapInfo->seq);  kmerNNGraph.makeBruijnGraphForGapClose(kmerNNCounter, minCoverage);     (gapCloseDBG.h:91-161)
// for a whole batch of gaps with ONE device call (include/pbgpu.h pb_gap_graphs_build / _fetch), and the step that turns a gap's node list
// back into the reference's tables: the nodes come in the order of the reference's setJunctionTable / insertStraight calls, so inserting
// them in that order into tables with the reference's hashers (node_graph.hpp: Graph<W>) gives tables that iterate as the reference's do.
#pragma once
#include "node_graph.hpp"
#include <algorithm>
#include <exception>
#include <stdexcept>
#include <thread>
#include <string>
#include <vector>

extern "C" {
#include "../../include/pbgpu.h"
}

namespace pbh {

struct GapBatch {                      // input: gap g owns sequences seq_first[g] .. seq_first[g+1]-1 (the ones the reference counts: without a
    std::vector<uint64_t> seq_first{0};//        gap's first GAP_CANDIDATE_SEQ_1ST = 2 entries), sequence s = codes[seq_off[s] .. seq_off[s+1])
    std::vector<uint64_t> seq_off{0};
    std::vector<uint8_t> codes;        // 0..3, 4 = unknown
    void add_sequence(const std::vector<char> &s) { codes.insert(codes.end(), s.begin(), s.end()); seq_off.push_back(codes.size()); }
    void end_gap() { seq_first.push_back(seq_off.size() - 1); }
    uint64_t n_gaps() const { return seq_first.size() - 1; }
};

struct GapNodes {                      // output of pb_gap_graphs_fetch
    std::vector<uint64_t> node_first, word_off, words;
    std::vector<uint8_t> kind, out;
    std::vector<uint16_t> cov;
    std::vector<uint32_t> length;
};

inline GapNodes gap_graphs_build(pb_ctx *ctx, unsigned k, long min_coverage, const GapBatch &b)
{
    GapNodes r;
    uint64_t nn = 0, nw = 0;
    int rc = pb_gap_graphs_build(ctx, k, min_coverage, b.n_gaps(), b.seq_first.data(), b.seq_off.size() - 1, b.seq_off.data(), b.codes.data(), &nn, &nw);
    if (rc != PB_OK) throw std::runtime_error(std::string("libpbgpu: ") + pb_last_error(ctx));
    r.node_first.resize(b.n_gaps() + 1); r.word_off.resize(nn + 1); r.words.resize(nw ? nw : 1);
    r.kind.resize(nn ? nn : 1); r.out.resize(nn ? nn : 1); r.cov.resize(nn ? nn : 1); r.length.resize(nn ? nn : 1);
    rc = pb_gap_graphs_fetch(ctx, r.node_first.data(), r.kind.data(), r.cov.data(), r.out.data(), r.length.data(), r.word_off.data(), r.words.data());
    if (rc != PB_OK) throw std::runtime_error(std::string("libpbgpu: ") + pb_last_error(ctx));
    return r;
}

// gap g's nodes into `gr` (emptied first: graph.h:2133 deleteAllTable), as the reference's function leaves its tables
template <int W>
void gap_graph_tables(Graph<W> &gr, unsigned k, const GapNodes &r, uint64_t g)
{
    gr.set_k(k);
    gr.delete_all();
    for (uint64_t i = r.node_first[g]; i < r.node_first[g + 1]; ++i) {
        const uint64_t *w = r.words.data() + r.word_off[i];
        if (r.kind[i]) {
            Straight st;
            st.length = r.length[i];
            st.coverage = r.cov[i];
            st.out = r.out[i];
            st.seq.assign(w, w + (r.word_off[i + 1] - r.word_off[i]));
            gr.insert_straight(std::move(st));                   // graph.h:2338
        } else {
            Key<W> key;
            for (int j = 0; j < W; ++j) key.w[j] = w[j];
            Junction jn;
            jn.coverage = r.cov[i];
            jn.out = r.out[i];
            gr.junctions[key] = jn;                              // graph.h:2197 setJunctionTable
        }
    }
}

// ---- the graph part of GapCloseDBG<M, N>::gapAssemble (gapCloseDBG.h:416-436) for a batch of gaps -------------------------------------
// low k and high k: counted + built on the device for all gaps at once, branch-cut per gap on the host (the node-level steps of the assemble
// path: node_graph.hpp); mergeGraph (gapCloseDBG.h:166-258): the live straight nodes of both graphs of every gap, in the order in which the
// reference's saveLargeKmerForGapClose calls walk them, go back to the device (pb_gap_graphs_build_from_nodes), which fills the high table
// and builds the high graph again; branch cut and bubble crush per gap on the host.  low[g] / high[g] end up as kmerNNGraph of either width.
struct GapNodeBatch {
    std::vector<uint64_t> node_first{0}, word_off{0}, words;
    std::vector<uint32_t> n_bases;
    std::vector<uint16_t> coverage;
    template <int W> void add_live_straights(const Graph<W> &g, unsigned next_k)
    {
        for (auto &e : g.left) {
            const Straight &st = e.second;
            if (st.coverage == 65535 || st.length + g.k - 1 < next_k) continue;      // graph.h:2359
            n_bases.push_back((uint32_t)(st.length + g.k - 1));
            coverage.push_back(st.coverage);
            words.insert(words.end(), st.seq.begin(), st.seq.end());
            for (size_t have = st.seq.size(); have < ((size_t)n_bases.back() + 31) / 32; ++have) words.push_back(0);
            word_off.push_back(words.size());
        }
    }
    void end_gap() { node_first.push_back(n_bases.size()); }
};

inline GapNodes gap_graphs_build_from_nodes(pb_ctx *ctx, unsigned k, long min_coverage, const GapNodeBatch &b)
{
    GapNodes r;
    uint64_t nn = 0, nw = 0;
    const uint64_t n_gaps = b.node_first.size() - 1;
    int rc = pb_gap_graphs_build_from_nodes(ctx, k, min_coverage, n_gaps, b.node_first.data(), b.n_bases.size(), b.n_bases.data(), b.coverage.data(), b.word_off.data(),
                                            b.words.data(), &nn, &nw);
    if (rc != PB_OK) throw std::runtime_error(std::string("libpbgpu: ") + pb_last_error(ctx));
    r.node_first.resize(n_gaps + 1); r.word_off.resize(nn + 1); r.words.resize(nw ? nw : 1);
    r.kind.resize(nn ? nn : 1); r.out.resize(nn ? nn : 1); r.cov.resize(nn ? nn : 1); r.length.resize(nn ? nn : 1);
    rc = pb_gap_graphs_fetch(ctx, r.node_first.data(), r.kind.data(), r.cov.data(), r.out.data(), r.length.data(), r.word_off.data(), r.words.data());
    if (rc != PB_OK) throw std::runtime_error(std::string("libpbgpu: ") + pb_last_error(ctx));
    return r;
}

template <int WL, int WH>
void gap_assemble_batch(pb_ctx *ctx, unsigned k_low, unsigned k_high, long min_coverage, const GapBatch &b, std::vector<Graph<WL>> &low, std::vector<Graph<WH>> &high)
{
    const uint64_t n = b.n_gaps();
    low = std::vector<Graph<WL>>(n);
    high = std::vector<Graph<WH>>(n);
    auto cut = [](auto &g) { u64 d; do { d = g.cut_branch(); g.concatinate_nodes(); } while (d > 0); };     // gapCloseDBG.h:450-453
    // the per-gap host steps: gaps are independent of each other (their own tables), so they are spread over host threads
    auto for_gaps = [&](auto &&body) {
        const unsigned nt = (unsigned)std::min<uint64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())), std::max<uint64_t>(n / 64, 1));
        std::vector<std::exception_ptr> err(nt);
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t]() {
                try { for (uint64_t g = n * t / nt; g < n * (t + 1) / nt; ++g) body(g); }
                catch (...) { err[t] = std::current_exception(); }
            });
        for (auto &x : th) x.join();
        for (auto &e : err) if (e) std::rethrow_exception(e);
    };
    {
        const GapNodes nl = gap_graphs_build(ctx, k_low, min_coverage, b);                                 // assembleLowKmer
        for_gaps([&](uint64_t g) { low[g].set_thresholds(0.1, 0.5); gap_graph_tables<WL>(low[g], k_low, nl, g); cut(low[g]); });
    }
    {
        const GapNodes nh = gap_graphs_build(ctx, k_high, min_coverage, b);                                // assembleHighKmer
        for_gaps([&](uint64_t g) { high[g].set_thresholds(0.1, 0.5); gap_graph_tables<WH>(high[g], k_high, nh, g); cut(high[g]); });
    }
    GapNodeBatch nb;                                                                                       // mergeGraph
    for (uint64_t g = 0; g < n; ++g) { nb.add_live_straights(low[g], k_high); nb.add_live_straights(high[g], k_high); nb.end_gap(); }
    const GapNodes nm = gap_graphs_build_from_nodes(ctx, k_high, min_coverage, nb);                        // highKmerMakeBruijnGraph on the merged table
    for_gaps([&](uint64_t g) {
        gap_graph_tables<WH>(high[g], k_high, nm, g);                                                      // (the same Graph object: clear() keeps its bucket array, as the reference's does)
        cut(high[g]);
        u64 d;
        do { d = high[g].crush_bubble(65535.0); high[g].concatinate_nodes(); } while (d > 0);              // gapCloseDBG.h:431-434
    });
}

} // namespace pbh
