// Host side of the batched gap-closing graphs: what gapCloseDBG.h does per gap with
//   kmerNNCounter.countKmerForGapClose(gapInfo->seq);  kmerNNGraph.makeBruijnGraphForGapClose(kmerNNCounter, minCoverage);     (gapCloseDBG.h:91-161)
// for a whole batch of gaps with ONE device call (include/pbgpu.h pb_gap_graphs_build / _fetch), and the step that turns a gap's node list
// back into the reference's tables: the nodes come in the order of the reference's setJunctionTable / insertStraight calls, so inserting
// them in that order into tables with the reference's hashers (node_graph.hpp: Graph<W>) gives tables that iterate as the reference's do.
#pragma once
#include "node_graph.hpp"
#include <stdexcept>
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

} // namespace pbh
