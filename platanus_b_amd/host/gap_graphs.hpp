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

// ---- what a gap's final graph says about the gap: GapCloseDBG<M, N>::closedGap / closedGapPartial / extendEdgeEntity ------------------------
// (gapCloseDBG.h:479-741).  The flanks are GAP::headSeq (the bases in front of the gap, left to right) and GAP::tailSeq (behind it), codes
// 0..3 / 4; a straight node's base p counts from its 3' end (graph.h:67-71), so its left-to-right base j is get(contig_length - 1 - j).
namespace pbh {

struct GapFlanks {
    std::vector<char> head, tail;      // GAP::headSeq, GAP::tailSeq
    int start = 0, end = 0;            // GAP::start, GAP::end (closedGapPartial's remaining length; extendEdge's side)
};
struct GapClosure {                    // GapCloseDBG::seq, seqLength, remainedGap
    std::vector<char> seq;
    int seq_length = 0, remained_gap = 0;
};

struct GapCloser {
    unsigned max_k = 0, min_overlap = 32;      // maxKmer; -vd (gapClose.cpp:57)
    double max_miss_rate = 0.01;               // -ed (gapClose.cpp:60)
    const GapFlanks *gap = nullptr;

    // gapCloseDBG.h:599-635: the longest suffix-of-head / prefix-of-node overlap within the error rate, and its error rate
    std::pair<long, double> mismatch_left(const Straight &st) const
    {
        long overlap = 0;
        double miss_rate = 1.0;
        const long contig_length = (long)(st.length + max_k - 1);
        const unsigned long long head_size = gap->head.size();
        for (long length = contig_length; length >= (long)min_overlap; --length) {
            double tol = length * max_miss_rate + 0.5;
            unsigned long long miss = 0;
            long i = 0;
            for (; i < (long)min_overlap; ++i)
                if (gap->head[head_size - 1 - i] != (char)st.get((u64)(contig_length - length + i))) { ++miss; if (miss > tol) break; }
            if (i < (long)min_overlap) continue;
            const long max_overlap = (long)std::min<unsigned long long>((unsigned long long)length, head_size);
            tol = max_overlap * max_miss_rate + 0.5;
            for (; i < max_overlap; ++i)
                if (gap->head[head_size - 1 - i] != (char)st.get((u64)(contig_length - length + i))) { ++miss; if (miss > tol) break; }
            if (miss <= tol && static_cast<double>(miss) / i < miss_rate) { miss_rate = static_cast<double>(miss) / i; overlap = length; }
        }
        return {overlap, miss_rate};
    }
    // gapCloseDBG.h:640-676
    std::pair<long, double> mismatch_right(const Straight &st) const
    {
        long overlap = 0;
        double miss_rate = 1.0;
        const long contig_length = (long)(st.length + max_k - 1);
        for (long length = contig_length; length >= (long)min_overlap; --length) {
            double tol = length * max_miss_rate + 0.5;
            unsigned long long miss = 0;
            long i = 0;
            for (; i < (long)min_overlap; ++i)
                if (gap->tail[i] != (char)st.get((u64)(length - 1 - i))) { ++miss; if (miss > tol) break; }
            if (i < (long)min_overlap) continue;
            const long max_overlap = (long)std::min<unsigned long long>((unsigned long long)length, (unsigned long long)gap->tail.size());
            tol = max_overlap * max_miss_rate + 0.5;
            for (; i < max_overlap; ++i)
                if (gap->tail[i] != (char)st.get((u64)(length - 1 - i))) { ++miss; if (miss > tol) break; }
            if (miss <= tol && static_cast<double>(miss) / i < miss_rate) { miss_rate = static_cast<double>(miss) / i; overlap = length; }
        }
        return {overlap, miss_rate};
    }
    bool enabled(size_t n_seqs) const { return gap->head.size() >= min_overlap && gap->tail.size() >= min_overlap && n_seqs >= 2; }     // gapCloseDBG.h:405-408

    // gapCloseDBG.h:479-532: exactly one live node overlaps both flanks
    template <int W> bool closed(Graph<W> &g, GapClosure &r) const
    {
        long max_left = 0, max_right = 0;
        double min_rate = 1.0;
        const Straight *best = nullptr;
        for (auto &e : g.left) {
            const Straight &st = e.second;
            if (st.coverage == 65535) continue;
            const auto left = mismatch_left(st);
            if (left.first == 0) continue;
            const auto right = mismatch_right(st);
            if (right.first == 0) continue;
            if (best != nullptr) return false;
            const double rate = (right.second * right.first + left.second * left.first) / (right.first + left.first);
            if (rate < min_rate) { max_left = left.first; max_right = right.first; min_rate = rate; best = &st; }
        }
        if (max_left == 0) return false;
        const long contig_length = (long)(best->length + max_k - 1);
        r.seq_length = (int)(contig_length - max_left - max_right);
        if (r.seq_length > 0) {
            for (int i = 0; i < r.seq_length; ++i) r.seq.emplace_back((char)best->get((u64)(contig_length - 1 - max_left - i)));
        } else {
            const unsigned long long t = (unsigned long long)(-r.seq_length);
            if (t > gap->head.size() || t > gap->tail.size()) return false;
            for (int i = 0; i < -r.seq_length; ++i)
                if (gap->head[gap->head.size() - 1 - i] != gap->tail[-r.seq_length - 1 - i]) return false;
            --r.seq_length;
        }
        return true;
    }
    // gapCloseDBG.h:535-596: the best node on either side, unknown bases between them
    template <int W> bool closed_partial(Graph<W> &g, GapClosure &r) const
    {
        const int MIN_GAP_LENGTH = 10;
        long max_left = 0, max_right = 0;
        double min_left_rate = 1.0, min_right_rate = 1.0;
        const Straight *ls = nullptr, *rs = nullptr;
        for (auto &e : g.left) {
            const Straight &st = e.second;
            if (st.coverage == 65535) continue;
            const auto left = mismatch_left(st);
            if (left.first > 0 && (left.second < min_left_rate || (left.second < min_left_rate && left.first > max_left))) {
                max_left = left.first; min_left_rate = left.second; ls = &st;
                continue;
            }
            const auto right = mismatch_right(st);
            if (right.first > 0 && (right.second < min_right_rate || (right.second < min_right_rate && right.first > max_right))) {
                max_right = right.first; min_right_rate = right.second; rs = &st;
            }
        }
        if (max_left == 0 && max_right == 0) return false;
        const int lcl = ls != nullptr ? (int)(ls->length + max_k - 1) : 0, rcl = rs != nullptr ? (int)(rs->length + max_k - 1) : 0;
        r.seq_length = (int)((lcl - max_left) + (rcl - max_right));
        int gap_length = MIN_GAP_LENGTH;
        if ((gap->end - gap->start) - r.seq_length > MIN_GAP_LENGTH) gap_length = (gap->end - gap->start) - r.seq_length;
        r.seq_length += gap_length;
        r.remained_gap = gap_length;
        for (int i = 0; i < lcl - max_left; ++i) r.seq.emplace_back((char)ls->get((u64)(lcl - 1 - max_left - i)));
        for (int i = 0; i < gap_length; ++i) r.seq.emplace_back((char)4);
        for (int i = 0; i < rcl - max_right; ++i) r.seq.emplace_back((char)rs->get((u64)i));
        return true;
    }
    // gapCloseDBG.h:687-741: a gap at a scaffold's end: the node that continues the one flank there is
    template <int W> bool extend_edge(Graph<W> &g, GapClosure &r) const
    {
        double min_rate = 1.0;
        long min_rate_overlap = 0;
        const Straight *best = nullptr;
        for (auto &e : g.left) {
            const Straight &st = e.second;
            if (st.coverage == 65535) continue;
            const auto edge = gap->start != 0 ? mismatch_left(st) : mismatch_right(st);
            if (edge.second < min_rate) { min_rate_overlap = edge.first; min_rate = edge.second; best = &st; }
        }
        if (min_rate_overlap == 0) return false;
        const long contig_length = (long)(best->length + max_k - 1);
        r.seq_length = (int)(contig_length - min_rate_overlap);
        if (r.seq_length <= 0) return false;
        if (gap->start != 0) for (long i = 0; i < r.seq_length; ++i) r.seq.emplace_back((char)best->get((u64)(contig_length - 1 - min_rate_overlap - i)));
        else for (long i = 0; i < r.seq_length; ++i) r.seq.emplace_back((char)best->get((u64)(contig_length - 1 - i)));
        return true;
    }
};

} // namespace pbh
