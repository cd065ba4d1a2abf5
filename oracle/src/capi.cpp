// TEST INFRASTRUCTURE ONLY -- C entry points of the oracle for ctypes (tests/, smoke(), bench cpu_baseline).
#include "assemble.hpp"
#include "counter.hpp"
#include "fastx.hpp"
#include "../../platanus_b_amd/host/schedule.hpp"
#include <cstring>
#include <memory>

using namespace pbo;

namespace {
struct Handle {
    int W = 0;
    unsigned k = 0;
    std::unique_ptr<Counter<1>> c1; std::unique_ptr<Counter<2>> c2; std::unique_ptr<Counter<3>> c3;
    std::unique_ptr<Counter<4>> c4; std::unique_ptr<Counter<5>> c5;
    std::vector<u64> sorted;     // n * W
    std::vector<u16> sorted_counts;
    u64 dh_size = 0;
};
template <int W> Counter<W> &get(Handle *h);
template <> Counter<1> &get<1>(Handle *h) { return *h->c1; }
template <> Counter<2> &get<2>(Handle *h) { return *h->c2; }
template <> Counter<3> &get<3>(Handle *h) { return *h->c3; }
template <> Counter<4> &get<4>(Handle *h) { return *h->c4; }
template <> Counter<5> &get<5>(Handle *h) { return *h->c5; }

template <int W> void do_count(Handle *h, const ReadSet &rs, unsigned k, u64 memory)
{
    Counter<W> &c = get<W>(h);
    h->dh_size = c.count_reads(k, rs, memory);
}
template <int W> void do_retain(Handle *h, u64 min_occ)
{
    Counter<W> &c = get<W>(h);
    // (key,count) pairs with count >= min_occ sorted by key (sorted_keys() + the counts that go with them)
    std::vector<std::pair<Key<W>, u16>> v;
    for (size_t i = 0; i < c.kmer_keys.size(); ++i) if (c.kmer_vals[i] >= min_occ) v.push_back({c.kmer_keys[i], c.kmer_vals[i]});
    std::sort(v.begin(), v.end(), [](const std::pair<Key<W>, u16> &a, const std::pair<Key<W>, u16> &b) { return a.first < b.first; });
    h->sorted.resize(v.size() * W);
    h->sorted_counts.resize(v.size());
    for (size_t i = 0; i < v.size(); ++i) {
        for (int w = 0; w < W; ++w) h->sorted[i * W + w] = v[i].first.w[w];
        h->sorted_counts[i] = v[i].second;
    }
}
} // namespace

extern "C" {

void *pbo_create() { return new Handle(); }
void pbo_destroy(void *p) { delete (Handle *)p; }

// counter.h:275-383 at -t 1.  codes: 0..3, 4 = N.  Returns doubleHashSize, or 0 on error.
unsigned long long pbo_count(void *p, const uint8_t *codes, const unsigned long long *off, unsigned long long n_reads,
                             unsigned k, unsigned long long memory_bytes)
{
    Handle *h = (Handle *)p;
    ReadSet rs;
    rs.bases.assign(codes + off[0], codes + off[n_reads]);
    rs.off.resize(n_reads + 1);
    for (u64 i = 0; i <= n_reads; ++i) rs.off[i] = off[i] - off[0];
    h->k = k;
    h->W = words_for_k(k);
    try {
        switch (h->W) {
        case 1: h->c1.reset(new Counter<1>()); do_count<1>(h, rs, k, memory_bytes); break;
        case 2: h->c2.reset(new Counter<2>()); do_count<2>(h, rs, k, memory_bytes); break;
        case 3: h->c3.reset(new Counter<3>()); do_count<3>(h, rs, k, memory_bytes); break;
        case 4: h->c4.reset(new Counter<4>()); do_count<4>(h, rs, k, memory_bytes); break;
        case 5: h->c5.reset(new Counter<5>()); do_count<5>(h, rs, k, memory_bytes); break;
        default: return 0;
        }
    } catch (...) { return 0; }
    return h->dh_size;
}

#define DISPATCH(h, expr) switch ((h)->W) { case 1: { auto &c = get<1>(h); expr; } break; case 2: { auto &c = get<2>(h); expr; } break; \
    case 3: { auto &c = get<3>(h); expr; } break; case 4: { auto &c = get<4>(h); expr; } break; default: { auto &c = get<5>(h); expr; } break; }

void pbo_occ_hist(void *p, unsigned long long *hist /*65535*/, unsigned long long *max_occ)
{
    Handle *h = (Handle *)p;
    DISPATCH(h, { for (int i = 0; i < 65535; ++i) hist[i] = c.occ_hist[i]; *max_occ = c.max_occ; });
}
void pbo_length_hist(void *p, unsigned long long *hist, unsigned long long n)
{
    Handle *h = (Handle *)p;
    DISPATCH(h, { for (u64 i = 0; i < n; ++i) hist[i] = i < c.length_hist.size() ? c.length_hist[i] : 0; });
}
unsigned long long pbo_left_local_minimal_value(void *p, unsigned long long window)
{
    Handle *h = (Handle *)p; u64 r = 0;
    DISPATCH(h, r = c.left_local_minimal_value(window));
    return r;
}
unsigned long long pbo_num_distinct(void *p)
{
    Handle *h = (Handle *)p; u64 r = 0;
    DISPATCH(h, r = c.kmer_keys.size());
    return r;
}
// kmerFP order (counter.h:482-507)
void pbo_fetch_file_order(void *p, unsigned long long *keys, uint16_t *counts)
{
    Handle *h = (Handle *)p;
    DISPATCH(h, { const int W = (int)(sizeof(c.kmer_keys[0]) / 8);
        for (size_t i = 0; i < c.kmer_keys.size(); ++i) { for (int w = 0; w < W; ++w) keys[i * W + w] = c.kmer_keys[i].w[w]; counts[i] = c.kmer_vals[i]; } });
}
unsigned long long pbo_retain(void *p, unsigned long long min_occ)
{
    Handle *h = (Handle *)p;
    switch (h->W) { case 1: do_retain<1>(h, min_occ); break; case 2: do_retain<2>(h, min_occ); break; case 3: do_retain<3>(h, min_occ); break;
                    case 4: do_retain<4>(h, min_occ); break; default: do_retain<5>(h, min_occ); break; }
    return h->sorted_counts.size();
}
void pbo_fetch_retained(void *p, unsigned long long *keys, uint16_t *counts)
{
    Handle *h = (Handle *)p;
    if (!h->sorted.empty()) memcpy(keys, h->sorted.data(), h->sorted.size() * 8);
    if (!h->sorted_counts.empty()) memcpy(counts, h->sorted_counts.data(), h->sorted_counts.size() * 2);
}
// loadKmer + outputOccurrenceTableBinary (counter.h:599-640,954-963); consumes the kmer file
int pbo_write_table_binary(void *p, unsigned long long min_occ, const char *path)
{
    Handle *h = (Handle *)p;
    try { DISPATCH(h, { c.load_kmers(min_occ, h->dh_size); c.write_table_binary(path); }); } catch (...) { return -1; }
    return 0;
}
int pbo_write_occ_distribution(void *p, const char *path)
{
    Handle *h = (Handle *)p;
    try { DISPATCH(h, c.write_occ_distribution(path)); } catch (...) { return -1; }
    return 0;
}

// whole pipeline on in-memory reads; argv-style options as for `platanus_b assemble`
int pbo_assemble(const uint8_t *codes, const unsigned long long *off, unsigned long long n_reads, const char *out_prefix,
                 unsigned k0, unsigned step, unsigned n_cutoff, unsigned long long memory_gb, int kmer_occ_only, int verbose, const char *bubble)
{
    ReadSet rs;
    rs.bases.assign(codes + off[0], codes + off[n_reads]);
    rs.off.resize(n_reads + 1);
    for (u64 i = 0; i <= n_reads; ++i) rs.off[i] = off[i] - off[0];
    AssembleOptions o;
    o.out_prefix = out_prefix; o.k0 = k0; o.step = step; o.n_cutoff = n_cutoff; o.memory_gb = memory_gb;
    o.kmer_occ_only = kmer_occ_only != 0; o.verbose = verbose != 0;
    if (bubble && *bubble) o.bubble = bubble;
    try { return run_assemble(o, rs); } catch (const std::exception &e) { fprintf(stderr, "oracle: %s\n", e.what()); return -1; }
}

// schedule helper (assemble.cpp:657-736) for host-logic tests
unsigned long long pbo_extend_kmer(double min_log_p_join, double avg_cov, double avg_len, unsigned long long min_cov, double max_k_ratio,
                                   unsigned k0, unsigned long long cutoff0, unsigned long long step, unsigned *k_out,
                                   unsigned long long *cutoff_out, unsigned cap)
{
    std::vector<unsigned> k{k0};
    std::vector<u64> c{cutoff0};
    const u64 n = extend_kmer(min_log_p_join, avg_cov, avg_len, min_cov, max_k_ratio, k, c, step, false);
    for (u64 i = 0; i < n && i < cap; ++i) { k_out[i] = k[i]; cutoff_out[i] = c[i]; }
    return n;
}

} // extern "C"

// ---- host-logic check without a GPU ---------------------------------------------------------------------------------
// Builds the retained set with the oracle's counter, computes the adjacency arrays the device would deliver
// (pb_graph_adjacency's contract, include/pbgpu.h) with plain CPU probes, runs the PRODUCT's DFS replay
// (platanus_b_amd/host/device_graph.hpp) and the oracle's k-mer-level DFS on the same set, and compares the two node
// tables in container iteration order.  Both product replays are checked: the k-mer-level one on the adjacency arrays
// (codes 1-7) and the node-level one on the compact hand-over of pb_graph_chains / pb_graph_nodes, which is restated here
// with plain loops (codes 11-17).  Returns 0 when identical, a positive code for the first difference.
#include "graph.hpp"
#include "../../platanus_b_amd/host/device_graph.hpp"

namespace {
template <int W>
int replay_check(const ReadSet &rs, unsigned k, u64 min_occ)
{
    Counter<W> c;
    const u64 dh = c.count_reads(k, rs, 1000000000ull);
    std::vector<Key<W>> sorted = c.sorted_keys(min_occ);
    // counts in sorted order + CPU adjacency
    std::vector<std::pair<Key<W>, u16>> kv;
    for (size_t i = 0; i < c.kmer_keys.size(); ++i) if (c.kmer_vals[i] >= min_occ) kv.push_back({c.kmer_keys[i], c.kmer_vals[i]});
    std::sort(kv.begin(), kv.end(), [](const std::pair<Key<W>, u16> &a, const std::pair<Key<W>, u16> &b) { return a.first < b.first; });
    const u64 n = kv.size();
    std::vector<u64> keys(n * W);
    std::vector<u16> counts(n);
    for (u64 i = 0; i < n; ++i) { for (int w = 0; w < W; ++w) keys[i * W + w] = kv[i].first.w[w]; counts[i] = kv[i].second; }
    c.load_kmers(min_occ, dh);
    Graph<W> ref;
    ref.set_k(k);
    ref.make_initial(c.table, sorted);

    auto find_idx = [&](const Key<W> &canon) -> long {
        auto it = std::lower_bound(kv.begin(), kv.end(), canon, [](const std::pair<Key<W>, u16> &a, const Key<W> &b) { return a.first < b; });
        return (it != kv.end() && it->first == canon) ? (long)(it - kv.begin()) : -1;
    };
    std::vector<uint8_t> adj(n);
    std::vector<uint32_t> nbr(2 * n, 0xFFFFFFFFu);
    Graph<W> tmp;
    tmp.set_k(k);
    for (u64 i = 0; i < n; ++i) {
        unsigned rf = 0, lf = 0;
        uint32_t rn = 0xFFFFFFFFu, ln = 0xFFFFFFFFu;
        for (unsigned b = 0; b < 4; ++b) {
            for (int side = 0; side < 2; ++side) {
                const Key<W> s = side == 0 ? tmp.succ(kv[i].first, b) : tmp.pred(kv[i].first, b);
                const Key<W> r = revcomp(s, k);
                const bool flip = r < s;
                const long j = find_idx(flip ? r : s);
                if (j < 0) continue;
                if (side == 0) { rf |= 1u << b; rn = ((uint32_t)j << 1) | (flip ? 1u : 0u); }
                else { lf |= 1u << b; ln = ((uint32_t)j << 1) | (flip ? 1u : 0u); }
            }
        }
        adj[i] = (uint8_t)((lf << 4) | rf);
        if (flag_count(rf) == 1) nbr[2 * i] = rn;
        if (flag_count(lf) == 1) nbr[2 * i + 1] = ln;
    }
    DeviceGraphInput<W> in;
    in.k = k; in.n = n; in.keys = keys.data(); in.counts = counts.data(); in.adj = adj.data(); in.nbr = nbr.data();
    Graph<W> got;
    got.set_k(k);
    make_initial_from_device(got, in);

    // ---- the compact hand-over (pb_graph_chains + pb_graph_nodes contract, include/pbgpu.h) restated with plain loops, and the
    //      product's node-level replay on it
    Graph<W> got2;
    got2.set_k(k);
    {
        auto is_junction = [&](u64 i) { return flag_count(adj[i] >> 4) > 1 || flag_count(adj[i] & 15u) > 1; };
        const u64 NONE = ~0ull;
        std::vector<u64> nxt(2 * n, NONE);
        for (u64 s = 0; s < 2 * n; ++s) {
            const u64 i = s >> 1;
            const unsigned d = (unsigned)(s & 1);
            if (is_junction(i)) continue;
            const uint32_t v = nbr[2 * i + d];
            if (v == 0xFFFFFFFFu) continue;
            const u64 j = v >> 1;
            if (!is_junction(j)) nxt[s] = 2 * j + (d ^ (v & 1u));
        }
        // terminal state and distance of every state (NONE: on a cycle)
        auto walk = [&](u64 s, u64 &dist) -> u64 {
            dist = 0;
            u64 cur = s;
            while (nxt[cur] != NONE) { cur = nxt[cur]; ++dist; if (dist > 2 * n) return NONE; }
            return cur;
        };
        std::vector<uint32_t> chain_of(n, CHAIN_JUNCTION), pos_a(n, 0);
        std::vector<uint8_t> orient(n, 0);
        std::vector<u64> term_a(n, NONE);
        for (u64 i = 0; i < n; ++i) {
            if (is_junction(i)) continue;
            u64 d0, d1;
            const u64 t0 = walk(2 * i, d0), t1 = walk(2 * i + 1, d1);
            if (t0 == NONE || t1 == NONE || t0 == t1) { chain_of[i] = CHAIN_COMPLEX; continue; }
            const bool a_is_0 = t0 < t1;
            term_a[i] = a_is_0 ? t0 : t1;
            pos_a[i] = (uint32_t)(a_is_0 ? d0 : d1);
            orient[i] = a_is_0 ? 1 : 0;
            chain_of[i] = 0;
        }
        // dense chain ids in an arbitrary order (descending index of the A end, to differ from any natural order)
        ChainRecords cr;
        std::vector<long> id_of_a(n, -1);
        for (u64 ii = n; ii-- > 0;) {
            if (chain_of[ii] >= CHAIN_COMPLEX || pos_a[ii] != 0) continue;
            id_of_a[ii] = (long)cr.len.size();
            u64 db;
            const u64 sb = 2 * ii + (orient[ii] ? 1u : 0u);
            const u64 tb = walk(sb, db);
            cr.len.push_back((uint32_t)(db + 1));
            cr.end_a.push_back((uint32_t)((ii << 1) | orient[ii]));
            cr.end_b.push_back((uint32_t)(((tb >> 1) << 1) | (tb & 1u)));
            cr.sum.push_back(0);
            cr.root.push_back(0xFFFFFFFFu);
            cr.seq_woff.push_back(0);
        }
        const u64 nc = cr.len.size();
        u64 words = 0;
        for (u64 c2 = 0; c2 < nc; ++c2) { cr.seq_woff[c2] = words; words += ((u64)cr.len[c2] + k - 1 + 31) / 32; }
        cr.seq.assign(words, 0);
        for (u64 i = 0; i < n; ++i) {
            if (chain_of[i] >= CHAIN_COMPLEX) continue;
            const uint32_t cid = (uint32_t)id_of_a[term_a[i] >> 1];
            chain_of[i] = cid;
            cr.sum[cid] += counts[i];
            cr.root[cid] = std::min(cr.root[cid], (uint32_t)((i << 1) | orient[i]));
            Key<W> key = kv[i].first;
            if (orient[i]) key = revcomp(key, k);
            const u64 ls = (u64)cr.len[cid] + k - 1;
            u64 *dst = cr.seq.data() + cr.seq_woff[cid];
            auto put = [&](u64 p, unsigned code) { dst[p >> 5] |= (u64)code << ((p & 31) * 2); };
            if (pos_a[i] == 0) for (unsigned b = 0; b < k; ++b) put(ls - 1 - b, key.get(k - 1 - b));
            else put(ls - 1 - (k - 1 + pos_a[i]), key.get(0));
        }
        NodeSet<W> nd;
        nd.k = k;
        std::vector<uint8_t> mark(n, 0);
        for (u64 c2 = 0; c2 < nc; ++c2) { mark[cr.end_a[c2] >> 1] = 1; mark[cr.end_b[c2] >> 1] = 1; }
        for (u64 ii = n; ii-- > 0;) {                              // unsorted on purpose: the replay sorts
            if (!(chain_of[ii] >= CHAIN_COMPLEX || mark[ii])) continue;
            nd.idx.push_back((uint32_t)ii);
            nd.meta.push_back((uint32_t)adj[ii] | ((uint32_t)(chain_of[ii] < CHAIN_COMPLEX ? orient[ii] : 0) << 8) | ((uint32_t)counts[ii] << 16));
            nd.chain.push_back(chain_of[ii]);
            nd.nbr.push_back(nbr[2 * ii]);
            nd.nbr.push_back(nbr[2 * ii + 1]);
            for (int w = 0; w < W; ++w) nd.keys.push_back(kv[ii].first.w[w]);
            for (unsigned b = 0; b < 8; ++b) {
                uint32_t e = 0xFFFFFFFFu;
                if (chain_of[ii] == CHAIN_JUNCTION) {
                    const Key<W> x = b < 4 ? tmp.succ(kv[ii].first, b) : tmp.pred(kv[ii].first, b - 4);
                    const Key<W> r = revcomp(x, k);
                    const bool flip = r < x;
                    const long j = find_idx(flip ? r : x);
                    if (j >= 0) e = ((uint32_t)j << 1) | (flip ? 1u : 0u);
                }
                nd.nb8.push_back(e);
            }
        }
        make_initial_from_nodes(got2, nd, cr);
    }
    for (int which = 0; which < 2; ++which) {
        Graph<W> &gx = which == 0 ? got : got2;
        const int base = which * 10;
        if (gx.left.size() != ref.left.size()) return base + 1;
        if (gx.junctions.size() != ref.junctions.size()) return base + 2;
        auto a = gx.left.begin();
        for (auto b = ref.left.begin(); b != ref.left.end(); ++a, ++b) {
            if (!(a->first == b->first)) return base + 3;
            if (a->second.length != b->second.length || a->second.coverage != b->second.coverage || a->second.out != b->second.out) return base + 4;
            if (a->second.seq != b->second.seq) return base + 5;
        }
        auto ja = gx.junctions.begin();
        for (auto jb = ref.junctions.begin(); jb != ref.junctions.end(); ++ja, ++jb) {
            if (!(ja->first == jb->first)) return base + 6;
            if (ja->second.coverage != jb->second.coverage || ja->second.out != jb->second.out) return base + 7;
        }
    }
    return 0;
}
} // namespace

extern "C" int pbo_replay_check(const uint8_t *codes, const unsigned long long *off, unsigned long long n_reads, unsigned k,
                                unsigned long long min_occ, unsigned long long *n_straight, unsigned long long *n_junction)
{
    ReadSet rs;
    rs.bases.assign(codes + off[0], codes + off[n_reads]);
    rs.off.resize(n_reads + 1);
    for (u64 i = 0; i <= n_reads; ++i) rs.off[i] = off[i] - off[0];
    (void)n_straight; (void)n_junction;
    try {
        switch (words_for_k(k)) {
        case 1: return replay_check<1>(rs, k, min_occ);
        case 2: return replay_check<2>(rs, k, min_occ);
        case 3: return replay_check<3>(rs, k, min_occ);
        case 4: return replay_check<4>(rs, k, min_occ);
        default: return replay_check<5>(rs, k, min_occ);
        }
    } catch (...) { return -1; }
}

// ---- host-logic check without a GPU: the product's piece-wise parallel reader against the oracle's sequential one -------
#include "fastx.hpp"
#include "../../platanus_b_amd/host/read_files.hpp"

// parts > 1: the byte ranges of `parts` ranks (read_file's part / parts), concatenated in rank order, must be the sequential result too
extern "C" int pbo_parse_file_check_parts(const char *path, unsigned max_threads, unsigned parts, unsigned long long *n_reads, unsigned long long *n_pieces)
{
    try {
        pbo::ReadSet rs;
        pbo::read_fastx(path, rs);
        std::vector<pbh::ParsedPiece> pieces;
        for (unsigned part = 0; part < std::max(parts, 1u); ++part) {
            std::vector<pbh::ParsedPiece> mine;
            pbh::read_file(path, mine, max_threads, part, std::max(parts, 1u));
            for (auto &pc : mine) pieces.push_back(std::move(pc));
        }
        static const unsigned char tab[16] = {46, 0, 46, 1, 3, 46, 46, 2, 46, 46, 46, 46, 46, 46, 4, 46};   // common.h:256, as the pack kernel applies it
        u64 r = 0;
        for (const pbh::ParsedPiece &pc : pieces) {
            for (size_t i = 0; i + 1 < pc.off.size(); ++i, ++r) {
                if (r + 1 >= rs.off.size()) return 1;                                  // more reads than the sequential reader found
                const u64 len = pc.off[i + 1] - pc.off[i];
                if (len != rs.off[r + 1] - rs.off[r]) return 2;
                for (u64 j = 0; j < len; ++j) {
                    unsigned c = tab[(unsigned char)pc.bases[pc.off[i] + j] & 0xF];
                    if (c > 3) c = 4;
                    if (c != rs.bases[rs.off[r] + j]) return 3;
                }
            }
        }
        if (r + 1 != rs.off.size()) return 4;
        if (n_reads) *n_reads = r;
        if (n_pieces) *n_pieces = pieces.size();
        return 0;
    } catch (...) { return -1; }
}

extern "C" int pbo_parse_file_check(const char *path, unsigned max_threads, unsigned long long *n_reads, unsigned long long *n_pieces)
{
    return pbo_parse_file_check_parts(path, max_threads, 1, n_reads, n_pieces);
}

// ---- local de Bruijn graphs of gap closing (gapclose.hpp): a batch of gaps, nodes in the reference's insertion order ---------------------
#include "gapclose.hpp"
#include "../../platanus_b_amd/host/gap_graphs.hpp"
#include <sstream>
namespace {
struct GapResult {
    std::vector<unsigned long long> node_first, word_off, words;
    std::vector<unsigned char> kind, out;
    std::vector<unsigned short> cov;
    std::vector<unsigned> length;
    std::vector<std::string> dumps;          // per gap: the tables of a fresh Graph<W> after the insertions, in iteration order
};
GapResult g_gap;

template <int W>
void gap_run(unsigned k, long min_cov, unsigned long long n_gaps, const unsigned long long *seq_first, const unsigned long long *seq_off, const unsigned char *codes)
{
    g_gap = GapResult();
    g_gap.node_first.push_back(0);
    g_gap.word_off.push_back(0);
    for (unsigned long long g = 0; g < n_gaps; ++g) {
        pbo::GapGraph<W> gg;
        gg.k = k;
        std::vector<std::vector<unsigned char>> seqs;
        for (unsigned long long s = seq_first[g]; s < seq_first[g + 1]; ++s) seqs.emplace_back(codes + seq_off[s], codes + seq_off[s + 1]);
        gg.count(seqs);
        const auto nodes = gg.make_graph(min_cov);
        pbh::Graph<W> gr;
        gr.set_k(k);
        for (const auto &n : nodes) {
            g_gap.kind.push_back(n.straight ? 1 : 0);
            g_gap.cov.push_back(n.straight ? n.st.coverage : n.jn.coverage);
            g_gap.out.push_back(n.straight ? n.st.out : n.jn.out);
            g_gap.length.push_back(n.straight ? (unsigned)n.st.length : 1u);
            if (n.straight) { for (auto w : n.st.seq) g_gap.words.push_back(w); gr.insert_straight(n.st); }
            else { for (int w = 0; w < W; ++w) g_gap.words.push_back(n.key.w[w]); gr.junctions[n.key] = n.jn; }
            g_gap.word_off.push_back(g_gap.words.size());
        }
        g_gap.node_first.push_back(g_gap.kind.size());
        // the tables as the reference's own would be walked (format shared with oracle/ref_harness/gapclose_ref.cpp)
        std::ostringstream o;
        o << std::hex;
        for (auto &e : gr.junctions) { o << "J"; for (int w = W - 1; w >= 0; --w) o << " " << e.first.w[w]; o << " " << e.second.coverage << " " << (unsigned)e.second.out << "\n"; }
        for (auto &e : gr.left) {
            o << "S " << e.second.length << " " << e.second.coverage << " " << (unsigned)e.second.out;
            const unsigned long long nb = e.second.length + k - 1;
            for (unsigned long long i = 0; i < (nb + 31) / 32; ++i) {
                unsigned long long w = i < e.second.seq.size() ? e.second.seq[i] : 0;
                if (i == (nb + 31) / 32 - 1 && (nb % 32)) w &= ~0ull >> (64 - 2 * (nb % 32));
                o << " " << w;
            }
            o << "\n";
        }
        g_gap.dumps.push_back(o.str());
    }
}
} // namespace

extern "C" int pbo_gap_graphs_run(unsigned k, long min_cov, unsigned long long n_gaps, const unsigned long long *seq_first, const unsigned long long *seq_off,
                                  const unsigned char *codes, unsigned long long *n_nodes, unsigned long long *n_words)
{
    try {
        switch (pbh::words_for_k(k)) {
        case 1: gap_run<1>(k, min_cov, n_gaps, seq_first, seq_off, codes); break;
        case 2: gap_run<2>(k, min_cov, n_gaps, seq_first, seq_off, codes); break;
        case 3: gap_run<3>(k, min_cov, n_gaps, seq_first, seq_off, codes); break;
        case 4: gap_run<4>(k, min_cov, n_gaps, seq_first, seq_off, codes); break;
        case 5: gap_run<5>(k, min_cov, n_gaps, seq_first, seq_off, codes); break;
        default: return -2;
        }
    } catch (...) { return -1; }
    *n_nodes = g_gap.kind.size();
    *n_words = g_gap.words.size();
    return 0;
}
extern "C" void pbo_gap_graphs_fetch(unsigned long long *node_first, unsigned char *kind, unsigned short *cov, unsigned char *out, unsigned *length,
                                     unsigned long long *word_off, unsigned long long *words)
{
    std::copy(g_gap.node_first.begin(), g_gap.node_first.end(), node_first);
    std::copy(g_gap.kind.begin(), g_gap.kind.end(), kind);
    std::copy(g_gap.cov.begin(), g_gap.cov.end(), cov);
    std::copy(g_gap.out.begin(), g_gap.out.end(), out);
    std::copy(g_gap.length.begin(), g_gap.length.end(), length);
    std::copy(g_gap.word_off.begin(), g_gap.word_off.end(), word_off);
    std::copy(g_gap.words.begin(), g_gap.words.end(), words);
}
// the table dump of gap g (NUL-terminated; valid until the next pbo_gap_graphs_run)
extern "C" const char *pbo_gap_graphs_dump(unsigned long long g) { return g < g_gap.dumps.size() ? g_gap.dumps[g].c_str() : ""; }

// ---- the whole graph part of a gap's assembly (gapCloseDBG.h:416-436), gap by gap: dumps of the low and the high graph ----------------------
namespace {
template <int W> void dump_graph(std::ostringstream &o, pbh::Graph<W> &gr, unsigned k)
{
    for (auto &e : gr.junctions) {
        if (e.second.coverage == 65535) continue;
        o << "J"; for (int w = W - 1; w >= 0; --w) o << " " << e.first.w[w]; o << " " << e.second.coverage << " " << (unsigned)e.second.out << "\n";
    }
    for (auto &e : gr.left) {
        if (e.second.coverage == 65535) continue;
        o << "S " << e.second.length << " " << e.second.coverage << " " << (unsigned)e.second.out;
        const unsigned long long nb = e.second.length + k - 1;
        for (unsigned long long i = 0; i < (nb + 31) / 32; ++i) {
            unsigned long long w = i < e.second.seq.size() ? e.second.seq[i] : 0;
            if (i == (nb + 31) / 32 - 1 && (nb % 32)) w &= ~0ull >> (64 - 2 * (nb % 32));
            o << " " << w;
        }
        o << "\n";
    }
}
std::vector<std::string> g_asm_dumps;
// flanks (optional): per gap i32 start, i32 end, u8 at a scaffold's end, u32 head length, u32 tail length, head codes, tail codes
struct FlankReader {
    const unsigned char *p = nullptr;
    bool next(pbh::GapFlanks &f, bool &at_end)
    {
        if (!p) return false;
        int32_t se[2]; uint32_t hl, tl;
        memcpy(se, p, 8); at_end = p[8] != 0; memcpy(&hl, p + 9, 4); memcpy(&tl, p + 13, 4);
        p += 17;
        f.start = se[0]; f.end = se[1];
        f.head.assign((const char *)p, (const char *)p + hl); p += hl;
        f.tail.assign((const char *)p, (const char *)p + tl); p += tl;
        return true;
    }
};
void closure_line(std::ostringstream &o, const char *tag, bool ok, const pbh::GapClosure &c, bool with_remained)
{
    o << std::dec << tag << " " << (ok ? 1 : 0) << " " << (ok ? c.seq_length : 0);
    if (with_remained) o << " " << (ok ? c.remained_gap : 0);
    if (ok) for (char b : c.seq) o << " " << (int)b;
    o << "\n";
}
template <int WL, int WH>
void asm_run(unsigned kl, unsigned kh, long min_cov, unsigned long long n_gaps, const unsigned long long *seq_first, const unsigned long long *seq_off, const unsigned char *codes,
             const unsigned char *flanks)
{
    g_asm_dumps.clear();
    FlankReader fr{flanks};
    for (unsigned long long g = 0; g < n_gaps; ++g) {
        std::vector<std::vector<unsigned char>> seqs;
        for (unsigned long long s = seq_first[g]; s < seq_first[g + 1]; ++s) seqs.emplace_back(codes + seq_off[s], codes + seq_off[s + 1]);
        pbh::GapFlanks fl;
        bool at_end = false;
        const bool have_flanks = fr.next(fl, at_end);
        pbh::GapCloser cl;
        cl.max_k = kh; cl.gap = &fl;
        std::ostringstream o;
        if (have_flanks && !at_end && !cl.enabled(seqs.size() + 2)) { g_asm_dumps.push_back("X\n"); continue; }      // (+ 2: the entries of GAP::seq that are not counted)
        pbh::Graph<WL> low;
        pbh::Graph<WH> high;
        pbo::gap_assemble<WL, WH>(seqs, kl, kh, min_cov, low, high);
        o << std::hex << "L\n";
        dump_graph<WL>(o, low, kl);
        o << "H\n";
        dump_graph<WH>(o, high, kh);
        if (have_flanks) {
            if (!at_end) {
                pbh::GapClosure c;
                const bool closed = cl.closed(high, c);
                closure_line(o, "C", closed, c, false);
                if (!closed) { pbh::GapClosure pc; const bool part = cl.closed_partial(high, pc); closure_line(o, "P", part, pc, true); }
            } else {
                pbh::GapClosure c;
                const bool ext = cl.extend_edge(high, c);
                closure_line(o, "E", ext, c, false);
            }
        }
        g_asm_dumps.push_back(o.str());
    }
}
} // namespace
extern "C" int pbo_gap_assemble_run(unsigned k_low, unsigned k_high, long min_cov, unsigned long long n_gaps, const unsigned long long *seq_first,
                                    const unsigned long long *seq_off, const unsigned char *codes, const unsigned char *flanks)
{
    const int wl = pbh::words_for_k(k_low), wh = pbh::words_for_k(k_high);
    try {
        if (wl == 1 && wh == 1) asm_run<1, 1>(k_low, k_high, min_cov, n_gaps, seq_first, seq_off, codes, flanks);
        else if (wl == 1 && wh == 2) asm_run<1, 2>(k_low, k_high, min_cov, n_gaps, seq_first, seq_off, codes, flanks);
        else if (wl == 1 && wh == 3) asm_run<1, 3>(k_low, k_high, min_cov, n_gaps, seq_first, seq_off, codes, flanks);
        else if (wl == 1 && wh == 4) asm_run<1, 4>(k_low, k_high, min_cov, n_gaps, seq_first, seq_off, codes, flanks);
        else if (wl == 2 && wh == 3) asm_run<2, 3>(k_low, k_high, min_cov, n_gaps, seq_first, seq_off, codes, flanks);
        else return -2;
    } catch (...) { return -1; }
    return 0;
}
extern "C" const char *pbo_gap_assemble_dump(unsigned long long g) { return g < g_asm_dumps.size() ? g_asm_dumps[g].c_str() : ""; }
