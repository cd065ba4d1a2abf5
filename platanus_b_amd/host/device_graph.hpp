// BruijnGraph::makeInitialBruijnGraph (graph.h:289-579) on top of the device's adjacency arrays.
//
// The reference discovers nodes with a sequential stack DFS whose visiting order and strand choices end up in the
// output (contig numbering = unordered_map insertion order, contig strand = strand of the DFS entry k-mer).
// All of its cost is in hash probes: 8 neighbour probes per k-mer plus one per extension step.  Here those probes
// were done by the GPU (pb_graph_adjacency: adj[] = presence flags, nbr[] = index + strand of a unique neighbour), so the
// same DFS -- same roots in ascending key order, same stack discipline, same stop rules, same "visited" semantics
// for cycles and hairpins -- is replayed on the host as O(n) array steps.  An oriented k-mer is (index, flip): flip = 1
// means "reverse complement of the canonical key at that index".
#pragma once
#include "node_graph.hpp"
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace pbh {

inline unsigned rev4(unsigned f) { return ((f & 1) << 3) | ((f & 2) << 1) | ((f & 4) >> 1) | ((f & 8) >> 3); }

template <int W>
struct DeviceGraphInput {
    unsigned k = 0;
    u64 n = 0;
    const u64 *keys = nullptr;        // n * W, ascending canonical keys
    const u16 *counts = nullptr;      // n
    const uint8_t *adj = nullptr;     // n
    const uint32_t *nbr = nullptr;    // 2n

    Key<W> key(u64 i) const { Key<W> r; for (int w = 0; w < W; ++w) r.w[w] = keys[i * W + w]; return r; }
    Key<W> oriented(u64 i, unsigned f) const { const Key<W> c = key(i); return f ? revcomp(c, k) : c; }
    unsigned rflags(u64 i, unsigned f) const { return f ? rev4(adj[i] >> 4) : (adj[i] & 15u); }
    unsigned lflags(u64 i, unsigned f) const { return f ? rev4(adj[i] & 15u) : (adj[i] >> 4); }
    // unique neighbour on the right / left of oriented k-mer (i,f); valid only when that side has exactly one flag
    void right_of(u64 i, unsigned f, u64 &j, unsigned &g) const
    {
        const uint32_t v = f ? nbr[2 * i + 1] : nbr[2 * i];
        j = v >> 1;
        g = (v & 1u) ^ f;
    }
    void left_of(u64 i, unsigned f, u64 &j, unsigned &g) const
    {
        const uint32_t v = f ? nbr[2 * i] : nbr[2 * i + 1];
        j = v >> 1;
        g = (v & 1u) ^ f;
    }
    // index + strand of an arbitrary oriented k-mer that is known to be in the set (binary search; used for
    // junction neighbours only, i.e. O(#junctions) times)
    void locate(const Key<W> &s, u64 &j, unsigned &g) const
    {
        const Key<W> r = revcomp(s, k);
        const bool flip = r < s;
        const Key<W> &c = flip ? r : s;
        u64 lo = 0, hi = n;
        while (lo < hi) {
            const u64 mid = (lo + hi) >> 1;
            if (key(mid) < c) lo = mid + 1; else hi = mid;
        }
        j = lo;
        g = flip ? 1u : 0u;
    }
};

template <int W>
void make_initial_from_device(Graph<W> &gr, const DeviceGraphInput<W> &in)
{
    typedef Key<W> K;
    const unsigned k = in.k;
    gr.junctions.clear();
    gr.left.clear();
    gr.right.clear();
    std::vector<uint8_t> visited(in.n, 0);
    struct Ori { u64 i; unsigned f; };
    std::vector<Ori> stack;
    std::vector<unsigned char> lseq, rseq;
    for (u64 root = 0; root < in.n; ++root) {
        if (visited[root]) continue;              // the reference pops it and drops it right away (graph.h:337-341)
        stack.clear();
        stack.push_back({root, 0u});
        while (!stack.empty()) {
            const Ori s = stack.back();
            stack.pop_back();
            if (visited[s.i]) continue;
            visited[s.i] = 1;
            u64 sum = in.counts[s.i];
            unsigned lfl = in.lflags(s.i, s.f), rfl = in.rflags(s.i, s.f);
            const K start = in.oriented(s.i, s.f);
            if (flag_count(lfl) > 1 || flag_count(rfl) > 1) {                    // graph.h:383-404
                for (unsigned b = 0; b < 4; ++b) {
                    Ori o;
                    if (rfl & (1u << b)) { in.locate(gr.succ(start, b), o.i, o.f); stack.push_back(o); }
                    if (lfl & (1u << b)) { in.locate(gr.pred(start, b), o.i, o.f); stack.push_back(o); }
                }
                Junction j;
                j.coverage = (u16)sum;
                j.out = (uint8_t)((lfl << 4) | rfl);
                gr.junctions[start] = j;
                continue;
            }
            lseq.clear();
            rseq.clear();
            if (lfl) lseq.push_back((unsigned char)flag2base(lfl));
            if (rfl) rseq.push_back((unsigned char)flag2base(rfl));
            Ori rstop = s, lstop = s;
            if (!rseq.empty()) {                                                  // graph.h:406-477
                Ori v;
                in.right_of(s.i, s.f, v.i, v.f);
                while (true) {
                    rstop = v;
                    if (flag_count(in.lflags(v.i, v.f)) > 1 || visited[v.i]) {
                        rfl = 1u << rseq.back();
                        rseq.pop_back();
                        break;
                    }
                    const unsigned vr = in.rflags(v.i, v.f);
                    if (vr == 0) {
                        rfl = 0;
                        sum += in.counts[v.i];
                        visited[v.i] = 1;
                        break;
                    } else if (flag_count(vr) > 1) {
                        rfl = 1u << rseq.back();
                        rseq.pop_back();
                        break;
                    }
                    rfl = vr;
                    rseq.push_back((unsigned char)flag2base(vr));
                    sum += in.counts[v.i];
                    visited[v.i] = 1;
                    Ori nx;
                    in.right_of(v.i, v.f, nx.i, nx.f);
                    v = nx;
                }
            }
            if (!lseq.empty()) {                                                  // graph.h:480-542
                Ori v;
                in.left_of(s.i, s.f, v.i, v.f);
                while (true) {
                    lstop = v;
                    if (flag_count(in.rflags(v.i, v.f)) > 1 || visited[v.i]) {
                        lfl = 1u << lseq.back();
                        lseq.pop_back();
                        break;
                    }
                    const unsigned vl = in.lflags(v.i, v.f);
                    if (vl == 0) {
                        lfl = 0;
                        sum += in.counts[v.i];
                        visited[v.i] = 1;
                        break;
                    } else if (flag_count(vl) > 1) {
                        lfl = 1u << lseq.back();
                        lseq.pop_back();
                        break;
                    }
                    lfl = vl;
                    lseq.push_back((unsigned char)flag2base(vl));
                    sum += in.counts[v.i];
                    visited[v.i] = 1;
                    Ori nx;
                    in.left_of(v.i, v.f, nx.i, nx.f);
                    v = nx;
                }
            }
            // graph.h:545-555: left stopper first, so the right one is expanded first
            if (lfl != 0) stack.push_back(lstop);
            if (rfl != 0) stack.push_back(rstop);

            Straight st;                                                          // graph.h:557-575
            st.length = lseq.size() + rseq.size() + 1;
            st.seq.assign((st.length + k + 31) / 32, 0);
            const u64 rn = rseq.size();
            for (u64 j = 0; j < rn; ++j) st.set(rn - 1 - j, rseq[j]);
            for (unsigned j = 0; j < k; ++j) st.set(rn + j, start.get(j));
            for (u64 j = 0; j < lseq.size(); ++j) st.set(rn + k + j, lseq[j]);
            st.coverage = (u16)(static_cast<double>(sum) / static_cast<double>(st.length) + 0.5);
            st.out = (uint8_t)((lfl << 4) | rfl);
            gr.insert_straight(std::move(st));
        }
    }
}


// ---- chain-level replay --------------------------------------------------------------------------------------------
// With pb_graph_chains the device has already walked every simple chain (list ranking), summed its counts and gathered
// its sequence, so the replay touches one record per node instead of one per k-mer.  A simple chain is consumed whole by
// whichever of its k-mers the DFS reaches first (nothing inside it can be "visited" earlier, and both of its ends stop at
// a junction or at nothing), so "visited" is kept per chain; only the strand depends on the entry k-mer.  K-mers on
// cycles / hairpins / self loops (chain_of == 0xFFFFFFFE) are still walked one by one exactly as above.
struct ChainSet {
    u64 n_chains = 0;
    const uint32_t *chain_of = nullptr;   // n
    const uint8_t *orient = nullptr;      // n
    const uint32_t *len = nullptr;        // per chain ...
    const u64 *sum = nullptr;
    const uint32_t *end_a = nullptr, *end_b = nullptr;
    const u64 *seq_woff = nullptr;
    const u64 *seq = nullptr;
};
static const uint32_t CHAIN_JUNCTION = 0xFFFFFFFFu, CHAIN_COMPLEX = 0xFFFFFFFEu;

// reverse complement of a packed 2-bit string of `len` symbols (position 0 in the low bits of word 0)
inline void revcomp_packed(const u64 *src, u64 len, std::vector<u64> &dst)
{
    const u64 nw = (len + 31) / 32;
    dst.assign(nw, 0);
    // symbol p of dst = 3 - symbol (len-1-p) of src: reverse whole words, then shift out the padding
    std::vector<u64> tmp(nw + 1, 0);
    for (u64 i = 0; i < nw; ++i) tmp[i] = revcom_u64(src[nw - 1 - i]);     // now the string sits in the TOP of the nw-word block
    const unsigned pad = (unsigned)((nw * 32 - len) * 2);                   // bits to shift down
    if (pad == 0) { for (u64 i = 0; i < nw; ++i) dst[i] = tmp[i]; }
    else { for (u64 i = 0; i < nw; ++i) dst[i] = (tmp[i] >> pad) | (tmp[i + 1] << (64 - pad)); }
    // revcom_u64 complemented the padding bits of the source's top word too; they are shifted out above
    if (len & 31) dst[nw - 1] &= (1ull << (2 * (len & 31))) - 1;
}

template <int W>
void make_initial_from_chains(Graph<W> &gr, const DeviceGraphInput<W> &in, const ChainSet &cs)
{
    typedef Key<W> K;
    const unsigned k = in.k;
    gr.junctions.clear();
    gr.left.clear();
    gr.right.clear();
    std::vector<uint8_t> visited(in.n, 0);              // junction / complex k-mers
    std::vector<uint8_t> chain_done(cs.n_chains, 0);
    auto is_visited = [&](u64 i) -> bool {
        const uint32_t c = cs.chain_of[i];
        return c < CHAIN_COMPLEX ? chain_done[c] != 0 : visited[i] != 0;
    };
    struct Ori { u64 i; unsigned f; };
    std::vector<Ori> stack;
    std::vector<unsigned char> lseq, rseq;
    std::vector<u64> rc_buf;
    for (u64 root = 0; root < in.n; ++root) {
        if (is_visited(root)) continue;
        stack.clear();
        stack.push_back({root, 0u});
        while (!stack.empty()) {
            const Ori s = stack.back();
            stack.pop_back();
            if (is_visited(s.i)) continue;
            const uint32_t c = cs.chain_of[s.i];
            if (c < CHAIN_COMPLEX) {
                // ---- a whole simple chain becomes one Straight ----
                chain_done[c] = 1;
                const bool aligned = cs.orient[s.i] == s.f;          // entry k-mer points along A -> B
                const u64 len = cs.len[c], ls = len + k - 1;
                Ori ea{cs.end_a[c] >> 1, cs.end_a[c] & 1u}, eb{cs.end_b[c] >> 1, cs.end_b[c] & 1u};
                const Ori le = aligned ? ea : Ori{eb.i, eb.f ^ 1u};
                const Ori re = aligned ? eb : Ori{ea.i, ea.f ^ 1u};
                const unsigned lfl = in.lflags(le.i, le.f), rfl = in.rflags(re.i, re.f);
                if (lfl != 0) { Ori v; in.left_of(le.i, le.f, v.i, v.f); stack.push_back(v); }
                if (rfl != 0) { Ori v; in.right_of(re.i, re.f, v.i, v.f); stack.push_back(v); }
                Straight st;
                st.length = len;
                const u64 nw = (ls + 31) / 32;
                const u64 *src = cs.seq + cs.seq_woff[c];
                if (aligned) st.seq.assign(src, src + nw);
                else { revcomp_packed(src, ls, rc_buf); st.seq = rc_buf; }
                st.seq.resize((len + k + 31) / 32, 0);                // binstr_t(length + k) word count (graph.h:559)
                st.coverage = (u16)(static_cast<double>(cs.sum[c]) / static_cast<double>(len) + 0.5);
                st.out = (uint8_t)((lfl << 4) | rfl);
                gr.insert_straight(std::move(st));
                continue;
            }
            visited[s.i] = 1;
            u64 sum = in.counts[s.i];
            unsigned lfl = in.lflags(s.i, s.f), rfl = in.rflags(s.i, s.f);
            const K start = in.oriented(s.i, s.f);
            if (c == CHAIN_JUNCTION) {                                            // graph.h:383-404
                for (unsigned b = 0; b < 4; ++b) {
                    Ori o;
                    if (rfl & (1u << b)) { in.locate(gr.succ(start, b), o.i, o.f); stack.push_back(o); }
                    if (lfl & (1u << b)) { in.locate(gr.pred(start, b), o.i, o.f); stack.push_back(o); }
                }
                Junction j;
                j.coverage = (u16)sum;
                j.out = (uint8_t)((lfl << 4) | rfl);
                gr.junctions[start] = j;
                continue;
            }
            // ---- k-mer on a cycle / hairpin / self loop: the reference's walk, k-mer by k-mer ----
            lseq.clear();
            rseq.clear();
            if (lfl) lseq.push_back((unsigned char)flag2base(lfl));
            if (rfl) rseq.push_back((unsigned char)flag2base(rfl));
            Ori rstop = s, lstop = s;
            if (!rseq.empty()) {
                Ori v;
                in.right_of(s.i, s.f, v.i, v.f);
                while (true) {
                    rstop = v;
                    if (flag_count(in.lflags(v.i, v.f)) > 1 || is_visited(v.i)) { rfl = 1u << rseq.back(); rseq.pop_back(); break; }
                    const unsigned vr = in.rflags(v.i, v.f);
                    if (vr == 0) { rfl = 0; sum += in.counts[v.i]; visited[v.i] = 1; break; }
                    else if (flag_count(vr) > 1) { rfl = 1u << rseq.back(); rseq.pop_back(); break; }
                    rfl = vr;
                    rseq.push_back((unsigned char)flag2base(vr));
                    sum += in.counts[v.i];
                    visited[v.i] = 1;
                    Ori nx;
                    in.right_of(v.i, v.f, nx.i, nx.f);
                    v = nx;
                }
            }
            if (!lseq.empty()) {
                Ori v;
                in.left_of(s.i, s.f, v.i, v.f);
                while (true) {
                    lstop = v;
                    if (flag_count(in.rflags(v.i, v.f)) > 1 || is_visited(v.i)) { lfl = 1u << lseq.back(); lseq.pop_back(); break; }
                    const unsigned vl = in.lflags(v.i, v.f);
                    if (vl == 0) { lfl = 0; sum += in.counts[v.i]; visited[v.i] = 1; break; }
                    else if (flag_count(vl) > 1) { lfl = 1u << lseq.back(); lseq.pop_back(); break; }
                    lfl = vl;
                    lseq.push_back((unsigned char)flag2base(vl));
                    sum += in.counts[v.i];
                    visited[v.i] = 1;
                    Ori nx;
                    in.left_of(v.i, v.f, nx.i, nx.f);
                    v = nx;
                }
            }
            if (lfl != 0) stack.push_back(lstop);
            if (rfl != 0) stack.push_back(rstop);
            Straight st;
            st.length = lseq.size() + rseq.size() + 1;
            st.seq.assign((st.length + k + 31) / 32, 0);
            const u64 rn = rseq.size();
            for (u64 j = 0; j < rn; ++j) st.set(rn - 1 - j, rseq[j]);
            for (unsigned j = 0; j < k; ++j) st.set(rn + j, start.get(j));
            for (u64 j = 0; j < lseq.size(); ++j) st.set(rn + k + j, lseq[j]);
            st.coverage = (u16)(static_cast<double>(sum) / static_cast<double>(st.length) + 0.5);
            st.out = (uint8_t)((lfl << 4) | rfl);
            gr.insert_straight(std::move(st));
        }
    }
}

} // namespace pbh

// ---- replay from the compact hand-over (pb_graph_nodes / pb_graph_fetch_chain_records) -----------------------------------
// Same DFS as make_initial_from_chains, but everything it reads comes from two small tables: the "nodes" (junction k-mers,
// k-mers on cycles / hairpins, and the end k-mers of the simple chains -- the only k-mers the DFS can land on) and the
// per-chain records.  The root loop of the reference (ascending key order over all k-mers, graph.h:330-341) becomes a merge
// of the nodes with the chains' smallest members (root[c]): an interior k-mer of a chain is only ever a root through that.
namespace pbh {

[[noreturn]] inline void handover_fail(const std::string &what)
{
    throw std::runtime_error("device graph hand-over is inconsistent: " + what);
}

template <int W>
struct NodeSet {
    unsigned k = 0;
    std::vector<uint32_t> idx, meta, chain, nbr, nb8;   // as fetched; sort() orders everything by idx
    std::vector<u64> keys;
    u64 size() const { return idx.size(); }

    void sort()
    {
        const u64 n = idx.size();
        std::vector<uint32_t> order(n);
        for (u64 i = 0; i < n; ++i) order[i] = (uint32_t)i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return idx[a] < idx[b]; });
        std::vector<uint32_t> i2(n), m2(n), c2(n), n2(2 * n), b2(8 * n);
        std::vector<u64> k2(n * W);
        for (u64 s = 0; s < n; ++s) {
            const u64 o = order[s];
            i2[s] = idx[o]; m2[s] = meta[o]; c2[s] = chain[o];
            n2[2 * s] = nbr[2 * o]; n2[2 * s + 1] = nbr[2 * o + 1];
            for (int b = 0; b < 8; ++b) b2[8 * s + b] = nb8[8 * o + b];
            for (int w = 0; w < W; ++w) k2[s * W + w] = keys[o * W + w];
        }
        idx.swap(i2); meta.swap(m2); chain.swap(c2); nbr.swap(n2); nb8.swap(b2); keys.swap(k2);
        // first node of every block of 2^SLOT_SHIFT k-mer indices: slot_of looks at the one or two nodes of a block instead of taking
        // twenty steps of a binary search through a table that does not fit the caches (r3: a tenth of the replay's time at 100 Mbp)
        const u64 blocks = n ? ((u64)idx[n - 1] >> SLOT_SHIFT) + 2 : 1;
        block_start.assign(blocks, (uint32_t)n);
        for (u64 s = n; s-- > 0;) block_start[idx[s] >> SLOT_SHIFT] = (uint32_t)s;
        for (u64 b = blocks - 1; b-- > 0;) if (block_start[b] == (uint32_t)n || block_start[b] > block_start[b + 1]) block_start[b] = block_start[b + 1];
    }
    static constexpr unsigned SLOT_SHIFT = 6;
    std::vector<uint32_t> block_start;
    // i is a node by construction; a value from the device that is none ends the run with a message instead of an out-of-range access
    u64 slot_of(u64 i) const
    {
        const u64 b = i >> SLOT_SHIFT;
        if (b + 1 < block_start.size())
            for (u64 s = block_start[b], e = block_start[b + 1]; s < e; ++s) if (idx[s] == i) return s;
        handover_fail("k-mer index " + std::to_string(i) + " is not among the " + std::to_string(idx.size()) + " exported nodes");
    }
    unsigned adj(u64 s) const { return meta[s] & 0xFFu; }
    unsigned orient(u64 s) const { return (meta[s] >> 8) & 1u; }
    unsigned count(u64 s) const { return meta[s] >> 16; }
    Key<W> oriented(u64 s, unsigned f) const
    {
        Key<W> c;
        for (int w = 0; w < W; ++w) c.w[w] = keys[s * W + w];
        return f ? revcomp(c, k) : c;
    }
    unsigned rflags(u64 s, unsigned f) const { return f ? rev4(adj(s) >> 4) : (adj(s) & 15u); }
    unsigned lflags(u64 s, unsigned f) const { return f ? rev4(adj(s) & 15u) : (adj(s) >> 4); }
    void right_of(u64 s, unsigned f, u64 &j, unsigned &g) const { const uint32_t v = f ? nbr[2 * s + 1] : nbr[2 * s]; j = v >> 1; g = (v & 1u) ^ f; }
    void left_of(u64 s, unsigned f, u64 &j, unsigned &g) const { const uint32_t v = f ? nbr[2 * s] : nbr[2 * s + 1]; j = v >> 1; g = (v & 1u) ^ f; }
    // junctions: successor / predecessor with base b of the oriented k-mer (s,f).  succ(rc(K), b) = rc(pred(K, 3-b)).
    void succ_of(u64 s, unsigned f, unsigned b, u64 &j, unsigned &g) const
    {
        const uint32_t e = f ? nb8[8 * s + 4 + (3 - b)] : nb8[8 * s + b];
        j = e >> 1; g = (e & 1u) ^ f;
    }
    void pred_of(u64 s, unsigned f, unsigned b, u64 &j, unsigned &g) const
    {
        const uint32_t e = f ? nb8[8 * s + (3 - b)] : nb8[8 * s + 4 + b];
        j = e >> 1; g = (e & 1u) ^ f;
    }
};

struct ChainRecords {
    std::vector<uint32_t> len, end_a, end_b, root;
    std::vector<u64> sum, seq_woff, seq;
    u64 size() const { return len.size(); }
};

template <int W>
void make_initial_from_nodes(Graph<W> &gr, NodeSet<W> &nd, const ChainRecords &cs)
{
    typedef Key<W> K;
    const unsigned k = nd.k;
    gr.junctions.clear();
    gr.left.clear();
    gr.right.clear();
    nd.sort();
    const u64 nn = nd.size(), nc = cs.size();
    // everything the replay indexes with comes from the device: checked once here (chain ids, sequence ranges, node order) and in
    // slot_of (k-mer indices), so that a wrong word becomes "Error: ..." with the offending record, never a wild host access
    for (u64 s = 0; s < nn; ++s) {
        if (s && nd.idx[s] <= nd.idx[s - 1]) handover_fail("node " + std::to_string(s) + " repeats k-mer index " + std::to_string(nd.idx[s]));
        if (nd.chain[s] < CHAIN_COMPLEX && nd.chain[s] >= nc)
            handover_fail("node " + std::to_string(s) + " (k-mer " + std::to_string(nd.idx[s]) + ") names chain " + std::to_string(nd.chain[s]) + " of " + std::to_string(nc));
    }
    for (u64 c = 0; c < nc; ++c) {
        const u64 nw = ((u64)cs.len[c] + k - 1 + 31) / 32;
        if (cs.len[c] == 0 || cs.seq_woff[c] > cs.seq.size() || nw > cs.seq.size() - cs.seq_woff[c])
            handover_fail("chain " + std::to_string(c) + ": length " + std::to_string(cs.len[c]) + ", sequence words [" + std::to_string(cs.seq_woff[c]) + ", +" +
                          std::to_string(nw) + ") of " + std::to_string(cs.seq.size()));
    }
    std::vector<uint8_t> visited(nn, 0), chain_done(nc, 0);
    std::vector<uint32_t> by_root(nc);
    for (u64 c = 0; c < nc; ++c) by_root[c] = (uint32_t)c;
    std::sort(by_root.begin(), by_root.end(), [&](uint32_t a, uint32_t b) { return cs.root[a] < cs.root[b]; });
    struct Ori { u64 i; unsigned f; };
    std::vector<Ori> stack;
    std::vector<unsigned char> lseq, rseq;
    std::vector<u64> rc_buf;
    auto is_visited_slot = [&](u64 s) -> bool {
        const uint32_t c = nd.chain[s];
        return c < CHAIN_COMPLEX ? chain_done[c] != 0 : visited[s] != 0;
    };
    // a whole simple chain becomes one Straight; `aligned`: it is entered pointing along A -> B
    auto take_chain = [&](uint32_t c, bool aligned) {
        chain_done[c] = 1;
        const u64 len = cs.len[c], ls = len + k - 1;
        const Ori ea{cs.end_a[c] >> 1, cs.end_a[c] & 1u}, eb{cs.end_b[c] >> 1, cs.end_b[c] & 1u};
        const Ori le = aligned ? ea : Ori{eb.i, eb.f ^ 1u};
        const Ori re = aligned ? eb : Ori{ea.i, ea.f ^ 1u};
        const u64 sl = nd.slot_of(le.i), sr = nd.slot_of(re.i);
        const unsigned lfl = nd.lflags(sl, le.f), rfl = nd.rflags(sr, re.f);
        if (lfl != 0) { Ori v; nd.left_of(sl, le.f, v.i, v.f); stack.push_back(v); }
        if (rfl != 0) { Ori v; nd.right_of(sr, re.f, v.i, v.f); stack.push_back(v); }
        Straight st;
        st.length = len;
        const u64 nw = (ls + 31) / 32;
        const u64 *src = cs.seq.data() + cs.seq_woff[c];
        if (aligned) st.seq.assign(src, src + nw);
        else { revcomp_packed(src, ls, rc_buf); st.seq = rc_buf; }
        st.seq.resize((len + k + 31) / 32, 0);                // binstr_t(length + k) word count (graph.h:559)
        st.coverage = (u16)(static_cast<double>(cs.sum[c]) / static_cast<double>(len) + 0.5);
        st.out = (uint8_t)((lfl << 4) | rfl);
        gr.insert_straight(std::move(st));
    };
    auto run_stack = [&]() {
        while (!stack.empty()) {
            const Ori s = stack.back();
            stack.pop_back();
            const u64 ss = nd.slot_of(s.i);
            if (is_visited_slot(ss)) continue;
            const uint32_t c = nd.chain[ss];
            if (c < CHAIN_COMPLEX) { take_chain(c, nd.orient(ss) == s.f); continue; }
            visited[ss] = 1;
            u64 sum = nd.count(ss);
            unsigned lfl = nd.lflags(ss, s.f), rfl = nd.rflags(ss, s.f);
            const K start = nd.oriented(ss, s.f);
            if (c == CHAIN_JUNCTION) {                                            // graph.h:383-404
                for (unsigned b = 0; b < 4; ++b) {
                    Ori o;
                    if (rfl & (1u << b)) { nd.succ_of(ss, s.f, b, o.i, o.f); stack.push_back(o); }
                    if (lfl & (1u << b)) { nd.pred_of(ss, s.f, b, o.i, o.f); stack.push_back(o); }
                }
                Junction j;
                j.coverage = (u16)sum;
                j.out = (uint8_t)((lfl << 4) | rfl);
                gr.junctions[start] = j;
                continue;
            }
            // ---- k-mer on a cycle / hairpin / self loop: the reference's walk, k-mer by k-mer ----
            lseq.clear();
            rseq.clear();
            if (lfl) lseq.push_back((unsigned char)flag2base(lfl));
            if (rfl) rseq.push_back((unsigned char)flag2base(rfl));
            Ori rstop = s, lstop = s;
            if (!rseq.empty()) {
                Ori v;
                nd.right_of(ss, s.f, v.i, v.f);
                while (true) {
                    rstop = v;
                    const u64 sv = nd.slot_of(v.i);
                    if (flag_count(nd.lflags(sv, v.f)) > 1 || is_visited_slot(sv)) { rfl = 1u << rseq.back(); rseq.pop_back(); break; }
                    const unsigned vr = nd.rflags(sv, v.f);
                    if (vr == 0) { rfl = 0; sum += nd.count(sv); visited[sv] = 1; break; }
                    else if (flag_count(vr) > 1) { rfl = 1u << rseq.back(); rseq.pop_back(); break; }
                    rfl = vr;
                    rseq.push_back((unsigned char)flag2base(vr));
                    sum += nd.count(sv);
                    visited[sv] = 1;
                    Ori nx;
                    nd.right_of(sv, v.f, nx.i, nx.f);
                    v = nx;
                }
            }
            if (!lseq.empty()) {
                Ori v;
                nd.left_of(ss, s.f, v.i, v.f);
                while (true) {
                    lstop = v;
                    const u64 sv = nd.slot_of(v.i);
                    if (flag_count(nd.rflags(sv, v.f)) > 1 || is_visited_slot(sv)) { lfl = 1u << lseq.back(); lseq.pop_back(); break; }
                    const unsigned vl = nd.lflags(sv, v.f);
                    if (vl == 0) { lfl = 0; sum += nd.count(sv); visited[sv] = 1; break; }
                    else if (flag_count(vl) > 1) { lfl = 1u << lseq.back(); lseq.pop_back(); break; }
                    lfl = vl;
                    lseq.push_back((unsigned char)flag2base(vl));
                    sum += nd.count(sv);
                    visited[sv] = 1;
                    Ori nx;
                    nd.left_of(sv, v.f, nx.i, nx.f);
                    v = nx;
                }
            }
            if (lfl != 0) stack.push_back(lstop);
            if (rfl != 0) stack.push_back(rstop);
            Straight st;
            st.length = lseq.size() + rseq.size() + 1;
            st.seq.assign((st.length + k + 31) / 32, 0);
            const u64 rn = rseq.size();
            for (u64 j = 0; j < rn; ++j) st.set(rn - 1 - j, rseq[j]);
            for (unsigned j = 0; j < k; ++j) st.set(rn + j, start.get(j));
            for (u64 j = 0; j < lseq.size(); ++j) st.set(rn + k + j, lseq[j]);
            st.coverage = (u16)(static_cast<double>(sum) / static_cast<double>(st.length) + 0.5);
            st.out = (uint8_t)((lfl << 4) | rfl);
            gr.insert_straight(std::move(st));
        }
    };
    u64 pn = 0, pc = 0;
    while (true) {
        while (pn < nn && nd.chain[pn] < CHAIN_COMPLEX) ++pn;             // chain ends are reached through their chain's root
        while (pc < nc && chain_done[by_root[pc]]) ++pc;
        const bool have_n = pn < nn, have_c = pc < nc;
        if (!have_n && !have_c) break;
        const u64 in_ = have_n ? nd.idx[pn] : ~0ull, ic = have_c ? (u64)(cs.root[by_root[pc]] >> 1) : ~0ull;
        stack.clear();
        if (in_ < ic) {
            if (!visited[pn]) { stack.push_back({(u64)nd.idx[pn], 0u}); run_stack(); }
            ++pn;
        } else {
            const uint32_t c = by_root[pc];
            take_chain(c, (cs.root[c] & 1u) == 0);                         // root k-mer in its canonical orientation (graph.h:334)
            run_stack();
            ++pc;
        }
    }
}

} // namespace pbh
