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
            gr.insert_straight(st);
        }
    }
}

} // namespace pbh
