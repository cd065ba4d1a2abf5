// Host side of the de Bruijn graph stage: node tables and node-level cleaning of the reference's
// BruijnGraph<KMER> (graph.h:37-2431), restated -- hot-path functions only: makeInitialBruijnGraph, cutBranch, concatinateNodes (+ helpers), crushBubble,
// deleteErroneousStraightNode, saveEdgeKmer, saveContig<NEXT>, saveContigSimple, saveJunction,
// getLeftMinimalCoverage, getAverageCoverageExcludingBubble, and platanus::printContig (common.h:313-345).
//
// This header is product code (the C++ host of `platanus_b assemble` keeps these ~10^4-node steps on the CPU, the
// k-mer-level work runs on the GPU behind include/pbgpu.h); the oracle under oracle/ reuses it for the same steps
// and adds its own k-mer-level table.
//
// Node tables are real std::unordered_map objects with the reference's hashers and the reference's
// insertion sequence, because contig numbering in PREFIX_contig.fa is their iteration order and
// clear() keeps the bucket array across rounds.
#pragma once
#include "kmer.hpp"
#include "errors.hpp"
#include <sstream>
#include <algorithm>
#include <array>
#include <cstring>
#include <cstdio>
#include <iostream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <functional>
#include <map>
#include <thread>
#include <exception>
#include <vector>

namespace pbh {

inline unsigned flag2base(unsigned f)                 // common.h:300-304
{
    static const unsigned a[16] = {4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4};
    return a[f & 15];
}
inline unsigned flag_count(unsigned f) { return (f & 1) + ((f >> 1) & 1) + ((f >> 2) & 1) + ((f >> 3) & 1); }

// graph.h:53-91.  seq position p counts bases from the 3' (right) end.
struct Straight {
    u64 length = 0;               // number of k-mers; sequence length = length + k - 1
    u16 coverage = 0;
    uint8_t out = 0;              // (left one-hot << 4) | right one-hot
    mutable uint32_t export_id = 0;   // clean_pass_device: 1 + index of the node in the arrays it sends to the device (0: not sent); no meaning elsewhere
    std::vector<u64> seq;
    unsigned get(u64 pos) const { return (unsigned)((seq[pos / 32] >> ((pos % 32) * 2)) & 3); }
    void set(u64 pos, unsigned v)
    {
        seq[pos / 32] = (seq[pos / 32] & ~(3ull << ((pos % 32) * 2))) | ((u64)v << ((pos % 32) * 2));
    }
    void resize(unsigned k) { seq.resize((length + k + 30) / 32); }
    // this[at + p] = src[p] for p in [0, n): word-wise (the target positions must still be zero, bits of src above n are ignored)
    void put_bases(u64 at, const std::vector<u64> &src, u64 n)
    {
        const u64 ws = at / 32, bs = (at % 32) * 2, nw = (n + 31) / 32;
        for (u64 i = 0; i < nw; ++i) {
            u64 v = i < src.size() ? src[i] : 0;
            if (i == nw - 1 && (n % 32)) v &= ~0ull >> (64 - 2 * (n % 32));
            if (v == 0) continue;
            seq[ws + i] |= v << bs;
            if (bs && ws + i + 1 < seq.size()) seq[ws + i + 1] |= v >> (64 - bs);
        }
    }
    // bases n-1 .. 0 (left to right) appended to str: four bases per table look-up instead of one get() per base
    void append_bases(std::vector<unsigned char> &str, u64 n) const
    {
        static const std::array<uint32_t, 256> lut = [] {
            std::array<uint32_t, 256> t{};
            for (unsigned b = 0; b < 256; ++b) t[b] = ((b >> 6) & 3u) | (((b >> 4) & 3u) << 8) | (((b >> 2) & 3u) << 16) | ((b & 3u) << 24);
            return t;
        }();
        const size_t at = str.size();
        str.resize(at + n);
        unsigned char *o = str.data() + at;
        u64 j = 0;
        for (; j + 4 <= n; j += 4) {                       // bases j .. j+3 land at o[n-4-j .. n-1-j], highest first
            const uint32_t v = lut[(seq[j >> 5] >> ((j & 31) * 2)) & 0xff];
            std::memcpy(o + (n - 4 - j), &v, 4);
        }
        for (; j < n; ++j) o[n - 1 - j] = (unsigned char)get(j);
    }
};
struct Junction {                 // graph.h:99-106
    u16 coverage = 0;
    uint8_t out = 0;
};
struct ContigRecord {             // one record of contigFP / junctionFP (graph.h:1774-1823)
    u64 length;
    u16 cov;
    std::vector<u64> seq;
};

// common.h:313-345
inline void print_contigs(const std::string &path, const std::vector<ContigRecord> &recs, double cov_ratio,
                          double avg_len, long max_k, const char *prefix, bool append)
{
    FILE *f = fopen(path.c_str(), append ? "a" : "w");
    if (!f) throw file_error(path);
    u64 num = 0;
    std::string line;
    for (const ContigRecord &r : recs) {
        if (r.cov == 65535) continue;
        ++num;
        char head[256];
        // operator<<(double) prints with default precision 6 -> %g
        snprintf(head, sizeof head, ">%s%llu_len%llu_cov%u_read%g_maxK%ld\n", prefix, (unsigned long long)num,
                 (unsigned long long)r.length, (unsigned)static_cast<u16>(r.cov * cov_ratio + 0.5), avg_len, max_k);
        fputs(head, f);
        // 80 bases per line, built in memory and written once (a 5 Mbp contig is 5 M fputc calls otherwise)
        line.clear();
        line.reserve(r.length + r.length / 80 + 2);
        u64 j = 0;
        for (; j < r.length; ++j) {
            const u64 p = r.length - 1 - j;
            line.push_back(bin2char((unsigned)((r.seq[p / 32] >> ((p % 32) * 2)) & 3)));
            if ((j + 1) % 80 == 0) line.push_back('\n');
        }
        if (j % 80 != 0) line.push_back('\n');
        fwrite(line.data(), 1, line.size(), f);
    }
    fclose(f);
}

// graph.h:1586-1612 printBubble
inline void print_bubbles(const std::string &path, const std::vector<ContigRecord> &recs, double ratio)
{
    FILE *f = fopen(path.c_str(), "w");
    if (!f) throw file_error(path);
    u64 num = 0;
    for (const ContigRecord &r : recs) {
        const u16 occ = r.cov * ratio + 0.5;
        ++num;
        fprintf(f, ">seq%llu_len%llu_cov%u\n", (unsigned long long)num, (unsigned long long)r.length, (unsigned)occ);
        u64 j = 0;
        for (; j < r.length; ++j) {
            const u64 p = r.length - 1 - j;
            fputc(bin2char((unsigned)((r.seq[p / 32] >> ((p % 32) * 2)) & 3)), f);
            if ((j + 1) % 80 == 0) fputc('\n', f);
        }
        if (j % 80 != 0) fputc('\n', f);
    }
    fclose(f);
}

template <int W>
struct Graph {
    typedef Key<W> K;
    unsigned k = 0;
    double bubble_th = 0, branch_th = 0;
    std::unordered_map<K, Junction, KeyHash<W>> junctions;       // junctionTable
    std::unordered_map<K, Straight, KeyHash<W>> left;            // leftStraightTable (key = head k-mer)
    std::unordered_map<K, Straight *, KeyHash<W>> right;         // rightStraightTable (key = tail k-mer)
    std::vector<ContigRecord> bubbles;                           // bubbleFP of the last crush_bubble_iterative (graph.h:661-668)

    // Device hooks (set by the GPU build's driver; empty in the oracle, which then runs the reference's sequential loops below).
    // clean_decide: include/pbgpu.h pb_nodes_clean_decide on the live nodes; edit_distances: pb_edit_distances.  Both return false
    // when the device call was not made (the host loop takes over).
    std::function<bool(unsigned k, u64 n_s, const u64 *head, const u64 *tail, const u64 *length, const u16 *cov, const uint8_t *out, u64 n_t, const u64 *tkeys,
                       const uint32_t *tids, u64 n_j, const u64 *jkeys, const u16 *jcov, const uint8_t *jout, double branch_th, u64 length_cut, u64 coverage_cut,
                       uint8_t *decision)> clean_decide;
    std::function<bool(const uint8_t *bases, const u64 *offsets, u64 n_pairs, u64 *dist)> edit_distances;

    void set_k(unsigned kk) { k = kk; }
    void set_thresholds(double bubble, double branch) { bubble_th = bubble; branch_th = branch; }
    void delete_all() { junctions.clear(); left.clear(); right.clear(); }     // graph.h:222-227

    // ---- k-mer arithmetic on strand-specific keys ----
    K succ(const K &s, unsigned b) const { K r = s; r.shl2(k); r.set(0, b); return r; }
    K pred(const K &s, unsigned b) const { K r = s; r.shr2(); r.set(k - 1, b); return r; }
    K canon(const K &s) const { K r = revcomp(s, k); return r < s ? r : s; }
    // the k bases at positions at .. at + k - 1 of a node's sequence as a key (key position j = sequence position at + j): word shifts, not k
    // get / set pairs (a key and a sequence have the same bit layout; r3: the node tables of a 100 Mbp genome ask for millions of these per pass)
    K key_at(const Straight &s, u64 at) const
    {
        K r;
        const u64 ws = at >> 5, sz = s.seq.size();
        const unsigned bs = (unsigned)(at & 31) * 2;
        const int n = words_for_k(k);
        for (int i = 0; i < n; ++i) {
            const u64 lo = ws + i < sz ? s.seq[ws + i] : 0, hi = (bs && ws + i + 1 < sz) ? s.seq[ws + i + 1] : 0;
            r.w[i] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
        }
        r.mask(k);
        return r;
    }
    K head_of(const Straight &s) const { return key_at(s, s.length - 1); }
    K tail_of(const Straight &s) const { return key_at(s, 0); }
    // the k-mer that follows a node's tail on base b: bases 0 .. k-2 of the node at key positions 1 .. k-1, b at position 0
    K right_of_tail(const Straight &s, unsigned b) const { K r = key_at(s, 0); r.shl2(k); r.set(0, b); return r; }

    Straight *find_head(const K &key)                  // graph.h:194-201
    {
        auto it = left.find(key);
        return (it == left.end() || it->second.coverage == 65535) ? nullptr : &it->second;
    }
    Straight *find_tail(const K &key)                  // graph.h:203-210
    {
        auto it = right.find(key);
        return (it == right.end() || it->second->coverage == 65535) ? nullptr : it->second;
    }
    Junction *find_junction(const K &key)              // graph.h:212-219
    {
        auto it = junctions.find(key);
        return (it == junctions.end() || it->second.coverage == 65535) ? nullptr : &it->second;
    }
    void delete_straight(const Straight &s) { left[head_of(s)].coverage = 65535; }      // graph.h:769-778
    void delete_junction(const K &key) { junctions[key].coverage = 65535; }             // graph.h:786-789
    void insert_straight(const Straight &s)                                             // graph.h:809-821
    {
        const K h = head_of(s), t = tail_of(s);
        Straight &dst = left[h];
        dst = s;
        right[t] = &dst;
    }
    void insert_straight(Straight &&s)                                                  // (the same insertions in the same order; the sequence is moved, not copied)
    {
        const K h = head_of(s), t = tail_of(s);
        Straight &dst = left[h];
        dst = std::move(s);
        right[t] = &dst;
    }

    u64 estimate_kmers_on_straight() const              // graph.h:245-254
    {
        u64 n = 0;
        for (auto &e : left) if (e.second.coverage != 65535) n += e.second.length;
        return n;
    }

    // graph.h:289-579.  `table` is the counter's occurrence table; visited k-mers are overwritten with 65535.
    template <typename TABLE>
    void make_initial(TABLE &table, const std::vector<K> &sorted)
    {
        junctions.clear();
        left.clear();
        right.clear();
        std::vector<K> stack;
        std::vector<unsigned char> lseq, rseq;
        for (const K &root : sorted) {
            stack.clear();
            stack.push_back(root);
            while (!stack.empty()) {
                const K start = stack.back();
                stack.pop_back();
                const u64 slot = table.find_any(canon(start));
                const u16 val = table.vals[slot];
                table.vals[slot] = 65535;
                if (val == 65535) continue;
                u64 sum = val;

                unsigned lflags = 0, rflags = 0;
                for (unsigned b = 0; b < 4; ++b) if (table.value(canon(pred(start, b))) != 0) lflags |= 1u << b;
                for (unsigned b = 0; b < 4; ++b) if (table.value(canon(succ(start, b))) != 0) rflags |= 1u << b;
                lseq.clear();
                rseq.clear();
                if (lflags) lseq.push_back(flag2base(lflags));
                if (rflags) rseq.push_back(flag2base(rflags));

                if ((!lseq.empty() && lseq[0] == 4) || (!rseq.empty() && rseq[0] == 4)) {
                    for (unsigned b = 0; b < 4; ++b) {                  // graph.h:386-397
                        if (rflags & (1u << b)) stack.push_back(succ(start, b));
                        if (lflags & (1u << b)) stack.push_back(pred(start, b));
                    }
                    Junction j;
                    j.coverage = (u16)sum;
                    j.out = (uint8_t)((lflags << 4) | rflags);
                    junctions[start] = j;
                    continue;
                }

                K rstop = start, lstop = start;
                if (!rseq.empty()) {                                    // graph.h:406-477
                    K v = succ(start, rseq[0]);
                    u64 vslot = table.find_any(canon(v));
                    while (true) {
                        rstop = v;
                        unsigned in = 0;
                        K vp = v; vp.shr2();
                        for (unsigned b = 0; b < 4; ++b) { K p = vp; p.set(k - 1, b); if (table.value(canon(p)) != 0) in |= 1u << b; }
                        if (flag_count(in) > 1 || table.vals[vslot] == 65535) {
                            rflags = 1u << rseq.back();
                            rseq.pop_back();
                            break;
                        }
                        rflags = 0;
                        u64 nslot = vslot;
                        K nxt = v;
                        for (unsigned b = 0; b < 4; ++b) {
                            const K c = succ(v, b);
                            const u64 s2 = table.find_any(canon(c));
                            if (table.vals[s2] != 0) { rflags |= 1u << b; nslot = s2; nxt = c; }
                        }
                        if (rflags == 0) {
                            sum += table.vals[vslot];
                            table.vals[vslot] = 65535;
                            break;
                        } else if (flag_count(rflags) > 1) {
                            rflags = 1u << rseq.back();
                            rseq.pop_back();
                            break;
                        }
                        rseq.push_back(flag2base(rflags));
                        sum += table.vals[vslot];
                        table.vals[vslot] = 65535;
                        v = nxt;
                        vslot = nslot;
                    }
                }
                if (!lseq.empty()) {                                    // graph.h:480-542
                    K v = pred(start, lseq[0]);
                    u64 vslot = table.find_any(canon(v));
                    while (true) {
                        lstop = v;
                        unsigned outdeg = 0;
                        for (unsigned b = 0; b < 4; ++b) if (table.value(canon(succ(v, b))) != 0) outdeg |= 1u << b;
                        if (flag_count(outdeg) > 1 || table.vals[vslot] == 65535) {
                            lflags = 1u << lseq.back();
                            lseq.pop_back();
                            break;
                        }
                        lflags = 0;
                        u64 nslot = vslot;
                        K nxt = v;
                        for (unsigned b = 0; b < 4; ++b) {
                            const K c = pred(v, b);
                            const u64 s2 = table.find_any(canon(c));
                            if (table.vals[s2] != 0) { lflags |= 1u << b; nslot = s2; nxt = c; }
                        }
                        if (lflags == 0) {
                            sum += table.vals[vslot];
                            table.vals[vslot] = 65535;
                            break;
                        } else if (flag_count(lflags) > 1) {
                            lflags = 1u << lseq.back();
                            lseq.pop_back();
                            break;
                        }
                        lseq.push_back(flag2base(lflags));
                        sum += table.vals[vslot];
                        table.vals[vslot] = 65535;
                        v = nxt;
                        vslot = nslot;
                    }
                }
                // graph.h:545-555: left stopper is pushed first, so the right one is processed first
                if (lflags != 0) { K s = lstop; s.set(k - 1, flag2base(lflags)); stack.push_back(s); }
                if (rflags != 0) { K s = rstop; s.set(0, flag2base(rflags)); stack.push_back(s); }

                Straight st;                                            // graph.h:557-575
                st.length = lseq.size() + rseq.size() + 1;
                // binstr_t(length + k) -> (length+k+31)/32 words (graph.h:559), not Straight::resize()'s rounding
                st.seq.assign((st.length + k + 31) / 32, 0);
                const u64 rn = rseq.size();
                for (u64 j = 0; j < rn; ++j) st.set(rn - 1 - j, rseq[j]);
                for (unsigned j = 0; j < k; ++j) st.set(rn + j, start.get(j));
                for (u64 j = 0; j < lseq.size(); ++j) st.set(rn + k + j, lseq[j]);
                st.coverage = (u16)(static_cast<double>(sum) / static_cast<double>(st.length) + 0.5);
                st.out = (uint8_t)((lflags << 4) | rflags);
                insert_straight(st);
            }
        }
    }

    // cutBranch (+ deleteErroneousStraightNode when coverage_cut != 0) with the decisions taken on the device (node_kernels.inc): the live
    // nodes go over in container order, the deletions are applied in container order.  Returns false if the device was not used.
    bool clean_pass_device(u64 length_cut, u64 coverage_cut, u64 &num_cut, u64 &num_err)
    {
        if (!clean_decide || !(branch_th < 1.0)) return false;       // (>= 1: two tips of one junction may delete each other: order matters)
        std::vector<u64> head, tail, len, tkeys, jkeys;
        std::vector<u16> cov, jcov;
        std::vector<uint8_t> out, jout, dec;
        std::vector<uint32_t> tids;
        std::vector<typename std::unordered_map<K, Straight, KeyHash<W>>::iterator> its;
        std::vector<const Straight *> tptr;
        {
            const size_t n0 = left.size();                            // (at 100 Mbp a pass exports a million nodes: no regrowth, no pointer-keyed map)
            head.reserve(n0 * W); tail.reserve(n0 * W); len.reserve(n0); cov.reserve(n0); out.reserve(n0); its.reserve(n0);
            tkeys.reserve(right.size() * W); tptr.reserve(right.size());
            jkeys.reserve(junctions.size() * W); jcov.reserve(junctions.size()); jout.reserve(junctions.size());
        }
        // the three tables are walked in container order, each by itself: on large graphs by three threads (a walk is one cache miss per
        // node and nothing else: r3, 100 Mbp -- a tenth of the whole run was this function)
        auto walk_left = [&]() {
            for (auto it = left.begin(), end = left.end(); it != end; ++it) {
                const Straight &st = it->second;
                st.export_id = 0;
                if (st.coverage == 65535) continue;
                st.export_id = (uint32_t)its.size() + 1;
                its.push_back(it);
                const K t = tail_of(st);
                for (int w = 0; w < W; ++w) { head.push_back(it->first.w[w]); tail.push_back(t.w[w]); }
                len.push_back(st.length);
                cov.push_back(st.coverage);
                out.push_back(st.out);
            }
        };
        auto walk_right = [&]() {                                     // rightStraightTable as it stands (an entry may point to a straight with another tail)
            for (auto &e : right) {
                if (e.second->coverage == 65535) continue;
                for (int w = 0; w < W; ++w) tkeys.push_back(e.first.w[w]);
                tptr.push_back(e.second);
            }
        };
        auto walk_junctions = [&]() {
            for (auto &e : junctions) {
                if (e.second.coverage == 65535) continue;
                for (int w = 0; w < W; ++w) jkeys.push_back(e.first.w[w]);
                jcov.push_back(e.second.coverage);
                jout.push_back(e.second.out);
            }
        };
        if (left.size() + right.size() + junctions.size() >= (1u << 16)) {
            std::exception_ptr err[2];
            std::thread tr([&]() { try { walk_right(); } catch (...) { err[0] = std::current_exception(); } });
            std::thread tj([&]() { try { walk_junctions(); } catch (...) { err[1] = std::current_exception(); } });
            try { walk_left(); } catch (...) { tr.join(); tj.join(); throw; }
            tr.join();
            tj.join();
            for (auto &e : err) if (e) std::rethrow_exception(e);
        } else {
            walk_left();
            walk_right();
            walk_junctions();
        }
        if (its.empty()) return false;
        tids.resize(tptr.size());                                     // (the export ids are the left walk's: read after both walks are done)
        {
            size_t o = 0;
            for (size_t i = 0; i < tptr.size(); ++i) {
                if (tptr[i]->export_id == 0) continue;                // (a live straight that the left table does not hold: cannot happen)
                if (o != i) for (int w = 0; w < W; ++w) tkeys[o * W + w] = tkeys[i * W + w];
                tids[o++] = tptr[i]->export_id - 1;
            }
            tids.resize(o);
            tkeys.resize(o * W);
        }
        dec.assign(its.size(), 0);
        if (!clean_decide(k, its.size(), head.data(), tail.data(), len.data(), cov.data(), out.data(), tids.size(), tkeys.data(), tids.data(), jcov.size(),
                          jkeys.data(), jcov.data(), jout.data(), branch_th, length_cut, coverage_cut, dec.data()))
            return false;
        num_cut = num_err = 0;
        for (size_t i = 0; i < its.size(); ++i) {                     // graph.h:882-883 / 913-914
            if (!(dec[i] & 1)) continue;
            Straight *st = &its[i]->second;
            if (st->out >> 4) {
                Junction *jn = find_junction(pred(its[i]->first, flag2base(st->out >> 4)));
                if (!jn) throw std::runtime_error("cutBranch: the device marked a tip whose junction is not in the junction table");
                jn->out ^= 1u << st->get(st->length - 1);
            } else {
                const K jk = right_of_tail(*st, flag2base(st->out & 0xf));
                Junction *jn = find_junction(jk);
                if (!jn) throw std::runtime_error("cutBranch: the device marked a tip whose junction is not in the junction table");
                jn->out ^= 1u << (4 + st->get(k - 1));
            }
            delete_straight(*st);
            ++num_cut;
        }
        for (size_t i = 0; i < its.size(); ++i) {                     // graph.h:2050-2066
            if (!(dec[i] & 2)) continue;
            const Straight &st = its[i]->second;
            if (st.out >> 4) {
                Junction *jn = find_junction(pred(its[i]->first, flag2base(st.out >> 4)));
                if (jn) jn->out ^= 1u << st.get(st.length - 1);
            }
            if (st.out & 0xf) {
                Junction *jn = find_junction(right_of_tail(st, flag2base(st.out & 0xf)));
                if (jn) jn->out ^= 1u << (4 + st.get(k - 1));
            }
            delete_straight(st);
            ++num_err;
        }
        return true;
    }

    // graph.h:829-926 (single thread)
    u64 cut_branch()
    {
        {
            u64 nc = 0, ne = 0;
            if (clean_pass_device(0, 0, nc, ne)) return nc;
        }
        u64 num_delete = 0;
        for (auto it = left.begin(), end = left.end(); it != end; ++it) {
            Straight *st = &it->second;
            if (st->coverage == 65535 || st->length > k) continue;
            if (st->out >> 4) {
                if (st->out & 0xf) continue;
                const K jk = pred(it->first, flag2base(st->out >> 4));
                Junction *jn = find_junction(jk);
                if (!jn) continue;
                u64 max_cov = 0;
                for (unsigned b = 0; b < 4; ++b) {
                    if (!(jn->out & (1u << b))) continue;
                    const K nk = succ(jk, b);
                    Straight *t = find_head(nk);
                    if (t != st && t != nullptr && t->coverage > max_cov) { max_cov = t->coverage; continue; }
                    Junction *tj = find_junction(nk);
                    if (tj != nullptr && tj->coverage > max_cov) max_cov = tj->coverage;
                }
                if (st->coverage > max_cov * branch_th) continue;
                jn->out ^= 1u << st->get(st->length - 1);
            } else if (st->out & 0xf) {
                const K jk = right_of_tail(*st, flag2base(st->out & 0xf));
                Junction *jn = find_junction(jk);
                if (!jn) continue;
                u64 max_cov = 0;
                for (unsigned b = 0; b < 4; ++b) {
                    if (!(jn->out & (1u << (b + 4)))) continue;
                    const K nk = pred(jk, b);
                    Straight *t = find_tail(nk);
                    if (t != st && t != nullptr && t->coverage > max_cov) { max_cov = t->coverage; continue; }
                    Junction *tj = find_junction(nk);
                    if (tj != nullptr && tj->coverage > max_cov) max_cov = tj->coverage;
                }
                if (st->coverage > max_cov * branch_th) continue;
                jn->out ^= 1u << (4 + st->get(k - 1));
            } else
                continue;
            delete_straight(*st);
            ++num_delete;
        }
        return num_delete;
    }

    // graph.h:1038-1061
    Straight concat_ss(Straight *l, Straight *r)
    {
        if (l->seq == r->seq) return *l;
        Straight n;
        n.length = l->length + r->length;
        n.coverage = static_cast<double>(l->coverage * l->length + r->coverage * r->length) / static_cast<double>(n.length) + 0.5;
        n.out = (l->out & 0xf0) | (r->out & 0x0f);
        n.resize(k);
        n.put_bases(r->length, l->seq, l->length + k - 1);          // n[r.length + p] = l[p]
        n.put_bases(0, r->seq, r->length);                          // n[q] = r[q], q < r.length
        delete_straight(*l);
        delete_straight(*r);
        insert_straight(n);
        return n;
    }
    // graph.h:1066-1089
    Straight concat_sj(const Straight *s, const K &jkey)
    {
        Straight n;
        const Junction j = junctions[jkey];
        n.length = s->length + 1;
        n.coverage = static_cast<double>(s->coverage * s->length + j.coverage) / static_cast<double>(n.length) + 0.5;
        n.out = (s->out & 0xf0) | (j.out & 0x0f);
        n.resize(k);
        n.set(0, jkey.get(0));
        n.put_bases(1, s->seq, s->length + k - 1);
        delete_straight(*s);
        delete_junction(jkey);
        insert_straight(n);
        return n;
    }
    // graph.h:1122-1141
    Straight junction_to_straight(const K &key)
    {
        const Junction j = junctions[key];
        Straight n;
        n.length = 1;
        n.coverage = j.coverage;
        n.out = j.out;
        n.resize(k);
        for (unsigned i = 0; i < k; ++i) n.set(i, key.get(i));
        delete_junction(key);
        insert_straight(n);
        return n;
    }
    // graph.h:1094-1117
    Straight concat_jj(const K &lkey, const K &rkey)
    {
        if (lkey == rkey) return junction_to_straight(lkey);
        const Junction lj = junctions[lkey];
        const Junction rj = junctions[rkey];
        Straight n;
        n.length = 2;
        n.coverage = static_cast<double>(lj.coverage + rj.coverage) / 2.0 + 0.5;
        n.out = (lj.out & 0xf0) | (rj.out & 0x0f);
        n.resize(k);
        n.set(k, lkey.get(k - 1));
        for (unsigned i = 0; i < k; ++i) n.set(k - 1 - i, rkey.get(k - 1 - i));
        delete_junction(lkey);
        delete_junction(rkey);
        insert_straight(n);
        return n;
    }

    // graph.h:933-1033
    void concatinate_nodes()
    {
        Straight fresh;
        for (auto it = junctions.begin(), end = junctions.end(); it != end; ++it) {
            Junction *jn = &it->second;
            if (jn->coverage == 65535 || flag_count(jn->out >> 4) > 1 || flag_count(jn->out & 0xf) > 1) continue;
            Straight *ls = nullptr, *rs = nullptr;
            Junction *lj = nullptr, *rj = nullptr;
            K key;
            if (flag_count(jn->out >> 4)) {
                key = pred(it->first, flag2base(jn->out >> 4));
                ls = find_tail(key);
                if (ls == nullptr) {
                    lj = find_junction(key);
                    if (lj == nullptr || flag_count(lj->out >> 4) > 1 || flag_count(lj->out & 0xf) > 1) lj = nullptr;
                }
            }
            if (ls != nullptr) fresh = concat_sj(ls, it->first);
            else if (lj != nullptr) fresh = concat_jj(key, it->first);
            else fresh = junction_to_straight(it->first);
            ls = &fresh;

            if (flag_count(ls->out & 0xf)) {
                key = right_of_tail(*ls, flag2base(ls->out & 0xf));
                rs = find_head(key);
                if (rs == nullptr) {
                    rj = find_junction(key);
                    if (rj == nullptr || flag_count(rj->out >> 4) > 1 || flag_count(rj->out & 0xf) > 1) rj = nullptr;
                }
            }
            if (rs != nullptr) concat_ss(ls, rs);
            else if (rj != nullptr) concat_sj(ls, key);
        }

        for (auto it = left.begin(), end = left.end(); it != end; ++it) {
            Straight *st = &it->second;
            Straight merged;                       // stays empty (out == 0) unless a left merge happens (Q13)
            if (st->coverage == 65535) continue;
            if (flag_count(st->out >> 4)) {
                const K key = pred(it->first, flag2base(st->out >> 4));
                Straight *ls = find_tail(key);
                if (ls != nullptr) merged = concat_ss(ls, st);
                st = &merged;
            }
            if (flag_count(st->out & 0xf)) {
                const K key = right_of_tail(*st, flag2base(st->out & 0xf));
                Straight *rs = find_head(key);
                if (rs != nullptr) concat_ss(st, rs);
            }
        }
    }

    void cut_branch_iterative(bool verbose)              // graph.h:1829-1845
    {
        u64 total = 0, n;
        if (verbose) std::cerr << "removing branches...\nBRANCH_DELETE_THRESHOLD=" << branch_th << std::endl;
        do {
            n = cut_branch();
            if (verbose) std::cerr << "NUM_CUT=" << n << std::endl;
            concatinate_nodes();
            total += n;
        } while (n > 0);
        if (verbose) std::cerr << "TOTAL_NUM_CUT=" << total << std::endl;
    }

    // graph.h:2032-2082 (single thread)
    u64 delete_erroneous(u64 length_cut, u64 coverage_cut)
    {
        u64 num = 0;
        for (auto it = left.begin(), end = left.end(); it != end; ++it) {
            const Straight &st = it->second;
            if (st.coverage == 65535 || st.coverage >= coverage_cut || st.length >= length_cut) continue;
            if (st.out >> 4) {
                Junction *jn = find_junction(pred(it->first, flag2base(st.out >> 4)));
                if (jn) jn->out ^= 1u << st.get(st.length - 1);
            }
            if (st.out & 0xf) {
                Junction *jn = find_junction(right_of_tail(st, flag2base(st.out & 0xf)));
                if (jn) jn->out ^= 1u << (4 + st.get(k - 1));
            }
            delete_straight(st);
            ++num;
        }
        return num;
    }
    void delete_erroneous_iterative(u64 length_cut, u64 coverage_cut, bool verbose)      // graph.h:2091-2106
    {
        u64 total = 0, n;
        if (verbose) std::cerr << "removing erroneous nodes..." << std::endl;
        do {
            u64 nc = 0, ne = 0;
            if (coverage_cut && clean_pass_device(length_cut, coverage_cut, nc, ne)) n = nc + ne;       // both decisions in one device pass
            else {
                n = cut_branch();
                n += delete_erroneous(length_cut, coverage_cut);
            }
            if (verbose) std::cerr << "NUM_REMOVED_NODES=" << n << std::endl;
            concatinate_nodes();
            total += n;
        } while (n > 0);
        if (verbose) std::cerr << "TOTAL_NUM_REMOVED_NODES=" << total << std::endl;
    }

    // graph.h:687-706
    void search_bubble(Junction *root, Junction *jn[4], Straight *st[4], const K &root_key)
    {
        for (unsigned b = 0; b < 4; ++b) {
            if (!(root->out & (1u << b))) continue;
            st[b] = find_head(succ(root_key, b));
            if (st[b] != nullptr && (st[b]->out & 0x0f)) {
                jn[b] = find_junction(right_of_tail(*st[b], flag2base(st[b]->out & 0x0f)));
            }
        }
    }
    // graph.h:712-746
    u64 pairwise_alignment(const Straight *s1, const Straight *s2) const
    {
        u64 asize = k;
        std::vector<u64> sc(asize * 2);
        const u64 l1 = s1->length - k + 1, l2 = s2->length - k + 1;
        if (l2 + 1 > asize) { asize = l2 + 1; sc.resize(asize * 2); }
        u64 m, n;
        for (n = 0; n < l2 + 1; ++n) sc[n] = n;
        for (m = 0; m < l1; ++m) {
            sc[(m & 1) * asize] = m;
            sc[(~m & 1) * asize] = m + 1;
            for (n = 0; n < l2; ++n) {
                if (s1->get(s1->length - 1 - m) == s2->get(s2->length - 1 - n)) {
                    sc[(~m & 1) * asize + n + 1] = sc[(m & 1) * asize + n];
                    continue;
                }
                u64 &d = sc[(~m & 1) * asize + n + 1];
                d = sc[(m & 1) * asize + n] + 1;
                if (d > sc[(m & 1) * asize + n + 1] + 1) d = sc[(m & 1) * asize + n + 1] + 1;
                if (d > sc[(~m & 1) * asize + n] + 1) d = sc[(~m & 1) * asize + n] + 1;
            }
        }
        return sc[(m & 1) * asize + n];
    }
    // graph.h:591-681
    u64 crush_bubble(double avg_cov)
    {
        u64 num = 0;
        u64 cluster[4];
        Junction *jn[4];
        Straight *st[4], *mx[4];
        u16 cov_th = static_cast<u16>(std::min(static_cast<u64>(avg_cov * 1.5 + 0.5), static_cast<u64>(65535)));
        if (avg_cov >= 65535) cov_th = 65535;
        // Device: the edit distances of ALL pairs of branches that leave one junction and meet again in one junction, at once (the
        // sequences do not change while bubbles are crushed; nodes only disappear, so this is a superset of what the loop below asks for)
        std::map<std::pair<const Straight *, const Straight *>, u64> known;
        if (edit_distances) {
            std::vector<uint8_t> bases;
            std::vector<u64> offs{0};
            std::vector<std::pair<const Straight *, const Straight *>> pairs;
            for (auto it = junctions.begin(), end = junctions.end(); it != end; ++it) {
                Junction *root = &it->second;
                if (root->coverage == 65535 || flag_count(root->out & 0x0f) < 2) continue;
                for (unsigned b = 0; b < 4; ++b) { st[b] = nullptr; jn[b] = nullptr; }
                search_bubble(root, jn, st, it->first);
                for (unsigned b1 = 0; b1 < 3; ++b1)
                    for (unsigned b2 = b1 + 1; b2 < 4; ++b2) {
                        if (st[b1] == nullptr || st[b2] == nullptr || jn[b1] == nullptr || jn[b1] != jn[b2]) continue;
                        if (std::min(st[b1]->length, st[b2]->length) + 1 <= k) continue;
                        for (const Straight *sx : {st[b1], st[b2]}) {                  // graph.h:716-717,726: the part in front of the shared k - 1 bases
                            const u64 l = sx->length - k + 1;
                            for (u64 m = 0; m < l; ++m) bases.push_back((uint8_t)sx->get(sx->length - 1 - m));
                            offs.push_back(bases.size());
                        }
                        pairs.push_back({st[b1], st[b2]});
                    }
            }
            if (!pairs.empty()) {
                std::vector<u64> dist(pairs.size());
                if (edit_distances(bases.data(), offs.data(), pairs.size(), dist.data()))
                    for (size_t i = 0; i < pairs.size(); ++i) known[pairs[i]] = dist[i];
            }
        }
        for (auto it = junctions.begin(), end = junctions.end(); it != end; ++it) {
            Junction *root = &it->second;
            if (root->coverage == 65535 || flag_count(root->out & 0x0f) < 2) continue;
            for (unsigned b = 0; b < 4; ++b) { cluster[b] = b; st[b] = nullptr; jn[b] = nullptr; }
            search_bubble(root, jn, st, it->first);
            for (unsigned b1 = 0; b1 < 3; ++b1) {
                if (st[b1] == nullptr) continue;
                for (unsigned b2 = b1 + 1; b2 < 4; ++b2) {
                    if (st[b2] == nullptr || jn[b1] == nullptr || jn[b1] != jn[b2] || st[b1]->coverage + st[b2]->coverage > cov_th) continue;
                    const u64 ath = static_cast<u64>((std::max(st[b1]->length, st[b2]->length) + k - 1) * bubble_th + 0.5);
                    if (std::min(st[b1]->length, st[b2]->length) + 1 <= k) {
                        if ((u64)std::labs(static_cast<long>(st[b1]->length) - static_cast<long>(st[b2]->length)) > ath) continue;
                        if (cluster[b2] == b2) cluster[b2] = cluster[b1]; else cluster[b1] = cluster[b2];
                        continue;
                    }
                    const auto kn = known.find({st[b1], st[b2]});
                    if ((kn != known.end() ? kn->second : pairwise_alignment(st[b1], st[b2])) > ath) continue;
                    if (cluster[b2] == b2) cluster[b2] = cluster[b1]; else cluster[b1] = cluster[b2];
                }
            }
            for (unsigned b1 = 0; b1 < 3; ++b1) {
                mx[b1] = nullptr;
                for (unsigned b2 = b1; b2 < 4; ++b2) {
                    if (cluster[b2] != b1 || st[b2] == nullptr) continue;
                    if (mx[b1] == nullptr || st[b2]->coverage > mx[b1]->coverage) mx[b1] = st[b2];
                }
            }
            for (unsigned b1 = 0; b1 < 3; ++b1) {
                if (mx[b1] == nullptr) continue;
                for (unsigned b2 = b1; b2 < 4; ++b2) {
                    if (cluster[b2] != b1 || st[b2] == nullptr || st[b2] == mx[b1]) continue;
                    u64 tmp = mx[b1]->coverage * mx[b1]->length + st[b2]->coverage * st[b2]->length;
                    tmp = static_cast<double>(tmp) / mx[b1]->length + 0.5;
                    mx[b1]->coverage = std::min(tmp, static_cast<u64>(65534));
                    root->out ^= 1u << b2;
                    jn[b2]->out ^= 1u << (st[b2]->get(k - 1) + 4);
                    {
                        ContigRecord r;                                   // graph.h:661-668
                        r.length = st[b2]->length + k - 1;
                        r.cov = st[b2]->coverage;
                        r.seq.assign((r.length + 31) / 32, 0);
                        for (size_t i = 0; i < r.seq.size() && i < st[b2]->seq.size(); ++i) r.seq[i] = st[b2]->seq[i];
                        bubbles.push_back(r);
                    }
                    delete_straight(*st[b2]);
                    ++num;
                }
            }
        }
        return num;
    }
    void crush_bubble_iterative(double avg_cov, bool verbose)       // graph.h:1851-1870
    {
        if (verbose) std::cerr << "removing bubbles...\nBUBBLE_IDENTITY_THRESHOLD=" << bubble_th << std::endl;
        bubbles.clear();
        u64 total = 0, n;
        do {
            n = crush_bubble(avg_cov);
            if (verbose) std::cerr << "NUM_REMOVED_BUBBLES=" << n << std::endl;
            concatinate_nodes();
            total += n;
        } while (n > 0);
        if (verbose) std::cerr << "TOTAL_NUM_REMOVED_BUBBLES=" << total << std::endl;
    }

    // graph.h:2000-2022
    static u64 left_minimal(const std::vector<u64> &dist)
    {
        u64 i, pre;
        const u64 size = dist.size();
        if (size == 0) return 0;
        for (i = 0; i < size; ++i) if (dist[i] > 0) break;
        if (i >= size) return 1;          // the reference reads dist[size] here; an empty graph is out of scope
        pre = dist[i];
        for (++i; i < size; ++i) {
            if (dist[i] >= pre) break;
            pre = dist[i];
        }
        return i < size ? i - 1 : 1;
    }
    u64 left_minimal_coverage() const                  // graph.h:1981-1993
    {
        std::vector<u64> dist(65535);
        for (auto &e : left) if (e.second.coverage != 65535) dist[e.second.coverage] += e.second.length;
        return left_minimal(dist);
    }

    // graph.h:2374-2431, including the non-reset scratch arrays
    double average_coverage_excluding_bubble()
    {
        Junction *jn[4] = {nullptr, nullptr, nullptr, nullptr};
        Straight *st[4] = {nullptr, nullptr, nullptr, nullptr};
        std::unordered_map<K, bool, KeyHash<W>> is_bubble;
        for (auto it = junctions.begin(), end = junctions.end(); it != end; ++it) {
            Junction *root = &it->second;
            if (root->coverage == 65535 || flag_count(root->out & 0x0f) < 2) continue;
            search_bubble(root, jn, st, it->first);
            for (unsigned b1 = 0; b1 < 3; ++b1) {
                if (st[b1] == nullptr) continue;
                for (unsigned b2 = b1 + 1; b2 < 4; ++b2) {
                    if (st[b2] == nullptr || st[b1]->coverage == 65535 || st[b2]->coverage == 65535 || jn[b1] == nullptr || jn[b1] != jn[b2]) continue;
                    is_bubble[tail_of(*st[b1])] = true;
                    is_bubble[tail_of(*st[b2])] = true;
                }
            }
        }
        u64 sum = 0, num = 0;
        for (auto &e : left) {
            if (e.second.coverage == 65535) continue;
            if (is_bubble.find(tail_of(e.second)) == is_bubble.end()) {
                num += e.second.length;
                sum += e.second.length * e.second.coverage;
            }
        }
        for (auto &e : junctions) if (e.second.coverage != 65535) { ++num; sum += e.second.coverage; }
        return static_cast<double>(sum) / static_cast<double>(num);
    }

    // graph.h:1877-1939
    // put(canonical key, coverage): overwrite semantics (counter.h:90 setOccurrenceValue)
    typedef typename std::unordered_map<K, Straight, KeyHash<W>>::iterator LeftIt;
    // `parts` + 1 iterators that cut leftStraightTable's sequence into runs of equal length (one walk): the read-only passes over the
    // table -- edge k-mers, contig strings -- then run on several threads, each on its run, and their results are put together in order
    std::vector<LeftIt> left_cuts(unsigned parts)
    {
        std::vector<LeftIt> cuts;
        const size_t n = left.size();
        size_t i = 0, next = 0;
        unsigned c = 0;
        for (auto it = left.begin(), end = left.end(); it != end; ++it, ++i)
            if (i == next && c < parts) { cuts.push_back(it); ++c; next = n * c / parts; if (next <= i) next = i + 1; }
        while (cuts.size() < parts + 1) cuts.push_back(left.end());
        return cuts;
    }
    template <typename PUT>
    void save_edge_kmers(PUT &&put, unsigned next_k) { save_edge_kmers_range(left.begin(), left.end(), put, next_k); }
    template <typename PUT>
    void save_edge_kmers_range(LeftIt from, LeftIt to, PUT &&put, unsigned next_k) const
    {
        const unsigned diff = next_k - k;
        for (auto eit = from; eit != to; ++eit) {
            const Straight &st = eit->second;
            if (st.coverage == 65535) continue;
            auto emit_run = [&](u64 first_pos, u64 count) {
                // k-mers whose 3' bases are at positions first_pos, first_pos-1, ... (count of them): the first one is taken out of the
                // sequence whole, the others are rolled
                Kmer<W> km(k);
                km.set_forward(key_at(st, first_pos));
                for (u64 i = 0; i < count; ++i) {
                    if (i) km.push_right(st.get(first_pos - i));
                    if (st.coverage != 0) put(km.canon(), st.coverage);
                }
            };
            if (st.length < (u64)diff * 2) {
                emit_run(st.length - 1, st.length);
            } else if (diff > 0) {
                // the reference interleaves left and right inserts (graph.h:1919-1935); slot layout of this
                // scratch table is never observed, only membership, so the two runs are emitted in turn here too
                Kmer<W> lk(k), rk(k);
                lk.set_forward(key_at(st, st.length - 1));
                rk.set_forward(key_at(st, diff - 1));
                for (unsigned i = 0; i < diff; ++i) {
                    if (i) lk.push_right(st.get(st.length - 1 - i));
                    if (st.coverage != 0) put(lk.canon(), st.coverage);
                    if (i) rk.push_right(st.get(diff - 1 - i));
                    if (st.coverage != 0) put(rk.canon(), st.coverage);
                }
            }
        }
    }

    // graph.h:1655-1771
    // graph.h:1655-1737: for every live straight, the string [<= diff tail bases of the unique straight left of the left
    // junction] + [left junction base] + straight + [right junction base] + [<= diff head bases ...] (left -> right, codes
    // 0..3) and occ = (u16)(coverage * ratio + 0.5).  f(str, occ) is called for strings of at least next_k bases.
    template <typename F>
    void for_each_contig_string(unsigned next_k, double ratio, F &&f) { for_each_contig_string_range(left.begin(), left.end(), next_k, ratio, f); }
    template <typename F>
    void for_each_contig_string_range(LeftIt from, LeftIt to, unsigned next_k, double ratio, F &&f)
    {
        const u64 diff = next_k - k;
        std::vector<unsigned char> str;            // left -> right
        for (auto it = from, end = to; it != end; ++it) {
            const Straight *st = &it->second;
            if (st->coverage == 65535) continue;
            str.clear();
            if (st->out >> 4) {
                const K jk = pred(it->first, flag2base(st->out >> 4));
                Junction *jn = find_junction(jk);
                if (jn != nullptr) {
                    if (flag_count(jn->out >> 4) == 1) {
                        Straight *t = find_tail(pred(jk, flag2base(jn->out >> 4)));
                        if (t != nullptr) {
                            const unsigned ex = (unsigned)std::min<u64>(t->length, diff);
                            for (unsigned i = 0; i < ex; ++i) str.push_back(t->get(k + ex - 2 - i));
                        }
                    }
                    str.push_back(jk.get(k - 1));
                }
            }
            st->append_bases(str, st->length + k - 1);
            if (st->out & 0xf) {
                const K jk = right_of_tail(*st, flag2base(st->out & 0xf));
                Junction *jn = find_junction(jk);
                if (jn != nullptr) {
                    str.push_back(jk.get(0));
                    if (flag_count(jn->out & 0xf) == 1) {
                        Straight *t = find_head(succ(jk, flag2base(jn->out & 0xf)));
                        if (t != nullptr) {
                            const unsigned ex = (unsigned)std::min<u64>(t->length, diff);
                            for (unsigned i = 0; i < ex; ++i) str.push_back(t->get(t->length - 1 - i));
                        }
                    }
                }
            }
            if (str.size() < next_k) continue;
            const u16 occ = st->coverage * ratio + 0.5;
            f(str, occ);
        }
    }

    // graph.h:1738-1767 on top of the strings: maxput(canonical next-k key, occ) means value = max(value, occ), nothing
    // written when occ is not larger
    template <int WN, typename MAXPUT>
    void save_contig(unsigned next_k, double ratio, MAXPUT &&maxput)
    {
        for_each_contig_string(next_k, ratio, [&](const std::vector<unsigned char> &str, u16 occ) {
            Kmer<WN> km(next_k);
            for (unsigned i = 0; i < next_k - 1; ++i) km.push_right(str[i]);
            for (u64 i = next_k - 1; i < str.size(); ++i) {
                km.push_right(str[i]);
                maxput(km.canon(), occ);
            }
        });
    }

    void save_contig_simple(std::vector<ContigRecord> &out, double ratio) const       // graph.h:1774-1796
    {
        for (auto &e : left) {
            const Straight &st = e.second;
            if (st.coverage == 65535) continue;
            ContigRecord r;
            r.length = st.length + k - 1;
            r.cov = static_cast<u16>(st.coverage * ratio + 0.5);
            r.seq.assign((r.length + 31) / 32, 0);
            for (size_t i = 0; i < r.seq.size() && i < st.seq.size(); ++i) r.seq[i] = st.seq[i];
            out.push_back(r);
        }
    }
    void save_junction(std::vector<ContigRecord> &out, double ratio) const            // graph.h:1799-1823
    {
        for (auto &e : junctions) {
            if (e.second.coverage == 65535) continue;
            ContigRecord r;
            r.length = k;
            r.cov = static_cast<u16>(e.second.coverage * ratio + 0.5);
            r.seq.assign(e.first.w, e.first.w + (k + 31) / 32);
            out.push_back(r);
        }
    }
};

} // namespace pbh
