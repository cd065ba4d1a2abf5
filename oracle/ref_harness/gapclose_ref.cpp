// TEST INFRASTRUCTURE ONLY.  Calls the REFERENCE's own Counter<KMER>::countKmerForGapClose (counter.h:1015-1056) and
// BruijnGraph<KMER>::makeBruijnGraphForGapClose (graph.h:2116-2345) on a batch of gaps and prints the tables they leave, in the tables'
// own iteration order.  This file is ours; the reference's headers and objects are taken where they lie (oracle/Makefile, target
// `ref`: $(REF)/*.h by -I, the objects of $(OUT)/obj without main.o) and nothing of them is copied.  Output goes to oracle/_ref/.
// usage: gapclose_ref INPUT.bin > dump.txt                  the two functions at k
//        gapclose_ref INPUT.bin K_HIGH > dump.txt           GapCloseDBG<k, K_HIGH>::gapAssemble (gapCloseDBG.h:416-436) per gap: per gap "G <index>",
//                                                           "L", the low graph's live nodes, "H", the high graph's (same line format)
//        gapclose_ref INPUT.bin K_HIGH FLANKS.bin > dump.txt  ... and what the reference then decides about the gap (gapClose.cpp:716-760 without
//                                                           the OLC fallback; -vd 32, -ed 0.01): FLANKS.bin per gap: i32 start, i32 end, u8 at a
//                                                           scaffold's end, u32 head length, u32 tail length, head codes, tail codes; lines
//                                                           "X" (not enabled) | "C closed seqLength bases.." [+ "P closed seqLength remainedGap bases.."] | "E .."
//   INPUT.bin: u32 k, i64 min_coverage, u64 n_gaps, u64 seq_first[n_gaps + 1], u64 n_seqs, u64 seq_off[n_seqs + 1], u8 codes[] (0..3, 4 = N)
//   dump: per gap "G <index>", then "J <key words, most significant first> <coverage> <out>" per junction and
//         "S <length> <coverage> <out> <sequence words>" per straight node (hex), as oracle/src/capi.cpp pbo_gap_graphs_dump prints them
// (compiled with -fno-access-control: the tables are private members of BruijnGraph)
#include "counter.h"
#include "graph.h"
#include "gapClose.h"
#include "gapCloseDBG.h"
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

static void key_words(const Kmer31Key &k, unsigned W, std::vector<unsigned long long> &w) { (void)W; w.assign(1, k); }
template <typename B> static void key_words(const B &k, unsigned W, std::vector<unsigned long long> &w) { w.assign(k.value, k.value + W); }

template <typename KMER>
static void run(unsigned k, long min_cov, const std::vector<unsigned long long> &seq_first, const std::vector<unsigned long long> &seq_off, const std::vector<unsigned char> &codes)
{
    const unsigned W = (k + 31) / 32;
    for (size_t g = 0; g + 1 < seq_first.size(); ++g) {
        std::vector<std::vector<char> > seq(GapClose::GAP_CANDIDATE_SEQ_1ST);        // the entries the reference skips
        for (unsigned long long s = seq_first[g]; s < seq_first[g + 1]; ++s) seq.push_back(std::vector<char>(codes.begin() + seq_off[s], codes.begin() + seq_off[s + 1]));
        Counter<KMER> counter;
        counter.setKmerLength(k);
        counter.countKmerForGapClose(seq);
        BruijnGraph<KMER> graph;
        graph.setKmerLength(k);
        graph.makeBruijnGraphForGapClose(counter, min_cov);
        printf("G %zu\n", g);
        std::vector<unsigned long long> w;
        for (auto it = graph.junctionTable.begin(); it != graph.junctionTable.end(); ++it) {
            key_words(it->first, W, w);
            printf("J");
            for (int i = (int)W - 1; i >= 0; --i) printf(" %llx", w[i]);
            printf(" %x %x\n", (unsigned)it->second.coverage, (unsigned)it->second.out);
        }
        for (auto it = graph.leftStraightTable.begin(); it != graph.leftStraightTable.end(); ++it) {
            const Straight &st = it->second;
            printf("S %llx %x %x", (unsigned long long)st.length, (unsigned)st.coverage, (unsigned)st.out);
            const unsigned long long nb = st.length + k - 1;
            for (unsigned long long i = 0; i < (nb + 31) / 32; ++i) {
                unsigned long long v = i < st.seq.size() ? st.seq[i] : 0;
                if (i == (nb + 31) / 32 - 1 && (nb % 32)) v &= ~0ull >> (64 - 2 * (nb % 32));
                printf(" %llx", v);
            }
            printf("\n");
        }
    }
}

template <typename KMER>
static void dump_live(BruijnGraph<KMER> &graph, unsigned k)
{
    const unsigned W = (k + 31) / 32;
    std::vector<unsigned long long> w;
    for (auto it = graph.junctionTable.begin(); it != graph.junctionTable.end(); ++it) {
        if (it->second.coverage == UINT16_MAX) continue;
        key_words(it->first, W, w);
        printf("J");
        for (int i = (int)W - 1; i >= 0; --i) printf(" %llx", w[i]);
        printf(" %x %x\n", (unsigned)it->second.coverage, (unsigned)it->second.out);
    }
    for (auto it = graph.leftStraightTable.begin(); it != graph.leftStraightTable.end(); ++it) {
        const Straight &st = it->second;
        if (st.coverage == UINT16_MAX) continue;
        printf("S %llx %x %x", (unsigned long long)st.length, (unsigned)st.coverage, (unsigned)st.out);
        const unsigned long long nb = st.length + k - 1;
        for (unsigned long long i = 0; i < (nb + 31) / 32; ++i) {
            unsigned long long v = i < st.seq.size() ? st.seq[i] : 0;
            if (i == (nb + 31) / 32 - 1 && (nb % 32)) v &= ~0ull >> (64 - 2 * (nb % 32));
            printf(" %llx", v);
        }
        printf("\n");
    }
}
template <typename G> static void dump_width(G &e, unsigned k)
{
    if (k <= 32) dump_live(e.kmer31Graph, k);
    else if (k <= 64) dump_live(e.kmer63Graph, k);
    else if (k <= 96) dump_live(e.kmer95Graph, k);
    else if (k <= 128) dump_live(e.kmer127Graph, k);
    else dump_live(e.kmer159Graph, k);
}
static FILE *g_flanks = NULL;
static void print_seq(const char *tag, bool ok, int len, int remained, const std::vector<char> &seq, bool with_remained)
{
    printf("%s %d %d", tag, ok ? 1 : 0, ok ? len : 0);                   // (what a failed attempt leaves in seqLength is not part of the result)
    if (with_remained) printf(" %d", ok ? remained : 0);
    for (size_t i = 0; i < seq.size(); ++i) printf(" %d", (int)seq[i]);
    printf("\n");
}
template <unsigned KL, unsigned KH>
static void run_assemble(long min_cov, const std::vector<unsigned long long> &seq_first, const std::vector<unsigned long long> &seq_off, const std::vector<unsigned char> &codes)
{
    for (size_t g = 0; g + 1 < seq_first.size(); ++g) {
        GAP gap;
        gap.seq.resize(GapClose::GAP_CANDIDATE_SEQ_1ST);
        for (unsigned long long s = seq_first[g]; s < seq_first[g + 1]; ++s) gap.seq.push_back(std::vector<char>(codes.begin() + seq_off[s], codes.begin() + seq_off[s + 1]));
        bool at_end = false;
        if (g_flanks) {
            int32_t se[2]; uint8_t e; uint32_t hl, tl;
            if (fread(se, 4, 2, g_flanks) != 2 || fread(&e, 1, 1, g_flanks) != 1 || fread(&hl, 4, 1, g_flanks) != 1 || fread(&tl, 4, 1, g_flanks) != 1) exit(7);
            gap.start = se[0]; gap.end = se[1]; at_end = e != 0;
            gap.headSeq.resize(hl); gap.tailSeq.resize(tl);
            if ((hl && fread(gap.headSeq.data(), 1, hl, g_flanks) != hl) || (tl && fread(gap.tailSeq.data(), 1, tl, g_flanks) != tl)) exit(7);
        }
        GapCloseDBG<KL, KH> ele(KL, KH, min_cov, 32, 0.01, &gap);          // gapClose.cpp:57,60: -vd 32, -ed 0.01
        printf("G %zu\n", g);
        if (g_flanks && !at_end && !ele.enableGapClose()) { printf("X\n"); continue; }      // gapClose.cpp:716-717
        ele.gapAssemble();
        printf("L\n");
        dump_width(ele, KL);
        printf("H\n");
        dump_width(ele, KH);
        if (!g_flanks) continue;
        if (!at_end) {
            const bool closed = ele.isClosedGap();
            print_seq("C", closed, ele.getSeqLength(), 0, closed ? ele.getSeq() : std::vector<char>(), false);
            if (!closed) {                                                    // gapClose.cpp:743-751 (-no_partial off)
                const bool part = ele.isClosedGapPartial();
                print_seq("P", part, ele.getSeqLength(), ele.getRemainedGap(), part ? ele.getSeq() : std::vector<char>(), true);
            }
        } else {
            const bool ext = ele.extendEdge();                                // gapClose.cpp:754-760
            print_seq("E", ext, ele.getSeqLength(), 0, ext ? ele.getSeq() : std::vector<char>(), false);
        }
    }
}

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t k = 0; int64_t mc = 0; uint64_t ng = 0, ns = 0;
    if (fread(&k, 4, 1, f) != 1 || fread(&mc, 8, 1, f) != 1 || fread(&ng, 8, 1, f) != 1) return 3;
    std::vector<unsigned long long> seq_first(ng + 1);
    if (fread(seq_first.data(), 8, ng + 1, f) != ng + 1 || fread(&ns, 8, 1, f) != 1) return 3;
    std::vector<unsigned long long> seq_off(ns + 1);
    if (fread(seq_off.data(), 8, ns + 1, f) != ns + 1) return 3;
    std::vector<unsigned char> codes(seq_off[ns]);
    if (!codes.empty() && fread(codes.data(), 1, codes.size(), f) != codes.size()) return 3;
    fclose(f);
    if (argc >= 3) {
        const unsigned kh = (unsigned)atoi(argv[2]);
        if (argc >= 4 && !(g_flanks = fopen(argv[3], "rb"))) return 2;
        if (k == 20 && kh == 80) run_assemble<20, 80>(mc, seq_first, seq_off, codes);          // gapClose.cpp:41-42: the command's own pair
        else if (k == 25 && kh == 60) run_assemble<25, 60>(mc, seq_first, seq_off, codes);
        else if (k == 28 && kh == 45) run_assemble<28, 45>(mc, seq_first, seq_off, codes);
        else if (k == 32 && kh == 100) run_assemble<32, 100>(mc, seq_first, seq_off, codes);
        else if (k == 40 && kh == 90) run_assemble<40, 90>(mc, seq_first, seq_off, codes);
        else return 5;
        return 0;
    }
    if (k <= 32) run<Kmer31>(k, mc, seq_first, seq_off, codes);
    else if (k <= 64) run<KmerN<Binstr63> >(k, mc, seq_first, seq_off, codes);
    else if (k <= 96) run<KmerN<Binstr95> >(k, mc, seq_first, seq_off, codes);
    else if (k <= 128) run<KmerN<Binstr127> >(k, mc, seq_first, seq_off, codes);
    else if (k <= 160) run<KmerN<Binstr159> >(k, mc, seq_first, seq_off, codes);
    else return 4;
    return 0;
}
