// The graph part of a gap's assembly through the product: platanus_b_amd/host/gap_graphs.hpp gap_assemble_batch (device: counting, graph
// construction, merged table; host: the node-level cleaning) on the input format of oracle/ref_harness/gapclose_ref.cpp, printed in that
// harness's format -- so that its output can be compared with dumps of the reference's own GapCloseDBG<M, N>::gapAssemble tables.
// usage: gap_assemble_host INPUT.bin K_HIGH > dump.txt      (needs a GPU)
//        gap_assemble_host INPUT.bin K_HIGH time            no dump: wall time of gap_assemble_batch and the nodes left, one line
//        gap_assemble_host INPUT.bin K_HIGH FLANKS.bin      ... and the closing decisions (pbh::GapCloser: closedGap / closedGapPartial / extendEdge), as the harness prints them
#include "../../platanus_b_amd/host/gap_graphs.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

template <int W> static void dump(pbh::Graph<W> &gr, unsigned k)
{
    for (auto &e : gr.junctions) {
        if (e.second.coverage == 65535) continue;
        printf("J");
        for (int w = W - 1; w >= 0; --w) printf(" %llx", (unsigned long long)e.first.w[w]);
        printf(" %x %x\n", (unsigned)e.second.coverage, (unsigned)e.second.out);
    }
    for (auto &e : gr.left) {
        if (e.second.coverage == 65535) continue;
        printf("S %llx %x %x", (unsigned long long)e.second.length, (unsigned)e.second.coverage, (unsigned)e.second.out);
        const unsigned long long nb = e.second.length + k - 1;
        for (unsigned long long i = 0; i < (nb + 31) / 32; ++i) {
            unsigned long long w = i < e.second.seq.size() ? e.second.seq[i] : 0;
            if (i == (nb + 31) / 32 - 1 && (nb % 32)) w &= ~0ull >> (64 - 2 * (nb % 32));
            printf(" %llx", w);
        }
        printf("\n");
    }
}
static bool g_time_only = false;
static FILE *g_flanks = nullptr;
static void closure_line(const char *tag, bool ok, const pbh::GapClosure &c, bool with_remained)
{
    printf("%s %d %d", tag, ok ? 1 : 0, ok ? c.seq_length : 0);
    if (with_remained) printf(" %d", ok ? c.remained_gap : 0);
    if (ok) for (char b : c.seq) printf(" %d", (int)b);
    printf("\n");
}
template <int WL, int WH> static void run(pb_ctx *ctx, unsigned kl, unsigned kh, long mc, const pbh::GapBatch &b)
{
    std::vector<pbh::Graph<WL>> low;
    std::vector<pbh::Graph<WH>> high;
    const auto t0 = std::chrono::steady_clock::now();
    pbh::gap_assemble_batch<WL, WH>(ctx, kl, kh, mc, b, low, high);
    if (g_time_only) {
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long nodes = 0;
        for (auto &g : high) for (auto &e : g.left) nodes += e.second.coverage != 65535;
        float ms = 0;
        (void)pb_last_kernel_ms(ctx, "gap_graphs", &ms);
        printf("{\"gaps\": %zu, \"k_low\": %u, \"k_high\": %u, \"gap_assemble_batch_s\": %.3f, \"last_device_call_kernels_ms\": %.1f, \"straight_nodes_left\": %llu}\n", low.size(), kl, kh, s, ms, nodes);
        return;
    }
    for (size_t g = 0; g < low.size(); ++g) {
        pbh::GapFlanks fl;
        bool at_end = false;
        if (g_flanks) {
            int32_t se[2]; uint8_t e; uint32_t hl, tl;
            if (fread(se, 4, 2, g_flanks) != 2 || fread(&e, 1, 1, g_flanks) != 1 || fread(&hl, 4, 1, g_flanks) != 1 || fread(&tl, 4, 1, g_flanks) != 1) exit(7);
            fl.start = se[0]; fl.end = se[1]; at_end = e != 0;
            fl.head.resize(hl); fl.tail.resize(tl);
            if ((hl && fread(fl.head.data(), 1, hl, g_flanks) != hl) || (tl && fread(fl.tail.data(), 1, tl, g_flanks) != tl)) exit(7);
        }
        pbh::GapCloser cl;
        cl.max_k = kh; cl.gap = &fl;
        printf("G %zu\n", g);
        // (gapClose.cpp:716-717: a gap that is not enabled is not assembled at all; here its graphs exist -- the batch built them -- and are not shown)
        if (g_flanks && !at_end && !cl.enabled((size_t)(b.seq_first[g + 1] - b.seq_first[g]) + 2)) { printf("X\n"); continue; }
        printf("L\n");
        dump<WL>(low[g], kl);
        printf("H\n");
        dump<WH>(high[g], kh);
        if (!g_flanks) continue;
        if (!at_end) {
            pbh::GapClosure c;
            const bool closed = cl.closed(high[g], c);
            closure_line("C", closed, c, false);
            if (!closed) { pbh::GapClosure pc; const bool part = cl.closed_partial(high[g], pc); closure_line("P", part, pc, true); }
        } else {
            pbh::GapClosure c;
            const bool ext = cl.extend_edge(high[g], c);
            closure_line("E", ext, c, false);
        }
    }
}

int main(int argc, char **argv)
{
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint32_t k = 0; int64_t mc = 0; uint64_t ng = 0, ns = 0;
    if (fread(&k, 4, 1, f) != 1 || fread(&mc, 8, 1, f) != 1 || fread(&ng, 8, 1, f) != 1) return 3;
    std::vector<uint64_t> seq_first(ng + 1);
    if (fread(seq_first.data(), 8, ng + 1, f) != ng + 1 || fread(&ns, 8, 1, f) != 1) return 3;
    std::vector<uint64_t> seq_off(ns + 1);
    if (fread(seq_off.data(), 8, ns + 1, f) != ns + 1) return 3;
    std::vector<uint8_t> codes(seq_off[ns]);
    if (!codes.empty() && fread(codes.data(), 1, codes.size(), f) != codes.size()) return 3;
    fclose(f);
    pbh::GapBatch b;
    b.seq_first = seq_first; b.seq_off = seq_off; b.codes = codes;
    const unsigned kh = (unsigned)atoi(argv[2]);
    g_time_only = argc >= 4 && !strcmp(argv[3], "time");
    if (argc >= 4 && !g_time_only && !(g_flanks = fopen(argv[3], "rb"))) return 2;
    pb_ctx *ctx = nullptr;
    if (pb_create(0, &ctx) != PB_OK) { fprintf(stderr, "%s\n", pb_last_error(nullptr)); return 4; }
    try {
        const int wl = pbh::words_for_k(k), wh = pbh::words_for_k(kh);
        if (wl == 1 && wh == 2) run<1, 2>(ctx, k, kh, mc, b);
        else if (wl == 1 && wh == 3) run<1, 3>(ctx, k, kh, mc, b);
        else if (wl == 1 && wh == 4) run<1, 4>(ctx, k, kh, mc, b);
        else if (wl == 2 && wh == 3) run<2, 3>(ctx, k, kh, mc, b);
        else { fprintf(stderr, "no instantiation for this pair of k\n"); return 5; }
    } catch (const std::exception &e) { fprintf(stderr, "%s\n", e.what()); pb_destroy(ctx); return 6; }
    pb_destroy(ctx);
    return 0;
}
