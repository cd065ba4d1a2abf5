// `platanus_b merge` on the device path (merge.cpp:86-147).
#include "merge.hpp"
#include "device_handover.hpp"
#include <iostream>

extern "C" {
#include "../../include/pbgpu.h"
}

namespace pbh {

template <int W>
static void merge_run(pb_ctx *ctx, const MergeOptions &opt, const MergeContigs &mc, unsigned k, double avg_cov, u64 read_length, bool verbose)
{
    auto chk = [&](int rc) { if (rc != PB_OK) throw std::runtime_error(std::string("libpbgpu: ") + pb_last_error(ctx)); };
    // makeKmerReadDistributionFromContig: every k-mer of every piece, value = max(coverage, -c)
    if (verbose) std::cerr << "K = " << k << ", loading kmers from contigs...\n";
    std::string bases;
    std::vector<u64> off{0};
    std::vector<u16> vals;
    for (size_t i = 0; i < mc.seq.size(); ++i) {
        bases += mc.seq[i];
        off.push_back(bases.size());
        vals.push_back((u16)std::max<u64>(mc.coverage[i], opt.min_cov));
    }
    u64 nw = 0, nd = 0, max_occ = 0, n = 0;
    std::vector<u64> hist(PB_OCC_BINS);
    chk(pb_contig_table_build(ctx, k, (const uint8_t *)bases.data(), off.data(), vals.data(), vals.size(), &nw));
    chk(pb_count_considering_contigs(ctx, k, hist.data(), &max_occ, &nd));        // no reads resident: just the contig table
    if (verbose) std::cerr << "loading kmers..." << std::endl;
    chk(pb_retain(ctx, 0, &n));                                                    // sortedKeyFromKmerFile(0) + loadKmer(0, ...)
    (void)n;
    if (verbose) std::cerr << "connecting kmers..." << std::endl;
    Graph<W> g;
    g.set_k(k);
    g.set_thresholds(atof(opt.bubble.c_str()), opt.branch);
    // the order-independent decisions of cutBranch / crushBubble are taken on the device (node_kernels.inc), as in `assemble`
    g.clean_decide = [&](unsigned kk, u64 n_s, const u64 *head, const u64 *tail, const u64 *length, const u16 *cov, const uint8_t *out, u64 n_t, const u64 *tkeys,
                         const uint32_t *tids, u64 n_j, const u64 *jkeys, const u16 *jcov, const uint8_t *jout, double th, u64 len_cut, u64 cov_cut, uint8_t *dec) {
        chk(pb_nodes_clean_decide(ctx, kk, n_s, (const uint64_t *)head, (const uint64_t *)tail, (const uint64_t *)length, cov, out, n_t, (const uint64_t *)tkeys, tids, n_j,
                                  (const uint64_t *)jkeys, jcov, jout, th, len_cut, cov_cut, dec));
        return true;
    };
    g.edit_distances = [&](const uint8_t *b, const u64 *offsets, u64 n_pairs, u64 *dist_out) {
        chk(pb_edit_distances(ctx, b, (const uint64_t *)offsets, n_pairs, (uint64_t *)dist_out));
        return true;
    };
    // the same hand-over as `assemble` (device_handover.hpp): nodes + chain records, every device-produced index checked before use
    graph_from_device<W>(ctx, g, k, chk);
    g.cut_branch_iterative(verbose);
    g.crush_bubble_iterative(avg_cov, verbose);
    std::vector<ContigRecord> contigs, junctions;
    g.save_contig_simple(contigs, 1.0);
    g.save_junction(junctions, 1.0);
    print_contigs(opt.out_prefix + "_merged.fa", contigs, 1.0, (double)read_length, (long)k, "seq", false);
    print_contigs(opt.out_prefix + "_mergedJunctionKmer.fa", junctions, 1.0, (double)read_length, (long)k, "junction", false);
}

int run_merge_gpu(const MergeOptions &opt, int device, bool verbose)
{
    if (opt.files.empty()) throw std::runtime_error("not specified contig file");
    const u64 read_length = first_header_read_length(opt.files[0]);
    const unsigned k = (unsigned)(opt.k_factor * read_length + 0.5);
    const u64 min_len = (u64)(opt.len_factor * read_length + 0.5);
    if (k < 1 || k > PB_MAX_K) throw std::runtime_error("k out of range for the device path (1..160)");
    MergeContigs mc;
    for (const std::string &f : opt.files) read_contigs_cut_n(f, min_len, mc);
    if (mc.seq.empty()) throw std::runtime_error("no contig passes the length filter");
    const double avg_cov = merge_average_coverage(mc);
    pb_ctx *ctx = nullptr;
    if (pb_create(device, &ctx) != PB_OK) throw std::runtime_error(std::string("libpbgpu: ") + pb_last_error(nullptr));
    try {
        switch (words_for_k(k)) {
        case 1: merge_run<1>(ctx, opt, mc, k, avg_cov, read_length, verbose); break;
        case 2: merge_run<2>(ctx, opt, mc, k, avg_cov, read_length, verbose); break;
        case 3: merge_run<3>(ctx, opt, mc, k, avg_cov, read_length, verbose); break;
        case 4: merge_run<4>(ctx, opt, mc, k, avg_cov, read_length, verbose); break;
        default: merge_run<5>(ctx, opt, mc, k, avg_cov, read_length, verbose); break;
        }
    } catch (...) { pb_destroy(ctx); throw; }
    pb_destroy(ctx);
    return 0;
}

} // namespace pbh
