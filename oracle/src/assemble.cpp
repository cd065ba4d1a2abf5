// TEST INFRASTRUCTURE ONLY -- oracle restatement of `platanus_b assemble` (assemble.cpp:116-782),
// single-thread schedule.  Stage order and every floating-point expression follow the reference
// literally: they decide the k schedule, the cutoffs and the coverage roundings.
#include "assemble.hpp"
#include "merge_run.hpp"
#include "graph.hpp"
#include "../../platanus_b_amd/host/schedule.hpp"
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <tuple>

namespace pbo {

namespace {

struct Run {
    const AssembleOptions &opt;
    ReadSet all_reads;            // allReadFP
    ReadSet reads;                // readFP (cumulatively filtered, assemble.cpp:164-171, counter.h:908-909)
    u64 memory;
    double min_log_p_join;
    double avg_len = 0, avg_cov = 0;
    std::vector<unsigned> k;
    std::vector<u64> cutoff;
    u64 num_k = 0;
    u64 dh_size = 0;
    // one counter and one graph object per key width, reused across rounds (assemble.h:55-60)
    std::tuple<Counter<1>, Counter<2>, Counter<3>, Counter<4>, Counter<5>> counters;
    std::tuple<Graph<1>, Graph<2>, Graph<3>, Graph<4>, Graph<5>> graphs;
    std::vector<ContigRecord> contigs, junctions;   // contigFP / junctionFP

    explicit Run(const AssembleOptions &o) : opt(o) {}
    template <int W> Counter<W> &counter() { return std::get<W - 1>(counters); }
    template <int W> Graph<W> &graph() { return std::get<W - 1>(graphs); }

    // assemble.cpp:304-386
    template <int W> void initial(double cutoff_factor)
    {
        Counter<W> &c = counter<W>();
        Graph<W> &g = graph<W>();
        if (opt.verbose) std::cerr << "K = " << k[0] << ", saving kmers from reads..." << std::endl;
        g.set_k(k[0]);
        cutoff.clear();
        dh_size = c.count_reads(k[0], reads, memory);
        if (!opt.repeat)
            cutoff.push_back(opt.n_cutoff != 0 ? opt.n_cutoff : std::max<u64>(c.left_local_minimal_value(1) / 2, 2ull));
        else
            cutoff.push_back(opt.n_cutoff != 0 ? opt.n_cutoff : std::max<u64>(c.left_local_minimal_value(1), 2ull));
        avg_cov = c.occ_average(cutoff[0], c.max_occ);
        if (avg_cov * cutoff_factor > cutoff.back()) cutoff.back() = avg_cov * cutoff_factor;
        avg_cov = c.occ_average(cutoff[0], c.max_occ);
        if (opt.expected_cov != "") avg_cov = atof(opt.expected_cov.c_str());
        avg_len = c.length_average(0, MAX_READ_LEN);
        avg_cov = avg_cov * avg_len / (avg_len - k[0] + 1.0);
        if (opt.verbose) std::cerr << "AVE_READ_LEN=" << avg_len << std::endl;
        num_k = extend_kmer(min_log_p_join, avg_cov, avg_len, opt.min_cov, opt.max_k_ratio, k, cutoff, opt.step, opt.verbose);

        std::ostringstream oss;
        oss << opt.out_prefix << '_' << k[0] << "merFrq.tsv";
        c.write_occ_distribution(oss.str());

        std::vector<Key<W>> sorted = c.sorted_keys(cutoff[0]);
        dh_size = c.load_kmers(cutoff[0], dh_size);
        if (opt.kmer_occ_only) {
            c.write_table_binary(opt.out_prefix + "_kmer_occ.bin");
            return;
        }
        g.make_initial(c.table, sorted);
        c.table.release();
        g.cut_branch_iterative(opt.verbose);
        if (opt.expected_cov == "") avg_cov = g.average_coverage_excluding_bubble();
        else avg_cov = atof(opt.expected_cov.c_str());
        if (opt.repeat) {
            g.delete_erroneous_iterative(ULONG_MAX, 1.75 * avg_cov + 0.5, opt.verbose);
            g.crush_bubble_iterative(DBL_MAX, opt.verbose);
        }
        avg_cov = avg_cov * avg_len / (avg_len - k[0] + 1.0);
    }

    // assemble.cpp:449-459 saveGraph
    template <int W> void save_graph(unsigned next_k, ReadSet &rfp)
    {
        Counter<W> &c = counter<W>();
        c.table.resize(dh_size);
        graph<W>().save_edge_kmers([&](const Key<W> &key, u16 v) { c.table.put(key, v); }, next_k);
        if (opt.verbose) std::cerr << "extracting reads (containing kmer used in contig assemble)..." << std::endl;
        rfp = c.pickup_reads_matching_table(rfp);
        c.table.release();
    }

    // assemble.cpp:762-782
    template <int WP, int WN> void update_dh_size()
    {
        const u64 estimate = graph<WP>().estimate_kmers_on_straight();
        const u64 size = table_size_for_keys(estimate);
        dh_size = std::max(size, table_size_for_memory(memory, WN));
    }

    // assemble.cpp:467-498
    template <int W> void redo(unsigned kk, u64 position)
    {
        Counter<W> &c = counter<W>();
        Graph<W> &g = graph<W>();
        if (opt.verbose) std::cerr << "K = " << kk << ", saving additional kmers(not found in contigs) from reads..." << std::endl;
        c.count_reads_considering_table(kk, reads);
        if (opt.verbose) std::cerr << "COVERAGE_CUTOFF = " << cutoff[position] << std::endl;
        std::vector<Key<W>> sorted = c.sorted_keys(cutoff[position]);
        dh_size = c.load_kmers(cutoff[position], dh_size);
        g.set_k(kk);
        g.make_initial(c.table, sorted);
        c.table.release();
        g.cut_branch_iterative(opt.verbose);
        const unsigned kmer_cov = avg_cov * (avg_len - kk + 1.0) / avg_len + 0.5;
        if (opt.repeat) {
            g.delete_erroneous_iterative(ULONG_MAX, 1.75 * kmer_cov + 0.5, opt.verbose);
            g.crush_bubble_iterative(DBL_MAX, opt.verbose);
        }
    }

    // assemble.cpp:392-443
    template <int WP, int WN> void save_and_redo(unsigned kk, unsigned prev_k, u64 position)
    {
        save_graph<WP>(kk, reads);
        update_dh_size<WP, WN>();
        DHash<WN> tmp;
        tmp.resize(dh_size);
        graph<WP>().template save_contig<WN>(kk, (avg_len - kk + 1.0) / (avg_len - prev_k + 1.0), [&](const Key<WN> &key, u16 occ) {
            const u64 s = tmp.find_any(key);
            if (tmp.vals[s] < occ) { tmp.vals[s] = occ; tmp.keys[s] = key; }
        });
        graph<WP>().delete_all();
        if (opt.verbose) std::cerr << "K = " << kk << ", loading kmers from contigs..." << std::endl;
        counter<WN>().table = std::move(tmp);
        redo<WN>(kk, position);
    }

    // assemble.cpp:506-622
    template <int W> void output_and_after(unsigned kk)
    {
        Counter<W> &c = counter<W>();
        Graph<W> &g = graph<W>();
        const u64 left_min = g.left_minimal_coverage();
        const u64 length_cutoff = kk * 2;
        if (opt.verbose) {
            std::cerr << "LENGTH_CUTOFF = " << length_cutoff << std::endl;
            std::cerr << "COVERAGE_CUTOFF = " << left_min << std::endl;
        }
        g.delete_erroneous_iterative(length_cutoff, left_min, opt.verbose);
        avg_cov = avg_cov * (avg_len - kk + 1.0) / avg_len;
        if (atoi(opt.bubble.c_str()) > 0) {                     // assemble.cpp:520-526 (Q1: only -u >= 1 gets here)
            if (opt.verbose) std::cerr << "AVE_KMER_COV_REMOVING_BUBBLE=" << avg_cov << std::endl;
            g.crush_bubble_iterative(avg_cov, opt.verbose);
            print_bubbles(opt.out_prefix + "_contigBubble.fa", g.bubbles, avg_len / (avg_len - kk + 1.0));
        }
        save_graph<W>(kk, all_reads);          // diff == 0: inserts nothing, keeps no read
        if (W > 1) dh_size = table_size_for_memory(memory, W);
        DHash<W> tmp;
        tmp.resize(dh_size);
        g.template save_contig<W>(kk, 1.0, [&](const Key<W> &key, u16 occ) {
            const u64 s = tmp.find_any(key);
            if (tmp.vals[s] < occ) { tmp.vals[s] = occ; tmp.keys[s] = key; }
        });
        g.delete_all();
        c.table = std::move(tmp);
        c.count_reads_considering_table(kk, all_reads);
        std::vector<Key<W>> sorted = c.sorted_keys(cutoff[num_k - 1]);
        dh_size = c.load_kmers(cutoff[num_k - 1], dh_size);
        g.set_k(kk);
        g.make_initial(c.table, sorted);
        g.save_contig_simple(contigs, avg_len / (avg_len - kk + 1.0));
        g.save_junction(junctions, avg_len / (avg_len - kk + 1.0));
    }
};

template <typename F> void dispatch_w(unsigned k, F &&f)
{
    if (k <= 32) f(std::integral_constant<int, 1>());
    else if (k <= 64) f(std::integral_constant<int, 2>());
    else if (k <= 96) f(std::integral_constant<int, 3>());
    else if (k <= 128) f(std::integral_constant<int, 4>());
    else if (k <= 160) f(std::integral_constant<int, 5>());
    else throw std::runtime_error("k > 160 is outside the oracle's scope");
}

} // namespace

int run_assemble(const AssembleOptions &opt, const ReadSet &input)
{
    Run run(opt);
    run.memory = opt.memory_gb * 1000000000ull;
    run.min_log_p_join = log(1.0 - pow(10.0, -opt.safety));
    double bubble = atof(opt.bubble.c_str());
    if (opt.repeat) bubble = 0.95;
    std::get<0>(run.graphs).set_thresholds(bubble, opt.branch);
    std::get<1>(run.graphs).set_thresholds(bubble, opt.branch);
    std::get<2>(run.graphs).set_thresholds(bubble, opt.branch);
    std::get<3>(run.graphs).set_thresholds(bubble, opt.branch);
    std::get<4>(run.graphs).set_thresholds(bubble, opt.branch);
    // assemble.cpp:148-240: -repeat runs the whole multi-k assembly three times with the coverage cutoff raised to 0, 1 and 2
    // times the average coverage; the contigs of all passes are then merged through one more graph at the largest k reached
    const int max_factor = opt.repeat ? 2 : 0;
    unsigned max_k = 0;
    for (int factor = 0; factor <= max_factor; ++factor) {
        run.all_reads = input;
        run.reads = input;
        run.k.assign(1, opt.k0);
        dispatch_w(opt.k0, [&](auto w) { run.template initial<decltype(w)::value>((double)factor); });
        if (opt.kmer_occ_only) {
            if (opt.verbose) std::cerr << "assemble completed!" << std::endl;
            return 0;
        }
        for (u64 i = 1; i < run.num_k; ++i) {
            const unsigned kp = run.k[i - 1], kn = run.k[i];
            dispatch_w(kp, [&](auto wp) {
                dispatch_w(kn, [&](auto wn) {
                    constexpr int WP = decltype(wp)::value, WN = decltype(wn)::value;
                    if constexpr (WN >= WP) run.template save_and_redo<WP, WN>(kn, kp, i);
                });
            });
        }
        const unsigned klast = run.k[run.num_k - 1];
        dispatch_w(klast, [&](auto w) { run.template output_and_after<decltype(w)::value>(klast); });
        max_k = std::max(max_k, klast);
    }
    const std::string out = opt.out_prefix + "_contig.fa";
    if (!opt.repeat) {
        print_contigs(out, run.contigs, 1.0, run.avg_len, max_k, "seq", false);
        print_contigs(out, run.junctions, 1.0, run.avg_len, max_k, "junction", true);
    } else {
        // mergeContig (assemble.cpp:270-295): Contig::readContigFP takes every seq record of the three passes as it was written
        pbh::MergeContigs mc;
        pbh::contig_records_to_pieces(run.contigs, mc);
        const double merge_cov = pbh::merge_average_coverage(mc);
        std::vector<ContigRecord> merged, merged_junctions;
        dispatch_w(max_k, [&](auto w) {
            constexpr int W = decltype(w)::value;
            merge_contigs_into_graph<W>(run.template graph<W>(), mc, max_k, 1, run.memory, merge_cov, run.avg_len / (run.avg_len - max_k + 1.0), opt.verbose,
                                        merged, merged_junctions);
        });
        print_contigs(out, merged, 1.0, run.avg_len, max_k, "seq", false);
    }
    if (opt.verbose) std::cerr << "assemble completed!" << std::endl;
    return 0;
}

} // namespace pbo
