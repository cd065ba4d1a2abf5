// `platanus_b assemble` host driver on top of libpbgpu (include/pbgpu.h).
//
// Control flow, option handling, floating-point expressions and stderr progress lines follow the reference's
// Assemble::exec / initialKmerAssemble / saveAndRedoAssemble / outputAndAfterTreatment (assemble.cpp:140-622); every
// Counter / BruijnGraph call whose cost is per k-mer or per read window is a C-ABI call into the HIP library, the
// ~10^4-node cleaning steps stay on the host (node_graph.hpp).  No CPU fallback: without the device the constructor fails.
#include "assemble_gpu.hpp"
#include "device_handover.hpp"
#include <memory>
#include "schedule.hpp"
#include "read_files.hpp"

#include <cfloat>
#include <chrono>
#include <map>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <tuple>
#include <thread>
#include <sys/mman.h>
#include <sys/stat.h>

extern "C" {
#include "../../include/pbgpu.h"
}

namespace pbh {

namespace {

struct GpuError : std::runtime_error {
    int code;
    GpuError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

struct StageClock {
    std::map<std::string, double> acc;
    std::vector<std::string> order;
    void add(const std::string &name, double s) { if (!acc.count(name)) order.push_back(name); acc[name] += s; }
};
static StageClock g_clock;
struct Tick {
    std::string name;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool live = true;
    explicit Tick(const std::string &n) : name(n) {}
    void stop() { if (live) g_clock.add(name, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); live = false; }
    ~Tick() { stop(); }
};

struct Driver {
    const GpuAssembleOptions &opt;
    pb_ctx *ctx = nullptr;
    double min_log_p_join = 0;
    double avg_len = 0, avg_cov = 0;
    std::vector<unsigned> k;
    std::vector<u64> cutoff;
    u64 num_k = 0;
    std::tuple<Graph<1>, Graph<2>, Graph<3>, Graph<4>, Graph<5>> graphs;       // reused across rounds (assemble.h:55-60)
    std::vector<ContigRecord> contigs, junctions;
    std::vector<u64> occ_hist = std::vector<u64>(PB_OCC_BINS);
    u64 max_occ = 0;

    // The device context (HIP runtime start-up, ~0.2 s) is created on a helper thread while the input files are parsed.
    std::thread creator;
    int create_rc = PB_OK;
    double create_s = 0;
    explicit Driver(const GpuAssembleOptions &o) : opt(o)
    {
        creator = std::thread([this]() {
            const auto t0 = std::chrono::steady_clock::now();
            create_rc = pb_create(opt.device, &ctx);
            create_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    void ready()                                   // before the first call that takes ctx
    {
        if (!creator.joinable()) return;
        creator.join();
        if (create_rc != PB_OK) throw GpuError(create_rc, std::string("libpbgpu: ") + pb_last_error(nullptr));
        for (const auto &nv : opt.gpu_options) chk(pb_set_option(ctx, nv.first.c_str(), nv.second));
        if (dist()) chk(pb_dist_attach(ctx, opt.shared, opt.rank, opt.world, opt.rccl ? 1 : 0));
        hook(std::get<0>(graphs)); hook(std::get<1>(graphs)); hook(std::get<2>(graphs)); hook(std::get<3>(graphs)); hook(std::get<4>(graphs));
    }
    // the order-independent decisions of the cleaning steps are taken on the device (node_kernels.inc); the graph applies them
    template <int W> void hook(Graph<W> &g)
    {
        g.clean_decide = [this](unsigned kk, u64 n_s, const u64 *head, const u64 *tail, const u64 *length, const u16 *cov, const uint8_t *out, u64 n_t, const u64 *tkeys,
                                const uint32_t *tids, u64 n_j, const u64 *jkeys, const u16 *jcov, const uint8_t *jout, double th, u64 len_cut, u64 cov_cut, uint8_t *dec) {
            static_assert(sizeof(u64) == sizeof(uint64_t), "key words are 64-bit");
            chk(pb_nodes_clean_decide(ctx, kk, n_s, (const uint64_t *)head, (const uint64_t *)tail, (const uint64_t *)length, cov, out, n_t, (const uint64_t *)tkeys, tids, n_j,
                                      (const uint64_t *)jkeys, jcov, jout, th, len_cut, cov_cut, dec));
            return true;
        };
        g.edit_distances = [this](const uint8_t *bases, const u64 *offsets, u64 n_pairs, u64 *dist_out) {
            chk(pb_edit_distances(ctx, bases, (const uint64_t *)offsets, n_pairs, (uint64_t *)dist_out));
            return true;
        };
    }
    // The node tables of a finished round are emptied (millions of nodes freed one by one: a second per round at 100 Mbp) by a helper thread
    // beside the device stages of the next round; joined before the next replay touches a graph.
    std::thread bg;
    double bg_s = 0;
    std::exception_ptr bg_err;
    template <typename F> void background(F f)
    {
        join_background();
        bg = std::thread([this, f]() {
            const auto t0 = std::chrono::steady_clock::now();
            try { f(); } catch (...) { bg_err = std::current_exception(); }
            bg_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        });
    }
    void join_background()
    {
        if (!bg.joinable()) return;
        {
            Tick t("wait for the graph reset (host)");
            bg.join();
        }
        g_clock.add("graph reset (host thread, beside the device stages)", bg_s);
        if (bg_err) { std::exception_ptr e = bg_err; bg_err = nullptr; std::rethrow_exception(e); }
    }
    bool dist() const { return opt.world > 1; }    // N ranks on this run: the table is merged over them before every graph stage
    bool writer() const { return opt.rank == 0; }  // every rank computes the same graph; rank 0 writes the files
    ~Driver()
    {
        if (creator.joinable()) creator.join();
        if (bg.joinable()) bg.join();
        pb_destroy(ctx);
    }
    void chk(int rc) const { if (rc != PB_OK) throw GpuError(rc, std::string("libpbgpu: ") + pb_last_error(ctx)); }
    template <int W> Graph<W> &graph() { return std::get<W - 1>(graphs); }
    void log(const std::string &s) const { if (opt.verbose) std::cerr << s << std::endl; }

    // sortedKeyFromKmerFile + loadKmer + makeInitialBruijnGraph (assemble.cpp:343-358 / 474-478 / 613-617)
    template <int W> void retain_and_build(unsigned kk, u64 min_occ, bool shards = true)
    {
        u64 n = 0;
        log("loading kmers...");
        std::vector<u64> keys;
        std::vector<u16> counts;
        std::vector<uint8_t> adj;
        std::vector<uint32_t> nbr;
        {
            Tick t("retain+sort (device)");
            chk(pb_retain(ctx, (unsigned)min_occ, &n));
        }
        if (dist() && shards) {
            // the table is spread over the ranks by owner stripe: every rank takes in the retained part of all shards (one
            // table feeds loadKmer and makeInitialBruijnGraph, assemble.cpp:343-354)
            Tick t("gather of the retained shards (ranks)");
            chk(pb_dist_gather_retained(ctx, &n));
        }
        Graph<W> &g = graph<W>();
        g.set_k(kk);
        if (!getenv("PB_GRAPH_FETCH")) {
            // compact hand-over: adjacency and chain arrays stay on the device, only the nodes the DFS can land on come back
            std::unique_ptr<Tick> t;
            graph_from_device<W>(ctx, g, kk, [this](int rc) { chk(rc); }, [&](const char *stage) {
                if (!strncmp(stage, "adjacency", 9)) log("connecting kmers...");
                t.reset();
                if (!strncmp(stage, "DFS", 3)) join_background();      // the replay starts with the tables of this width
                t.reset(new Tick(stage));
            });
            return;
        }
        // development / A-B: the full per-k-mer arrays on the host (PB_GRAPH_FETCH=full, PB_NO_CHAINS=1 for the k-mer-level replay)
        join_background();
        {
            Tick t("fetch retained + adjacency (device + D2H)");
            keys.resize(n * W); counts.resize(n); adj.resize(n); nbr.resize(2 * n);
            chk(pb_fetch_retained(ctx, keys.data(), counts.data()));
            log("connecting kmers...");
            chk(pb_graph_adjacency(ctx, adj.data(), nbr.data()));
        }
        DeviceGraphInput<W> in;
        in.k = kk; in.n = n; in.keys = keys.data(); in.counts = counts.data(); in.adj = adj.data(); in.nbr = nbr.data();
        if (getenv("PB_NO_CHAINS")) {                 // development switch: k-mer-level replay of the DFS
            Tick t("DFS replay, k-mer level (host)");
            make_initial_from_device(g, in);
            return;
        }
        std::vector<uint32_t> chain_of(n), ch_len, ch_end_a, ch_end_b;
        std::vector<uint8_t> orient(n);
        std::vector<u64> ch_sum, ch_woff, ch_seq;
        ChainSet cs;
        {
            Tick t("chain compaction (device + D2H)");
            u64 nc = 0, words = 0;
            chk(pb_graph_chains(ctx, &nc, &words));
            ch_len.resize(nc); ch_end_a.resize(nc); ch_end_b.resize(nc); ch_sum.resize(nc); ch_woff.resize(nc); ch_seq.resize(words);
            chk(pb_graph_fetch_chains(ctx, chain_of.data(), orient.data(), ch_len.data(), ch_sum.data(), ch_end_a.data(), ch_end_b.data(),
                                      ch_woff.data(), ch_seq.data()));
            cs.n_chains = nc; cs.chain_of = chain_of.data(); cs.orient = orient.data(); cs.len = ch_len.data(); cs.sum = ch_sum.data();
            cs.end_a = ch_end_a.data(); cs.end_b = ch_end_b.data(); cs.seq_woff = ch_woff.data(); cs.seq = ch_seq.data();
        }
        Tick t("DFS replay, chain level (host)");
        make_initial_from_chains(g, in, cs);
    }

    // assemble.cpp:304-386
    template <int W> void initial()
    {
        Graph<W> &g = graph<W>();
        if (opt.verbose) std::cerr << "K = " << k[0] << ", saving kmers from reads..." << std::endl;
        cutoff.clear();
        u64 n_inst = 0, n_distinct = 0;
        {
            Tick t(dist() ? "count k0 (device, all ranks)" : "count k0 (device)");
            // N ranks: every rank's reads, one global table spread over the ranks (run records exchanged by bucket owner and counted on
            // arrival; or count + owner exchange of (key,count) records + merge)
            if (dist()) chk(pb_dist_count(ctx, k[0], occ_hist.data(), &max_occ, &n_inst, &n_distinct));
            else chk(pb_count(ctx, k[0], occ_hist.data(), &max_occ, &n_inst, &n_distinct));
        }
        if (dist()) {                                  // (PB_TIMING: which of the two designs the ranks agreed on)
            float by_records = 0.f;
            (void)pb_last_kernel_ms(ctx, "dist_by_records", &by_records);
            g_clock.add(by_records > 0.f ? "  (k0: run records exchanged before the counting)" : "  (k0: counted per rank, (key,count) tables exchanged and merged)", 0.0);
        }
        const u64 llm = left_local_minimal_value(occ_hist, max_occ, 1);
        if (!opt.repeat) cutoff.push_back(opt.n_cutoff != 0 ? opt.n_cutoff : std::max<u64>(llm / 2, 2ull));
        else cutoff.push_back(opt.n_cutoff != 0 ? opt.n_cutoff : std::max<u64>(llm, 2ull));
        avg_cov = dist_average(occ_hist, cutoff[0], max_occ);
        if (opt.expected_cov != "") avg_cov = atof(opt.expected_cov.c_str());
        std::vector<u64> len_hist(500001);
        {
            Tick t("read length histogram (device)");
            chk(pb_reads_length_hist(ctx, len_hist.data(), len_hist.size()));
            if (dist()) chk(pb_dist_allreduce_u64(ctx, len_hist.data(), len_hist.size()));
        }
        avg_len = dist_average(len_hist, 0, 500000);
        avg_cov = avg_cov * avg_len / (avg_len - k[0] + 1.0);
        if (opt.verbose) std::cerr << "AVE_READ_LEN=" << avg_len << std::endl;
        num_k = extend_kmer(min_log_p_join, avg_cov, avg_len, opt.min_cov, opt.max_k_ratio, k, cutoff, opt.step, opt.verbose);
        // the device keys have at most five words (KmerN<Binstr159>, assemble.cpp:174-185); the reference's sixth type (binstr_t,
        // k > 160) needs reads of 321 bases or more at the default -K 0.5: refused here, before any graph work, not in the middle
        if (!k.empty() && k.back() > 160)
            throw std::runtime_error("the k schedule reaches k = " + std::to_string(k.back()) + " (-K times the average read length); this build supports k <= 160: "
                                     "lower -K or give a smaller maximum k");

        std::ostringstream oss;
        oss << opt.out_prefix << '_' << k[0] << "merFrq.tsv";
        if (writer()) {
            std::ofstream ofs(oss.str().c_str());                                  // counter.h:999-1007
            for (u64 i = 1; i <= max_occ; ++i) ofs << i << "\t" << occ_hist[i] << std::endl;
        }
        if (opt.kmer_occ_only) {
            write_kmer_occ<W>(cutoff[0]);
            return;
        }
        retain_and_build<W>(k[0], cutoff[0]);
        {
            Tick t("node cleaning (host)");
            g.cut_branch_iterative(opt.verbose);
        }
        if (opt.expected_cov == "") avg_cov = g.average_coverage_excluding_bubble();
        else avg_cov = atof(opt.expected_cov.c_str());
        avg_cov = avg_cov * avg_len / (avg_len - k[0] + 1.0);
    }

    // PREFIX_kmer_occ.bin (counter.h:954-963, doubleHash.h:266-278).  k <= 32: the device reproduces the reference's
    // -t 1 slot layout (pb_kmer_occ_layout), the file is byte-identical.  k > 32: the reference dumps raw C++ objects
    // (vptr + heap pointer per record) and is not reproducible even by itself; the file written here has the
    // same header, table size, record format (pointers zeroed) and (key,count) set, keys inserted in ascending order.
    template <int W> void write_kmer_occ(u64 min_occ)
    {
        if (W == 1) {
            const u64 memory = opt.memory_gb * 1000000000ull;
            long base = 16;                                                       // sizeof(pair<u64,u16>), counter.h:300
            u64 dh = memory / base;
            base = std::log(dh) / std::log(2);
            dh = std::pow(2, base);
            while (dh > memory) dh >>= 1;
            u64 tsize = 0, nrec = 0;
            {
                Tick t("kmer_occ layout (device)");
                chk(pb_kmer_occ_layout(ctx, dh, (unsigned)min_occ, &tsize, &nrec));
            }
            std::vector<u64> slots(nrec), keys(nrec);
            std::vector<u16> counts(nrec);
            chk(pb_kmer_occ_fetch(ctx, slots.data(), keys.data(), counts.data()));
            Tick t("kmer_occ write");
            FILE *f = fopen((opt.out_prefix + "_kmer_occ.bin").c_str(), "wb");
            if (!f) throw file_error(opt.out_prefix + "_kmer_occ.bin");          // counter.h:958
            const u64 kk = k[0], nm = tsize - 1;
            fwrite(&kk, 8, 1, f);
            fwrite(&nm, 8, 1, f);
            std::vector<unsigned char> buf(nrec * 18);
            for (u64 i = 0; i < nrec; ++i) {
                memcpy(&buf[i * 18], &slots[i], 8);
                memcpy(&buf[i * 18 + 8], &keys[i], 8);
                memcpy(&buf[i * 18 + 16], &counts[i], 2);
            }
            if (nrec) fwrite(buf.data(), 18, nrec, f);
            fclose(f);
            return;
        }
        u64 n = 0;
        chk(pb_retain(ctx, (unsigned)min_occ, &n));
        std::vector<u64> keys(n * W);
        std::vector<u16> counts(n);
        chk(pb_fetch_retained(ctx, keys.data(), counts.data()));
        static const u64 pair_bytes[6] = {0, 16, 48, 56, 64, 72};
        const u64 memory = opt.memory_gb * 1000000000ull;
        long base = (long)pair_bytes[W];
        u64 tmp = memory / base;
        base = std::log(tmp) / std::log(2);
        tmp = std::pow(2, base);
        while (tmp > memory) tmp >>= 1;
        u64 size = n ? (u64)(std::log(n / 0.9) / std::log(2)) : 0;      // (no record: log(0) is not taken; the table has the -m size)
        size = n ? (u64)std::pow(2, size + 1) : 0;
        const u64 slots = std::max(size, tmp);
        const u64 nmask = slots - 1;
        unsigned index_len = 64;
        for (unsigned i = 1; i < 64; ++i) if ((slots >> i) == 0) { index_len = i; break; }
        const unsigned shifter = index_len >= 32 ? 0 : 2 * index_len;
        std::vector<uint32_t> slot_of(slots, 0);                                 // 0 = empty, else record index + 1
        for (u64 i = 0; i < n; ++i) {
            u64 h = 0, st = 0;
            for (int w = 0; w < W; ++w) {
                const u64 v = keys[i * W + w];
                h += v + (v >> index_len) + (v >> shifter);
                st += ~v ^ (v >> index_len) ^ (v >> shifter);
            }
            h &= nmask;
            st |= 1;
            while (slot_of[h]) h = (h + st) & nmask;
            slot_of[h] = (uint32_t)(i + 1);
        }
        FILE *f = fopen((opt.out_prefix + "_kmer_occ.bin").c_str(), "wb");
        if (!f) throw file_error(opt.out_prefix + "_kmer_occ.bin");
        const u64 kk = k[0];
        fwrite(&kk, 8, 1, f);
        fwrite(&nmask, 8, 1, f);
        for (u64 s = 0; s < slots; ++s) {
            if (!slot_of[s]) continue;
            const u64 i = slot_of[s] - 1;
            fwrite(&s, 8, 1, f);
            if (W > 1) { const u64 z[3] = {0, 0, kk}; fwrite(z, 8, 3, f); }
            fwrite(&keys[i * W], 8, W, f);
            fwrite(&counts[i], 2, 1, f);
        }
        fclose(f);
    }

    // assemble.cpp:449-459 saveGraph (edge k-mers of the finished graph -> read filter on the device) and saveContig<NEXT> (graph.h:1655-1771:
    // strings + values go to the device, which derives the next-k table).  Both only read the graph: the strings are collected by a helper
    // thread while this one lists the edge k-mers and the device filters the reads.
    template <int W> void save_graph_and_contig_table(unsigned graph_k, unsigned next_k, double ratio)
    {
        // leftStraightTable's sequence cut into runs: four threads collect the strings, this one and a fifth list the edge k-mers; every
        // result is put together in the table's order (large graphs only: a 5 Mbp genome has 10^4 nodes and no use for threads)
        Graph<W> &g = graph<W>();
        const unsigned n_str = g.left.size() >= (1u << 16) ? 4u : 1u, n_edge = g.left.size() >= (1u << 16) ? 2u : 1u;
        const auto cuts_s = g.left_cuts(n_str), cuts_e = n_edge == n_str ? cuts_s : g.left_cuts(n_edge);
        struct Strings { std::vector<uint8_t> bases; std::vector<u64> off; std::vector<u16> vals; std::exception_ptr err; double s = 0; };
        std::vector<Strings> part(n_str);
        struct Joiner { std::vector<std::thread> t; ~Joiner() { for (auto &x : t) if (x.joinable()) x.join(); } } workers;
        for (unsigned i = 0; i < n_str; ++i)
            workers.t.emplace_back([&, i]() {
                const auto t0 = std::chrono::steady_clock::now();
                Strings &o = part[i];
                try {
                    g.for_each_contig_string_range(cuts_s[i], cuts_s[i + 1], next_k, ratio, [&](const std::vector<unsigned char> &str, u16 occ) {
                        o.bases.insert(o.bases.end(), str.begin(), str.end());
                        o.off.push_back(o.bases.size());
                        o.vals.push_back(occ);
                    });
                } catch (...) { o.err = std::current_exception(); }
                o.s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            });
        {
            std::vector<std::vector<u64>> edge(n_edge);
            std::vector<std::exception_ptr> eerr(n_edge);
            Tick t0("edge k-mers (host)");
            {
                Joiner helpers;
                for (unsigned i = 1; i < n_edge; ++i)
                    helpers.t.emplace_back([&, i]() {
                        try { g.save_edge_kmers_range(cuts_e[i], cuts_e[i + 1], [&](const Key<W> &key, u16) { for (int w = 0; w < W; ++w) edge[i].push_back(key.w[w]); }, next_k); }
                        catch (...) { eerr[i] = std::current_exception(); }
                    });
                g.save_edge_kmers_range(cuts_e[0], cuts_e[1], [&](const Key<W> &key, u16) { for (int w = 0; w < W; ++w) edge[0].push_back(key.w[w]); }, next_k);
            }
            for (auto &e : eerr) if (e) std::rethrow_exception(e);
            for (unsigned i = 1; i < n_edge; ++i) edge[0].insert(edge[0].end(), edge[i].begin(), edge[i].end());
            log("extracting reads (containing kmer used in contig assemble)...");
            u64 kept = 0;
            t0.stop();
            Tick t("read filter (device)");
            chk(pb_reads_filter(ctx, graph_k, edge[0].data(), edge[0].size() / W, &kept));
        }
        {
            Tick t("wait for the contig strings (host)");
            for (auto &x : workers.t) x.join();
        }
        double strings_s = 0;
        for (auto &o : part) { if (o.err) std::rethrow_exception(o.err); strings_s = std::max(strings_s, o.s); }
        g_clock.add("contig strings (host threads, beside edge k-mers + read filter)", strings_s);
        std::vector<uint8_t> bases;
        std::vector<u64> off{0};
        std::vector<u16> vals;
        {
            Tick t("contig strings put together (host)");
            size_t nb = 0, nv = 0;
            for (auto &o : part) { nb += o.bases.size(); nv += o.vals.size(); }
            bases.reserve(nb); off.reserve(nv + 1); vals.reserve(nv);
            for (auto &o : part) {
                const u64 base = bases.size();
                bases.insert(bases.end(), o.bases.begin(), o.bases.end());
                for (u64 v : o.off) off.push_back(base + v);
                vals.insert(vals.end(), o.vals.begin(), o.vals.end());
                std::vector<uint8_t>().swap(o.bases);
            }
        }
        u64 nw = 0;
        Tick t("contig table (device)");
        chk(pb_contig_table_build(ctx, next_k, bases.data(), off.data(), vals.data(), vals.size(), &nw));
    }

    // assemble.cpp:467-498 redoAssemble
    template <int W> void redo(unsigned kk, u64 position)
    {
        if (opt.verbose) std::cerr << "K = " << kk << ", saving additional kmers(not found in contigs) from reads..." << std::endl;
        u64 nd = 0;
        {
            Tick t("count missing k-mers (device)");
            if (dist()) chk(pb_dist_count_considering_contigs(ctx, kk, occ_hist.data(), &max_occ, &nd));
            else chk(pb_count_considering_contigs(ctx, kk, occ_hist.data(), &max_occ, &nd));
        }
        if (opt.verbose) std::cerr << "COVERAGE_CUTOFF = " << cutoff[position] << std::endl;
        retain_and_build<W>(kk, cutoff[position]);
        Tick t("node cleaning (host)");
        graph<W>().cut_branch_iterative(opt.verbose);
    }

    // assemble.cpp:392-443
    template <int WP, int WN> void save_and_redo(unsigned kk, unsigned prev_k, u64 position)
    {
        save_graph_and_contig_table<WP>(prev_k, kk, (avg_len - kk + 1.0) / (avg_len - prev_k + 1.0));
        background([this]() { graph<WP>().delete_all(); });
        if (opt.verbose) std::cerr << "K = " << kk << ", loading kmers from contigs..." << std::endl;
        redo<WN>(kk, position);
    }

    // assemble.cpp:506-622
    template <int W> void output_and_after(unsigned kk)
    {
        Graph<W> &g = graph<W>();
        const u64 left_min = g.left_minimal_coverage();
        const u64 length_cutoff = kk * 2;
        if (opt.verbose) {
            std::cerr << "LENGTH_CUTOFF = " << length_cutoff << std::endl;
            std::cerr << "COVERAGE_CUTOFF = " << left_min << std::endl;
        }
        {
            Tick t("node cleaning (host)");
            g.delete_erroneous_iterative(length_cutoff, left_min, opt.verbose);
        }
        avg_cov = avg_cov * (avg_len - kk + 1.0) / avg_len;
        if (atoi(opt.bubble.c_str()) > 0) {                     // assemble.cpp:520-526 (Q1: only -u >= 1 gets here)
            if (opt.verbose) std::cerr << "AVE_KMER_COV_REMOVING_BUBBLE=" << avg_cov << std::endl;
            g.crush_bubble_iterative(avg_cov, opt.verbose);
            if (writer()) print_bubbles(opt.out_prefix + "_contigBubble.fa", g.bubbles, avg_len / (avg_len - kk + 1.0));
        }
        // saveGraph(k, allReadFP): diff == 0, so no edge k-mer is stored and no read survives
        chk(pb_reads_restore(ctx));
        save_graph_and_contig_table<W>(kk, kk, 1.0);
        background([&g]() { g.delete_all(); });
        if (opt.verbose) std::cerr << "K = " << kk << ", loading kmers from contigs..." << std::endl;
        if (opt.verbose) std::cerr << "K = " << kk << ", saving additional kmers(not found in contigs) from reads..." << std::endl;
        u64 nd = 0;
        // (no read is left on any rank, so this is the contig-derived table alone: the same on every rank, nothing to merge)
        chk(pb_count_considering_contigs(ctx, kk, occ_hist.data(), &max_occ, &nd));
        retain_and_build<W>(kk, cutoff[num_k - 1], false);
        Tick t("contig records (host)");
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
    else throw std::runtime_error("k > 160 is not supported by the device path");
}

} // namespace

int run_assemble_gpu(const GpuAssembleOptions &opt)
{
    if (opt.repeat) throw std::runtime_error("-repeat is outside the scope of the device path (DESIGN.md section 1)");
    if (opt.world > 1 && opt.kmer_occ_only)
        throw std::runtime_error("-kmer_occ_only with -gpus > 1: the slot layout of PREFIX_kmer_occ.bin is defined by the order in which ONE process first sees "
                                 "every k-mer (counter.h:458-476 at -t 1); run it on one GPU");
    const auto t_start = std::chrono::steady_clock::now();
    Driver d(opt);
    auto print_timing = [&]() {
        if (!getenv("PB_TIMING")) return;
        std::cerr << "\n#### STAGE WALL TIMES (s) ####" << std::endl;
        for (const std::string &n : g_clock.order) std::cerr << n << "\t" << g_clock.acc[n] << std::endl;
        std::cerr << "total\t" << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << std::endl;
        std::cerr << "(device context creation, beside the parse\t" << d.create_s << ")" << std::endl;
        auto epoch = [](std::chrono::system_clock::time_point t) { return std::chrono::duration<double>(t.time_since_epoch()).count(); };
        std::cerr.precision(16);
        std::cerr << "(epoch at entry\t" << epoch(std::chrono::system_clock::now()) - std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count()
                  << ")\n(epoch at return\t" << epoch(std::chrono::system_clock::now()) << ")" << std::endl;
    };
    d.min_log_p_join = log(1.0 - pow(10.0, -opt.safety));
    double bubble = atof(opt.bubble.c_str());
    std::get<0>(d.graphs).set_thresholds(bubble, opt.branch);
    std::get<1>(d.graphs).set_thresholds(bubble, opt.branch);
    std::get<2>(d.graphs).set_thresholds(bubble, opt.branch);
    std::get<3>(d.graphs).set_thresholds(bubble, opt.branch);
    std::get<4>(d.graphs).set_thresholds(bubble, opt.branch);

    // the input files are parsed concurrently (one host thread per file, at most 8 at a time) and handed to the device in
    // the order of the command line, which is the order of the reference's temp file 0 at -t 1
    const size_t nf = opt.files.size();
    std::vector<std::thread> reapers;
    struct JoinAll { std::vector<std::thread> &v; ~JoinAll() { for (auto &t : v) if (t.joinable()) t.join(); } } join_reapers{reapers};
    std::vector<std::vector<ParsedPiece>> fpieces(nf);
    std::vector<std::exception_ptr> ferr(nf);
    const unsigned hw = std::max(2u, std::thread::hardware_concurrency());
    if (opt.device_parse) {
        // -device_parse: the bytes of this rank's range of every file go to the device as they are, pieces of at most
        // 1 GiB cut at record marks; newline index, line rules and read offsets are kernels (csrc/raw_parse.inc)
        {
            Tick t("wait for the device context");
            d.ready();
        }
        for (size_t i = 0; i < nf; ++i) {
            Tick t("read files + device-side parse (H2D + device)");
            std::vector<ParsedPiece> none;
            read_file(opt.files[i], none, 1, (unsigned)opt.rank, (unsigned)std::max(opt.world, 1), [&](const char *p, size_t nb, bool fq, bool first, bool last) {
                const int rc = pb_reads_append_raw(d.ctx, (const uint8_t *)p, nb, fq ? 1 : 0, first ? 1 : 0, last ? 1 : 0, nullptr, nullptr);
                if (rc == PB_ERANGE) throw read_too_long();              // common.h:465-467, as the host parser throws it
                d.chk(rc);
            });
        }
    }
    for (size_t base = 0; base < (opt.device_parse ? 0 : nf); base += 8) {
        const size_t hi_f = std::min(nf, base + 8);
        const unsigned per_file = std::max<unsigned>(1, std::min(16u, hw) / (unsigned)(hi_f - base));
        {
            Tick t("read + parse files (host threads)");
            std::vector<std::thread> th;
            for (size_t i = base; i < hi_f; ++i)
                th.emplace_back([&, i]() { try { read_file(opt.files[i], fpieces[i], per_file, (unsigned)opt.rank, (unsigned)std::max(opt.world, 1)); } catch (...) { ferr[i] = std::current_exception(); } });
            for (auto &t2 : th) t2.join();
        }
        {
            Tick t("wait for the device context");
            d.ready();
        }
        {
            Tick t("device allocation for the read stream");
            u64 tb = 0, tr = 0;
            for (size_t i = base; i < hi_f; ++i) for (const ParsedPiece &pc : fpieces[i]) { tb += pc.bases.size(); tr += pc.off.size() - 1; }
            d.chk(pb_reads_reserve(d.ctx, tb, tr));
        }
        for (size_t i = base; i < hi_f; ++i) {
            if (ferr[i]) std::rethrow_exception(ferr[i]);
            Tick t("upload + pack reads (H2D + device)");
            // N ranks: rank r parsed its byte range of every file (read_file cuts at record marks) and takes all of it.  The table that
            // comes out of the merge does not depend on the partition (counts are sums), so any partition gives the reference's files.
            u64 n_file = 0, seen = 0;
            for (const ParsedPiece &pc : fpieces[i]) n_file += pc.off.size() - 1;
            const u64 take_lo = 0, take_hi = n_file;
            for (ParsedPiece &pc : fpieces[i]) {
                const std::vector<u64> &off = pc.off;
                const u64 n_piece = off.size() - 1;
                const u64 p_lo = std::min(n_piece, take_lo > seen ? take_lo - seen : 0), p_hi = std::min(n_piece, take_hi > seen ? take_hi - seen : 0);
                seen += n_piece;
                u64 lo = p_lo;                       // batches of ~256 MB of bases
                while (lo < p_hi) {
                    u64 hi = lo;
                    while (hi < p_hi && off[hi + 1] - off[lo] < (256ull << 20)) ++hi;
                    if (hi == lo) ++hi;
                    d.chk(pb_reads_append(d.ctx, (const uint8_t *)pc.bases.data(), off.data() + lo, hi - lo, 1));
                    lo = hi;
                }
            }
            // Returning ~0.5 GB per file to the kernel costs ~0.15 ms per MB: done by a helper thread.  (HIP calls made meanwhile are
            // slowed down -- upload + first count 0.09 s instead of 0.055 s -- but leaving the buffers to the process exit costs the
            // same time there, serially; measured both ways)
            auto *dead = new std::vector<ParsedPiece>();
            dead->swap(fpieces[i]);
            reapers.emplace_back([dead]() {
                // the pages go back 2 MB at a time (madvise holds the address-space lock shared and briefly); the munmap inside
                // free() then finds nothing left to unmap page by page and no longer blocks the HIP calls of the main thread
                for (ParsedPiece &pc : *dead) {
                    const uintptr_t lo = ((uintptr_t)pc.bases.data() + 4095) & ~(uintptr_t)4095;
                    const uintptr_t hi = ((uintptr_t)pc.bases.data() + pc.bases.capacity()) & ~(uintptr_t)4095;
                    for (uintptr_t a = lo; a < hi; a += (2u << 20)) madvise((void *)a, std::min<uintptr_t>(2u << 20, hi - a), MADV_DONTNEED);
                }
                delete dead;
            });
        }
    }
    d.ready();
    d.k.assign(1, opt.k0);
    dispatch_w(opt.k0, [&](auto w) { d.template initial<decltype(w)::value>(); });
    if (opt.kmer_occ_only) {
        if (opt.verbose) std::cerr << "assemble completed!" << std::endl;
        print_timing();
        return 0;
    }
    for (u64 i = 1; i < d.num_k; ++i) {
        const unsigned kp = d.k[i - 1], kn = d.k[i];
        dispatch_w(kp, [&](auto wp) {
            dispatch_w(kn, [&](auto wn) {
                constexpr int WP = decltype(wp)::value, WN = decltype(wn)::value;
                if constexpr (WN >= WP) d.template save_and_redo<WP, WN>(kn, kp, i);
            });
        });
    }
    const unsigned klast = d.k[d.num_k - 1];
    dispatch_w(klast, [&](auto w) { d.template output_and_after<decltype(w)::value>(klast); });
    const std::string out = opt.out_prefix + "_contig.fa";
    if (d.writer()) {
        Tick t("write contigs");
        print_contigs(out, d.contigs, 1.0, d.avg_len, klast, "seq", false);
        print_contigs(out, d.junctions, 1.0, d.avg_len, klast, "junction", true);
    }
    if (opt.verbose) std::cerr << "assemble completed!" << std::endl;
    print_timing();
    return 0;
}

} // namespace pbh
