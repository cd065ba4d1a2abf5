// libpbgpu.so -- HIP kernels (gfx950 / CDNA4, wave64) and the C-ABI of include/pbgpu.h.
//
// Hot path of `platanus_b assemble` (reference file:line in the comment of each kernel):
//   K1 pack_reads_kernel      ASCII/codes -> 2-bit stream + unknown-base mask      common.h:460-477
//   K2+K3 count_kernel        canonical k-mer extraction + occurrence hashing       counter.h:390-476
//   K4 hist_kernel / compact  occurrenceDistribution + (key,count) stream           counter.h:482-507
//   K5 radix sort             sortedKeyFromKmerFile                                 counter.h:916-951
// All of it is integer / HBM-bound work: no MFMA anywhere (BASELINE.json north_star).
#include "pb_device.hpp"
#include "../../include/pbgpu.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <thread>
#include <mutex>
#include <condition_variable>

// ------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

// A few persistent host threads that copy slices of a buffer (upload_bases: pageable -> pinned staging).  Starting and joining
// threads for every 8 MB chunk cost more than the copies themselves (0.3 ms of 0.5 ms per chunk).
// buffers and host-side bookkeeping of the batched gap-closing graphs (gap_graphs.inc)
struct GapDesc {                     // one gap's share of the batch's arrays
    unsigned long long slot0;        // first slot of its table
    unsigned long long mask;         // capacity - 1
    unsigned long long list0;        // its key list starts here (capacity = table capacity)
    unsigned long long node0;        // its node arrays start here (capacity = its number of keys; set once the keys are counted)
    unsigned long long word0, word_cap;   // its share of the node words
    unsigned long long scr0;         // two byte arrays of (keys + 1) entries: bases collected to the right / to the left
};
struct PbGap {
    DevBuf codes, seq_off, seq_gap, desc, slots, keys, n_keys, n_nodes, n_words, kind, out, cov, length, word_at, words, scr;
    DevBuf in_words, in_word_off, in_bases, in_order, in_cov, in_node_first, slot_first;      // pb_gap_graphs_build_from_nodes
    DevBuf pk_to, pk_bytes, pk_cov, pk_len, pk_woff, pk_words;                               // pb_gap_graphs_fetch: the packed result
    std::vector<GapDesc> h_desc;
    unsigned long long n_gaps = 0, node_cap = 0;
    unsigned w_out = 1;
    std::vector<unsigned> h_n_nodes;
    std::vector<unsigned long long> h_n_words;
};

struct CopyPool {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_go, cv_done;
    u64 gen = 0;
    unsigned pending = 0;
    bool quit = false;
    char *dst = nullptr;
    const char *src = nullptr;
    u64 len = 0;
    void start(unsigned n)
    {
        for (unsigned t = 0; t < n; ++t)
            th.emplace_back([this, t, n]() {
                u64 seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(m);
                    cv_go.wait(lk, [&]() { return quit || gen != seen; });
                    if (quit) return;
                    seen = gen;
                    char *d = dst; const char *s = src; const u64 l = len;
                    lk.unlock();
                    const u64 part = (l + n - 1) / n, lo = std::min(l, (u64)t * part), hi = std::min(l, lo + part);
                    if (lo < hi) memcpy(d + lo, s + lo, hi - lo);
                    lk.lock();
                    if (--pending == 0) cv_done.notify_one();
                }
            });
    }
    void copy(void *d, const void *s, u64 l)           // returns when every slice has been copied
    {
        std::unique_lock<std::mutex> lk(m);
        dst = (char *)d; src = (const char *)s; len = l;
        pending = (unsigned)th.size();
        ++gen;
        cv_go.notify_all();
        cv_done.wait(lk, [&]() { return pending == 0; });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(m); quit = true; }
        cv_go.notify_all();
        for (auto &t : th) t.join();
    }
};

struct pb_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t lazy_ev[8] = {nullptr};          // LazyTimer: start / stop events of up to 4 stages that are read out later
    std::string err;
    std::map<std::string, float> stage_ms;

    // packed reads (K1 output)
    u64 n_reads = 0, n_bases = 0;
    u64 stream_len = 0;                 // positions incl. one separator per read
    DevBuf packed, nmask;               // u64 words
    std::vector<u64> len_hist;          // host side, grows with the longest read seen

    // occurrence table of the last pb_count
    unsigned k = 0;
    int W = 0;
    DevBuf table;
    u64 table_cap = 0;
    u64 n_instances = 0, n_distinct = 0;
    u64 n_novel = 0;                    // pb_count_considering_contigs: leading records of the distinct arrays that came from reads
    DevBuf d_hist;                      // u64[65536]
    DevBuf d_scalars;                   // u64[16] scratch counters

    // retained (sorted) set
    DevBuf ret_keys, ret_counts;        // u64[n*W], u16[n]
    u64 n_retained = 0;

    // distinct (key,count) arrays of the last pb_count / pb_merge_end (device form of kmerFP)
    DevBuf dist_keys, dist_counts;      // u64[n*W], u32[n]
    int count_path = 0;                 // 1 = direct (v1), 2 = partitioned (v2)
    DevBuf part_bucket1, part_bucket2, part_meta, part_spill;
    DevBuf part_pos1, part_pos2, sk_first;   // position-carrying pass (pb_kmer_occ_layout): record positions, first occurrence per distinct k-mer
    void *bounce[2] = {nullptr, nullptr};   // pinned staging buffers of upload_bases
    hipEvent_t bounce_ev[2];
    bool bounce_used[2] = {false, false};
    std::vector<u64> rel_keep[2];
    hipEvent_t rel_ev[2] = {nullptr, nullptr};
    int rel_next = 0;
    int bounce_next = 0;
    CopyPool *copy_pool = nullptr;
    u64 *h_hist = nullptr;              // pinned; filled behind sk_bucket_kernel so that the pass needs one host synchronisation
    bool h_hist_valid = false;
    // the k <= 32 counting pass as a captured graph (memsets, sk_part1, sk_part2, sk_bucket, scalars + histogram copy-out): replayed while
    // nothing the launches depend on has changed (sk_graph_key: every pointer, size and launch dimension that went into the capture)
    hipGraphExec_t sk_graph = nullptr;
    std::vector<u64> sk_graph_key;
    u64 *h_scalars = nullptr;           // pinned u64[16]: the pass scalars of a replay
    u64 h_hist_hi = 65536;              // bins of h_hist at or beyond this are zero
    int opt_graph = 1;                  // pb_set_option("graph"): 0 = every pass launched kernel by kernel (with per-kernel timers)
    int opt_hist_zeroed = 0;            // pb_set_option("hist_zeroed"): occ_hist arrays handed to the counting calls are all zero on entry
    int opt_bucket_blocks = 0;          // pb_set_option("sk_bucket_blocks"): grid of the bucket kernel (0 = 8192, each block striding over its buckets)
    int opt_bucket_form = 2;            // pb_set_option("sk_bucket_form"): 1 = sk_bucket_kernel (chunks of 768 records, per-wave piece lists), 2 = sk_bucket2_kernel (whole bucket)
    u64 sk_owner_cap1 = 0;          // pb_sk_owner_records: records per owner region
    u64 exp_cur[64] = {0};           // start of every owner's segment in the export (pb_export_begin -> pb_export_copy)
    std::vector<u32> sk_owner_cur;   // pb_sk_owner_records: records per (owner, XCD) region
    u64 exp2_key_start[128] = {0}, exp2_cnt_start[64] = {0};   // pb_export_split_begin -> pb_export_split_copy
    unsigned exp2_parts = 0;
    u64 sk_records = 0;            // super-k-mer records of the last counting pass (0 = other path)
    u64 sk_flagged = 0, sk_spilled = 0, sk_extra_for = 0;
    u64 sk_fcap = 0, sk_fcap_for = 0;   // capacity of the fallback table that an earlier pass over the same stream needed
    u64 sk_room6_for = 0;               // ... and the stream (by its window count) whose buckets needed 6 x the mean as capacity
    u64 sk_out_need = 0, sk_out_for = 0;   // ... and the distinct keys of a stream whose output arrays, sized by estimate, were too small
    int opt_out_div = 0;                // "sk_out_div" (tests): the output arrays of the k <= 32 pass are sized for windows / value keys whatever the input size
    u64 hist_done = 0;              // leading records of the distinct arrays already tallied into d_hist
    unsigned sk_bits = 0, sk_extra_bits = 0;       // bucket bits used; extra bits that an earlier pass over the same stream needed
    // pb_set_option: choices between equivalent implementations (tests and A/B runs; no option changes a result)
    int opt_count_path = 0;         // 0 = by input size, 1 = direct (global table), 2 = super-k-mer pass
    int sk_force_bits = 0;          // bucket bits of the super-k-mer pass (0 = by input size)
    int sk_force_l1 = 0;            // ... of which level 1 (0 = half)
    int opt_merge_table = 0, opt_merge_16b = 0, opt_chain_rank_full = 0, opt_grid_cap = 0, opt_verbose = 0;
    int opt_dist_design = 0;        // pb_dist_count: 0 = records before counting where it applies, 1 = always count + exchange (key,count) + merge

    // owner exchange (X1)
    DevBuf exp_keys, exp_counts, merge_table;
    u64 exp_total = 0, merge_cap = 0, merge_records = 0;
    unsigned exp_parts = 0;

    // graph stage / multi-k rounds
    DevBuf index_table;                 // retained key -> index + 1
    u64 index_table_cap = 0;
    DevBuf node_mark, node_idx, node_rec; // compact export of the k-mers the host replay touches
    DevBuf node_work, node_table;         // node-parallel cleaning decisions (node_kernels.inc)
    u64 n_nodes = 0, node_cap_hint = 0;
    DevBuf nmask_cur, alive, d_read_start;   // filtered read set: working mask copy, per-read flag, stream starts
    std::vector<u64> read_start;        // host copy: stream position of every read + final sentinel (= stream_len)
    size_t read_start_dev_n = 0;
    bool mask_filtered = false;
    DevBuf contig_packed, contig_mask, ctable;   // contig-derived next-k table (saveContig)
    u64 ctable_cap = 0;
    unsigned ctable_k = 0;
    DevBuf chain_work, chain_rec, chain_seq;     // K7 chain compaction
    DevBuf layout_work, layout_slots, layout_out, scan_sums, slot_blk;   // CT8 exact PREFIX_kmer_occ.bin layout
    PbGap gap;                                                           // pb_gap_graphs_build / _fetch
    DevBuf raw_bytes, raw_work, raw_tiles, raw_s3, raw_hist;              // pb_reads_append_raw: file bytes and the per-line arrays of the device-side parse
    u64 layout_records = 0;
    u64 n_chains = 0, chain_seq_words = 0;
    void *chain_ptrs[2] = {nullptr, nullptr};

    // N ranks on one run (dist.inc)
    struct PbDist *dist = nullptr;

    // scratch
    DevBuf scratch[6];
};

#define PB_STREAM_PAD_WORDS64 8      /* 512 positions of padding: a 160-mer window never leaves the buffer */

static std::string g_create_error;

#define HIPCHK(call)                                                                              \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
            return PB_EHIP;                                                                       \
        }                                                                                         \
    } while (0)

static int ensure(pb_ctx *ctx, DevBuf &b, size_t bytes, bool keep = false)
{
    if (b.bytes >= bytes) return PB_OK;
    size_t nb = bytes;
    if (keep) nb = std::max(bytes, b.bytes * 2);
    void *np = nullptr;
    hipError_t e = hipMalloc(&np, nb);
    if (e != hipSuccess) {
        ctx->err = "hipMalloc(" + std::to_string(nb) + "): " + hipGetErrorString(e);
        return PB_ENOMEM;
    }
    if (keep && b.p && b.bytes) {
        e = hipMemcpyAsync(np, b.p, b.bytes, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->err = std::string("grow copy: ") + hipGetErrorString(e); (void)hipFree(np); return PB_EHIP; }
    }
    if (b.p) (void)hipFree(b.p);
    b.p = np;
    b.bytes = nb;
    return PB_OK;
}
static void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}

struct StageTimer {
    pb_ctx *ctx;
    const char *name;
    StageTimer(pb_ctx *c, const char *n) : ctx(c), name(n) { (void)hipEventRecord(ctx->ev0, ctx->stream); }
    int stop()
    {
        (void)hipEventRecord(ctx->ev1, ctx->stream);
        hipError_t e = hipEventSynchronize(ctx->ev1);
        if (e != hipSuccess) { ctx->err = std::string(name) + ": " + hipGetErrorString(e); return PB_EHIP; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
        ctx->stage_ms[name] = ms;
        return PB_OK;
    }
};

// Stage time without a host synchronisation per stage: the events are recorded on the stream and read after a later
// synchronisation (LazyTimer::read).  slot 0..3.
struct LazyTimer {
    pb_ctx *ctx;
    const char *name;
    int slot;
    bool live;
    LazyTimer(pb_ctx *c, const char *n, int s, bool record = true) : ctx(c), name(n), slot(s), live(record)
    {
        for (int i = 2 * s; i < 2 * s + 2; ++i) if (!ctx->lazy_ev[i]) (void)hipEventCreate(&ctx->lazy_ev[i]);
        if (live) (void)hipEventRecord(ctx->lazy_ev[2 * s], ctx->stream);
    }
    void stop() { if (live) (void)hipEventRecord(ctx->lazy_ev[2 * slot + 1], ctx->stream); }
    void read()                                  // after the stream has been synchronised past stop()
    {
        float ms = 0;
        (void)hipEventElapsedTime(&ms, ctx->lazy_ev[2 * slot], ctx->lazy_ev[2 * slot + 1]);
        ctx->stage_ms[name] = ms;
    }
};

static inline unsigned grid_for(u64 n, unsigned block, unsigned max_blocks = 256 * 16)
{
    u64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (unsigned)g;
}

#define SK_HIST_HEAD 1024            /* histogram bins a replayed pass copies out with its scalars */
#include "kernels_count.inc"
#include "kernels_superk.inc"
#include "kernels_superk2.inc"
#include "kernels_superk3.inc"
#include "kernels_sort.inc"

// ------------------------------------------------------------------------------------------------------
// host side of the C-ABI
// ------------------------------------------------------------------------------------------------------
extern "C" {

int pb_create(int device_id, pb_ctx **out)
{
    if (!out) return PB_EINVAL;
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n == 0) {
        g_create_error = std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0");
        return PB_ENODEVICE;
    }
    if (device_id < 0 || device_id >= n) { g_create_error = "device id out of range"; return PB_EINVAL; }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return PB_EHIP; }
    pb_ctx *ctx = new pb_ctx();
    ctx->dev = device_id;
    if (hipStreamCreate(&ctx->stream) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess ||
        hipEventCreate(&ctx->ev1) != hipSuccess) {
        g_create_error = "stream/event creation failed";
        delete ctx;
        return PB_EHIP;
    }
    *out = ctx;
    return PB_OK;
}

extern "C" int pb_dist_detach(pb_ctx *ctx);
void pb_destroy(pb_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->dev);
    (void)hipStreamSynchronize(ctx->stream);
    (void)pb_dist_detach(ctx);
    release(ctx->packed); release(ctx->nmask); release(ctx->table); release(ctx->d_hist); release(ctx->d_scalars);
    release(ctx->ret_keys); release(ctx->ret_counts); release(ctx->exp_keys); release(ctx->exp_counts); release(ctx->merge_table);
    release(ctx->dist_keys); release(ctx->dist_counts); release(ctx->part_pos1); release(ctx->part_pos2); release(ctx->sk_first); release(ctx->part_bucket1); release(ctx->part_bucket2); release(ctx->part_meta); release(ctx->part_spill);
    release(ctx->node_work); release(ctx->node_table); release(ctx->index_table); release(ctx->node_mark); release(ctx->node_idx); release(ctx->node_rec); release(ctx->nmask_cur); release(ctx->alive); release(ctx->d_read_start);
    release(ctx->contig_packed); release(ctx->contig_mask); release(ctx->ctable);
    release(ctx->chain_work); release(ctx->chain_rec); release(ctx->chain_seq);
    for (DevBuf *b : {&ctx->gap.codes, &ctx->gap.seq_off, &ctx->gap.seq_gap, &ctx->gap.desc, &ctx->gap.slots, &ctx->gap.keys, &ctx->gap.n_keys, &ctx->gap.n_nodes, &ctx->gap.n_words,
                      &ctx->gap.kind, &ctx->gap.out, &ctx->gap.cov, &ctx->gap.length, &ctx->gap.word_at, &ctx->gap.words, &ctx->gap.scr, &ctx->gap.in_words, &ctx->gap.in_word_off,
                      &ctx->gap.in_bases, &ctx->gap.in_order, &ctx->gap.in_cov, &ctx->gap.in_node_first, &ctx->gap.slot_first, &ctx->gap.pk_to, &ctx->gap.pk_bytes, &ctx->gap.pk_cov,
                      &ctx->gap.pk_len, &ctx->gap.pk_woff, &ctx->gap.pk_words}) release(*b);
    release(ctx->raw_bytes); release(ctx->raw_work); release(ctx->raw_tiles); release(ctx->raw_s3); release(ctx->raw_hist);
    release(ctx->layout_work); release(ctx->layout_slots); release(ctx->layout_out); release(ctx->scan_sums); release(ctx->slot_blk);
    for (auto &s : ctx->scratch) release(s);
    delete ctx->copy_pool;
    if (ctx->sk_graph) (void)hipGraphExecDestroy(ctx->sk_graph);
    if (ctx->h_hist) (void)hipHostFree(ctx->h_hist);
    if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
    for (int b = 0; b < 2; ++b) if (ctx->bounce[b]) { (void)hipHostFree(ctx->bounce[b]); (void)hipEventDestroy(ctx->bounce_ev[b]); }
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    for (auto &e : ctx->lazy_ev) if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->rel_ev) if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *pb_last_error(const pb_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

// which field of the context an option sets; nullptr: unknown name or value out of range
static int *option_slot(pb_ctx *ctx, const std::string &n, long value)
{
    if (n == "count_path" && value >= 0 && value <= 2) return &ctx->opt_count_path;
    if (n == "sk_bucket_bits" && (value == 0 || (value >= 2 && value <= SK_MAX_B))) return &ctx->sk_force_bits;
    if (n == "sk_l1_bits" && value >= 0 && value <= 9) return &ctx->sk_force_l1;
    if (n == "merge_table" && (value == 0 || value == 1)) return &ctx->opt_merge_table;
    if (n == "merge_16b" && (value == 0 || value == 1)) return &ctx->opt_merge_16b;
    if (n == "chain_rank_full" && (value == 0 || value == 1)) return &ctx->opt_chain_rank_full;
    if (n == "verbose" && (value == 0 || value == 1)) return &ctx->opt_verbose;
    if (n == "dist_design" && (value == 0 || value == 1)) return &ctx->opt_dist_design;
    if (n == "grid_cap" && value >= 0 && value <= (1 << 20)) return &ctx->opt_grid_cap;
    if (n == "graph" && (value == 0 || value == 1)) return &ctx->opt_graph;
    if (n == "sk_bucket_form" && (value == 1 || value == 2)) return &ctx->opt_bucket_form;
    if (n == "sk_bucket_blocks" && value >= 0 && value <= 65536) return &ctx->opt_bucket_blocks;
    if (n == "hist_zeroed" && (value == 0 || value == 1)) return &ctx->opt_hist_zeroed;
    if (n == "sk_out_div" && value >= 0 && value <= (1 << 20)) return &ctx->opt_out_div;
    return nullptr;
}

int pb_check_option(const char *name, long value)
{
    if (!name) return PB_EINVAL;
    pb_ctx probe;                               // (plain fields only: no device is touched)
    return option_slot(&probe, name, value) ? PB_OK : PB_EINVAL;
}

int pb_set_option(pb_ctx *ctx, const char *name, long value)
{
    if (!ctx || !name) return PB_EINVAL;
    int *slot = option_slot(ctx, name, value);
    if (!slot) { ctx->err = std::string("pb_set_option: unknown option or value out of range: ") + name; return PB_EINVAL; }
    *slot = (int)value;
    return PB_OK;
}

int pb_device_info(const pb_ctx *ctx, char *name, size_t name_cap, int *n_cu, uint64_t *hbm_bytes)
{
    if (!ctx) return PB_EINVAL;
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, ctx->dev) != hipSuccess) return PB_EHIP;
    if (name && name_cap) { snprintf(name, name_cap, "%s (%s)", p.name, p.gcnArchName); }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return PB_OK;
}

int pb_reads_clear(pb_ctx *ctx)
{
    if (!ctx) return PB_EINVAL;
    ctx->n_reads = ctx->n_bases = ctx->stream_len = 0;
    ctx->len_hist.clear();
    ctx->read_start.clear();
    ctx->read_start_dev_n = 0;
    ctx->mask_filtered = false;
    return PB_OK;
}

int pb_reads_info(const pb_ctx *ctx, uint64_t *n_reads, uint64_t *n_bases)
{
    if (!ctx) return PB_EINVAL;
    if (n_reads) *n_reads = ctx->n_reads;
    if (n_bases) *n_bases = ctx->n_bases;
    return PB_OK;
}

int pb_reads_length_hist(pb_ctx *ctx, uint64_t *hist, uint64_t hist_n)
{
    if (!ctx || !hist) return PB_EINVAL;
    for (u64 i = 0; i < hist_n; ++i) hist[i] = i < ctx->len_hist.size() ? ctx->len_hist[i] : 0;
    return PB_OK;
}


// Host -> device copy of a large pageable buffer through two pinned bounce buffers: a pool of host threads fills one while the DMA
// engine drains the other (a plain hipMemcpy from pageable memory pins the pages on the fly: ~5 GB/s measured on first use of a
// buffer, against 55 GB/s from pinned memory).
static int upload_bases(pb_ctx *ctx, unsigned char *dst, const uint8_t *src, u64 n)
{
    const u64 CH = 16ull << 20;
    if (n < (4ull << 20)) {
        HIPCHK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, ctx->stream));
        return PB_OK;
    }
    if (!ctx->copy_pool) {                     // (keyed on the last thing set up: a failure half-way leaves nothing that looks ready)
        void *bb[2] = {nullptr, nullptr};
        hipEvent_t be[2] = {nullptr, nullptr};
        for (int b = 0; b < 2; ++b) {
            hipError_t e = hipHostMalloc(&bb[b], CH, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&be[b], hipEventDisableTiming);
            if (e != hipSuccess) {
                for (int q = 0; q < 2; ++q) { if (bb[q]) (void)hipHostFree(bb[q]); if (be[q]) (void)hipEventDestroy(be[q]); }
                ctx->err = std::string("pinned staging buffers: ") + hipGetErrorString(e);
                return PB_ENOMEM;
            }
        }
        for (int b = 0; b < 2; ++b) { ctx->bounce[b] = bb[b]; ctx->bounce_ev[b] = be[b]; }
        ctx->copy_pool = new CopyPool();
        ctx->copy_pool->start(std::min(8u, std::max(1u, std::thread::hardware_concurrency())));
    }
    for (u64 o = 0; o < n; o += CH) {
        const int b = ctx->bounce_next;
        ctx->bounce_next ^= 1;
        const u64 len = std::min(CH, n - o);
        if (ctx->bounce_used[b]) HIPCHK(hipEventSynchronize(ctx->bounce_ev[b]));      // the copy that last used this buffer has left it
        ctx->bounce_used[b] = true;
        ctx->copy_pool->copy(ctx->bounce[b], src + o, len);
        HIPCHK(hipMemcpyAsync(dst + o, ctx->bounce[b], len, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipEventRecord(ctx->bounce_ev[b], ctx->stream));
    }
    return PB_OK;
}

int pb_reads_reserve(pb_ctx *ctx, uint64_t n_bases, uint64_t n_reads)
{
    if (!ctx) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    const u64 new_len = ctx->stream_len + n_bases + n_reads;
    const u64 words64 = (new_len + 63) / 64 + PB_STREAM_PAD_WORDS64;
    int rc;
    if ((rc = ensure(ctx, ctx->packed, words64 * 16, true)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->nmask, words64 * 8, true)) != PB_OK) return rc;
    ctx->read_start.reserve(ctx->read_start.size() + n_reads + 1);
    return PB_OK;
}

int pb_reads_append(pb_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int ascii)
{
    if (!ctx || !offsets || (n_reads && !bases && offsets[n_reads] > 0)) return PB_EINVAL;
    if (n_reads == 0) return PB_OK;
    (void)hipSetDevice(ctx->dev);
    const u64 n_new_bases = offsets[n_reads] - offsets[0];
    static const bool trace = getenv("PB_TRACE_APPEND") != nullptr;
    auto now = []() { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    auto lap = [&](const char *what) { if (trace) { auto t1 = now(); fprintf(stderr, "  append %-10s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; } };
    // every length is checked before the context changes: a refused batch leaves no trace (lengthDistribution stays what it was)
    for (u64 r = 0; r < n_reads; ++r)
        if (offsets[r + 1] - offsets[r] >= 500000) { ctx->err = "read of 500000 bases or more (common.cpp:28)"; return PB_ERANGE; }
    for (u64 r = 0; r < n_reads; ++r) {
        const u64 len = offsets[r + 1] - offsets[r];
        if (len >= ctx->len_hist.size()) ctx->len_hist.resize(len + 1, 0);
        ++ctx->len_hist[len];
    }
    const u64 old_len = ctx->stream_len;
    const u64 new_len = old_len + n_new_bases + n_reads;
    const u64 words64 = (new_len + 63) / 64 + PB_STREAM_PAD_WORDS64;
    int rc;
    const bool fresh = ctx->packed.bytes < words64 * 16;
    if ((rc = ensure(ctx, ctx->packed, words64 * 16, true)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->nmask, words64 * 8, true)) != PB_OK) return rc;
    (void)fresh;
    lap("hist+ens");
    // staging copy of the host buffers
    DevBuf &d_bases = ctx->scratch[0], &d_off = ctx->scratch[1];
    if ((rc = ensure(ctx, d_bases, n_new_bases + 16)) != PB_OK) return rc;
    if ((rc = ensure(ctx, d_off, (n_reads + 1) * 8)) != PB_OK) return rc;
    // offsets relative to this batch; the vector stays alive in the context until two calls later (the copy below may read it
    // asynchronously: an event recorded behind the copy is waited for before the vector is written again)
    const int rslot = ctx->rel_next;
    ctx->rel_next ^= 1;
    if (!ctx->rel_ev[rslot]) HIPCHK(hipEventCreateWithFlags(&ctx->rel_ev[rslot], hipEventDisableTiming));
    else HIPCHK(hipEventSynchronize(ctx->rel_ev[rslot]));        // the copy that last read this vector is done
    std::vector<u64> &rel = ctx->rel_keep[rslot];
    rel.resize(n_reads + 1);
    for (u64 r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - offsets[0];
    lap("rel");
    HIPCHK(hipMemcpyAsync(d_off.p, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipEventRecord(ctx->rel_ev[rslot], ctx->stream));
    if ((rc = upload_bases(ctx, (unsigned char *)d_bases.p, bases + offsets[0], n_new_bases)) != PB_OK) return rc;
    if (n_new_bases < (4ull << 20)) HIPCHK(hipStreamSynchronize(ctx->stream));
    lap("upload");
    const u64 first64 = old_len / 64;
    const u64 n64 = words64 - first64;          // includes the all-unknown padding words
    pack_reads_kernel<<<(unsigned)((n64 + 255) / 256), 256, 0, ctx->stream>>>(
        (const unsigned char *)d_bases.p, (const u64 *)d_off.p, n_reads, old_len, new_len, (u64 *)ctx->packed.p,
        (u64 *)ctx->nmask.p, first64, n64, ascii);
    HIPCHK(hipGetLastError());
    // no synchronisation here: the next call's host-side staging overlaps this call's DMA and pack kernel (both are ordered on
    // the context's stream; every consumer of the stream runs on the same stream)
    lap("pack");
    if (!ctx->read_start.empty()) ctx->read_start.pop_back();
    for (u64 r = 0; r < n_reads; ++r) ctx->read_start.push_back(old_len + rel[r] + r);
    ctx->read_start.push_back(new_len);
    ctx->read_start_dev_n = 0;
    ctx->mask_filtered = false;
    ctx->stream_len = new_len;
    ctx->n_reads += n_reads;
    ctx->n_bases += n_new_bases;
    lap("starts");
    return PB_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------------
// counting: host orchestration
// ------------------------------------------------------------------------------------------------------
static const u64 *cur_mask(pb_ctx *ctx);
static int finish_count(pb_ctx *ctx, uint64_t *occ_hist, uint64_t *max_occ)
{
    // histogram of the distinct arrays (K4a) + copy out
    int rc;
    if ((rc = ensure(ctx, ctx->d_hist, 65536 * 8)) != PB_OK) return rc;
    // records [0, hist_done) were tallied into d_hist by the kernel that wrote them (sk_bucket_kernel's emit); the rest here
    const u64 done = ctx->hist_done <= ctx->n_distinct ? ctx->hist_done : 0;
    ctx->hist_done = 0;
    if (done == 0) HIPCHK(hipMemsetAsync(ctx->d_hist.p, 0, 65536 * 8, ctx->stream));
    ctx->stage_ms["hist"] = 0.f;
    if (ctx->n_distinct > done) {
        StageTimer tm(ctx, "hist");
        hist_array_kernel<<<grid_for(ctx->n_distinct - done, 256, 256 * 8), 256, 0, ctx->stream>>>((const u32 *)ctx->dist_counts.p + done, ctx->n_distinct - done,
                                                                                                  (u64 *)ctx->d_hist.p);
        HIPCHK(hipGetLastError());
        if ((rc = tm.stop()) != PB_OK) return rc;
    }
    std::vector<u64> hv;
    const u64 *h = ctx->h_hist;
    if (!(ctx->h_hist_valid && done == ctx->n_distinct && done > 0)) {      // (the super-k-mer pass copied it out with its scalars)
        hv.resize(65536);
        HIPCHK(hipMemcpyAsync(hv.data(), ctx->d_hist.p, 65536 * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        h = hv.data();
    }
    // bins at or beyond `hi` are zero (a replayed pass says how far its counts reach): nothing to look at there, and -- option
    // "hist_zeroed": the caller's array is zero already -- nothing to write (512 KB of fresh pages per call otherwise)
    const u64 hi = h == ctx->h_hist && ctx->h_hist_valid ? std::min<u64>(ctx->h_hist_hi, PB_OCC_BINS) : PB_OCC_BINS;
    ctx->h_hist_valid = false;
    static_assert(sizeof(uint64_t) == sizeof(u64), "histogram bins are 64-bit");
    if (occ_hist) {
        memcpy(occ_hist, h, (size_t)hi * 8);
        if (hi < PB_OCC_BINS && !ctx->opt_hist_zeroed) memset(occ_hist + hi, 0, (size_t)(PB_OCC_BINS - hi) * 8);
    }
    if (max_occ) { *max_occ = 0; for (int i = (int)std::min<u64>(PB_COUNT_MAX, hi - 1); i > 0; --i) if (h[i]) { *max_occ = (u64)i; break; } }
    return PB_OK;
}

// table (ctx->table) -> distinct arrays; sets n_distinct
template <int W>
static int table_to_arrays(pb_ctx *ctx, u64 upper_bound, bool append)
{
    int rc;
    const u64 base = append ? ctx->n_distinct : 0;
    const u64 cap_out = std::max<u64>(base + upper_bound, 1);
    if ((rc = ensure(ctx, ctx->dist_keys, cap_out * W * 8, append)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->dist_counts, cap_out * 4, append)) != PB_OK) return rc;
    TableView<W> tv{(u64 *)ctx->table.p, ctx->table_cap - 1};
    HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
    StageTimer tm(ctx, "compact");
    table_to_arrays_kernel<W><<<grid_for(ctx->table_cap, 256 * T2A_SPT, 256 * 8), 256, 0, ctx->stream>>>(
        tv, (u64 *)ctx->dist_keys.p + base * W, (u32 *)ctx->dist_counts.p + base, cap_out - base, (u64 *)ctx->d_scalars.p);
    HIPCHK(hipGetLastError());
    if ((rc = tm.stop()) != PB_OK) return rc;
    u64 n = 0;
    HIPCHK(hipMemcpyAsync(&n, ctx->d_scalars.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (n > cap_out - base) { ctx->err = "distinct array overflow"; return PB_ENOMEM; }
    ctx->n_distinct = base + n;
    return PB_OK;
}

// v1: one lane per window, global-atomic hashing (any W)
template <int W>
static int count_direct(pb_ctx *ctx, unsigned k)
{
    int rc;
    u64 windows = ctx->stream_len > (u64)k * ctx->n_reads ? ctx->stream_len - (u64)(k - 1) * ctx->n_reads : 1;
    u64 cap = 1ull << 16;
    while (cap < windows / 2) cap <<= 1;
    float ms_count = 0;
    for (int attempt = 0; attempt < 8; ++attempt) {
        if ((rc = ensure(ctx, ctx->table, cap * PB_SLOT_WORDS(W) * 8)) != PB_OK) return rc;
        ctx->table_cap = cap;
        TableView<W> tv{(u64 *)ctx->table.p, cap - 1};
        HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
        {
            StageTimer ti(ctx, "table_init");
            table_init_kernel<W><<<grid_for(cap * PB_SLOT_WORDS(W), 256, 256 * 32), 256, 0, ctx->stream>>>(tv.slots, cap);
            if ((rc = ti.stop()) != PB_OK) return rc;
        }
        StageTimer tm(ctx, "count");
        if (ctx->stream_len >= k) {
            count_kernel<W><<<grid_for(ctx->stream_len, 256, ctx->opt_grid_cap ? (unsigned)ctx->opt_grid_cap : 256 * 32), 256, 0, ctx->stream>>>(
                (const u64 *)ctx->packed.p, cur_mask(ctx), ctx->stream_len, k, tv, 256u, (u64 *)ctx->d_scalars.p);
        }
        HIPCHK(hipGetLastError());
        if ((rc = tm.stop()) != PB_OK) return rc;
        ms_count += ctx->stage_ms["count"];
        u64 sc[2];
        HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 16, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(hipStreamSynchronize(ctx->stream));
        if (sc[1] == 0) { ctx->n_instances = sc[0]; break; }
        cap <<= 1;
        if (attempt == 7) { ctx->err = "occurrence table overflow after 8 doublings"; return PB_ENOMEM; }
    }
    ctx->stage_ms["count"] = ms_count;
    ctx->stage_ms["part1"] = ctx->stage_ms["part2"] = ctx->stage_ms["bucket"] = ctx->stage_ms["seg_compact"] = 0.f;
    return table_to_arrays<W>(ctx, std::min<u64>(ctx->n_instances, ctx->table_cap), false);
}

// v2: super-k-mer records, two-level partition by minimizer bucket, LDS-resident counting (k <= 32)
static int count_superk(pb_ctx *ctx, unsigned k, bool *done, bool want_pos = false)
{
    int rc;
    *done = false;
    // upper bound on counted windows: every read loses k - 1 positions and its separator
    const u64 lost = (u64)k * ctx->n_reads;
    const u64 windows = ctx->stream_len > lost + 4096 ? ctx->stream_len - lost : ctx->stream_len;
    SkParams p;
    p.k = k;
    p.m = k < SK_M ? k : SK_M;
    p.w = k - p.m + 1;
    // ~SK_SLOTS * 3 windows per final bucket: with sequencing-depth data that is SK_SLOTS / 2 or fewer distinct keys for the
    // SK_SLOTS-slot LDS table.  Low-coverage data has more distinct keys per window; the pass is then repeated with 4x the
    // buckets (see below) and the context remembers what worked.
    unsigned B = 8;
    while (((u64)(SK_SLOTS * 3) << B) < windows && B < SK_MAX_B) ++B;
    if (ctx->sk_extra_bits && ctx->sk_extra_for == windows) B = std::min((unsigned)SK_MAX_B, B + ctx->sk_extra_bits);
    bool forced_b = false;
    if (ctx->sk_force_bits >= 2 && ctx->sk_force_bits <= SK_MAX_B) { B = (unsigned)ctx->sk_force_bits; forced_b = true; }
    u64 fcap = ctx->sk_fcap && ctx->sk_fcap_for == windows ? ctx->sk_fcap : 1ull << 20;
    double room2 = ctx->sk_room6_for && ctx->sk_room6_for == windows ? 6.0 : 3.0;      // final-bucket capacity over the mean
    for (int attempt = 0; attempt < 12; ++attempt) {
        const auto t_attempt = std::chrono::steady_clock::now();
        p.B = B;
        p.l1 = std::max(B / 2, B > 9 ? B - 9 : 0u);                  // (a level has at most 2^9 digits in sk_part2; sk_part1 takes up to 2^11)
        if (ctx->sk_force_l1 >= 1 && ctx->sk_force_l1 < (int)B && (int)B - ctx->sk_force_l1 <= 9) p.l1 = (unsigned)ctx->sk_force_l1;
        p.l2 = B - p.l1;
        // expected windows per record: a run ends when a smaller m-mer enters the window or the minimizer leaves it -- (w + 1) / 2
        // windows on random sequence; every read start and every unknown base cuts one more
        const double est_rec = (double)windows * 2.0 / (p.w + 1.0) + 2.0 * (double)ctx->n_reads;
        p.nx = PB_XCD;
        const u64 nb1 = (1ull << p.l1) * PB_XCD, nb2 = 1ull << B;
        p.cap1 = (u64)(est_rec / (double)nb1 * 1.5) + 4096;
        p.cap2 = (u32)std::min<double>(est_rec / (double)nb2 * room2 + 256.0, 1e9);   // minimizer sites are few per bucket: sizes vary a lot
        if ((rc = ensure(ctx, ctx->part_bucket1, nb1 * p.cap1 * SkStore16::BYTES)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->part_bucket2, (nb2 * (u64)p.cap2 + 1024) * SkStore16::BYTES)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->part_meta, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4)) != PB_OK) return rc;
        p.spill_cap = std::max<u64>((u64)(est_rec / 16), 1ull << 16);
        if ((rc = ensure(ctx, ctx->part_spill, p.spill_cap * SkStore16::BYTES)) != PB_OK) return rc;
        p.pos1 = p.pos2 = nullptr;
        if (want_pos) {
            if ((rc = ensure(ctx, ctx->part_pos1, nb1 * p.cap1 * 4)) != PB_OK) return rc;
            if ((rc = ensure(ctx, ctx->part_pos2, (nb2 * (u64)p.cap2 + 1024) * 4)) != PB_OK) return rc;
            p.pos1 = (u32 *)ctx->part_pos1.p;
            p.pos2 = (u32 *)ctx->part_pos2.p;
        }
        p.spill = ctx->part_spill.p;
        p.bucket1 = ctx->part_bucket1.p;
        p.bucket2 = ctx->part_bucket2.p;
        p.cursor1 = (u32 *)ctx->part_meta.p;
        p.cursor2 = p.cursor1 + nb1 * SK_CUR1_STRIDE;
        p.flag2 = p.cursor2 + nb2;
        // the distinct arrays are written in place: room for every bucket's table full, plus the fallback table's records.  For a large input
        // that bound is tens of GB that are never written (100 Mbp x 100x: 48 GB for 1.2 G keys of 7.9 G windows) and device memory costs
        // ~30 ms per GB to get: then the arrays are sized for windows / 5 keys (sequencing depth gives ~ windows / 6.5); a pass that finds
        // more says how many (scalars[4] counts on) and is repeated once with exactly that.
        const u64 out_bound = std::min<u64>(std::max<u64>(ctx->stream_len, 1), nb2 * (u64)SK_SLOT_LIMIT);
        u64 out_cap = out_bound;
        {
            const u64 div = ctx->opt_out_div ? (u64)ctx->opt_out_div : (out_bound * 12 > (8ull << 30) ? 5 : 0);
            if (ctx->sk_out_need && ctx->sk_out_for == windows) out_cap = std::min(out_bound, ctx->sk_out_need);
            else if (div) out_cap = std::min(out_bound, std::max<u64>(windows / div, 1ull << 12));
        }
        if ((rc = ensure(ctx, ctx->dist_keys, (out_cap + fcap) * 8)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->dist_counts, (out_cap + fcap) * 4)) != PB_OK) return rc;
        if (want_pos && (rc = ensure(ctx, ctx->sk_first, (out_cap + fcap) * 4)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->table, fcap * 16)) != PB_OK) return rc;
        ctx->table_cap = fcap;
        p.fallback = TableView<1>{(u64 *)ctx->table.p, fcap - 1};
        p.scalars = (u64 *)ctx->d_scalars.p;
        if ((rc = ensure(ctx, ctx->d_hist, 65536 * 8)) != PB_OK) return rc;
        p.hist = (u64 *)ctx->d_hist.p;
        const double alloc_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_attempt).count();
        ctx->stage_ms["table_init"] = 0.f;                         // (the fallback table is initialised only if a flagged bucket needs it, below)
        u64 sc[10];
        {
            // one host synchronisation for the three kernels (the flags are needed before anything else is decided)
            const u64 n_tiles = (ctx->stream_len + SK1_OWN - 1) / SK1_OWN;
            const size_t dig_bytes = (size_t)3 * 4 * (1u << p.l1);
            const u64 nw64 = (ctx->stream_len + 63) / 64 + PB_STREAM_PAD_WORDS64;
            const unsigned g1 = grid_for(n_tiles, 1, 256 * 16);
            // a few blocks per sub-stream, each striding over its tiles (most of cap1 is head-room: one block per tile would be mostly empty blocks)
            const u64 exp_tiles = (u64)(est_rec / (double)nb1) / SK2Q_RCAP + 1;
            const unsigned blocks_per_sub = (unsigned)std::min<u64>(std::max<u64>(exp_tiles / 4, 1), 64);
            const unsigned g3 = grid_for(nb2, 1, ctx->opt_bucket_blocks ? (unsigned)ctx->opt_bucket_blocks : 256 * 32);      // (r3 sweep, cfg2: 512 blocks 1.21 ms, 4096 1.17, 8192 1.13-1.16, 32768 1.26)
            const size_t meta_bytes = (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4;
            const u64 *mask = cur_mask(ctx);
            if (!ctx->h_hist) HIPCHK(hipHostMalloc((void **)&ctx->h_hist, 65536 * 8, hipHostMallocDefault));
            if (!ctx->h_scalars) HIPCHK(hipHostMalloc((void **)&ctx->h_scalars, 16 * 8, hipHostMallocDefault));
            // everything the pass enqueues, in order; `timed`: with the per-kernel event pairs (not inside a capture)
            auto enqueue = [&](bool timed, u64 *sc_dst) -> hipError_t {
                hipError_t e;
                if ((e = hipMemsetAsync(ctx->d_hist.p, 0, 65536 * 8, ctx->stream)) != hipSuccess) return e;
                if ((e = hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream)) != hipSuccess) return e;
                if ((e = hipMemsetAsync(ctx->part_meta.p, 0, meta_bytes, ctx->stream)) != hipSuccess) return e;
                LazyTimer t1(ctx, "part1", 0, timed);
                if (want_pos) sk_part1_kernel<0, true><<<g1, SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, mask, ctx->stream_len, nw64, p);
                else if (k == 32) sk_part1_kernel<32, false><<<g1, SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, mask, ctx->stream_len, nw64, p);
                else if (k == 31) sk_part1_kernel<31, false><<<g1, SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, mask, ctx->stream_len, nw64, p);   // (the usual odd k)
                else sk_part1_kernel<0, false><<<g1, SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, mask, ctx->stream_len, nw64, p);
                t1.stop();
                LazyTimer t2(ctx, "part2", 1, timed);
                if (want_pos) sk_part2_raw16_kernel<true><<<dim3(blocks_per_sub, (unsigned)nb1), SK_THREADS, 0, ctx->stream>>>(p);
                else sk_part2_raw16_kernel<false><<<dim3(blocks_per_sub, (unsigned)nb1), SK_THREADS, 0, ctx->stream>>>(p);
                t2.stop();
                LazyTimer t3(ctx, "bucket", 2, timed);
                if (ctx->opt_bucket_form == 2) {
                    if (want_pos) sk_bucket2_kernel<0, true><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, (u32 *)ctx->sk_first.p);
                    else if (k == 32) sk_bucket2_kernel<32, false><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, nullptr);
                    else sk_bucket2_kernel<0, false><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, nullptr);
                }
                else if (want_pos) sk_bucket_kernel<0, true><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, (u32 *)ctx->sk_first.p);
                else if (k == 32) sk_bucket_kernel<32, false><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, nullptr);
                else sk_bucket_kernel<0, false><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, nullptr);
                t3.stop();
                if ((e = hipGetLastError()) != hipSuccess) return e;
                if ((e = hipMemcpyAsync(sc_dst, ctx->d_scalars.p, 80, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess) return e;
                // the occurrence distribution (final unless buckets were flagged).  A replay takes only the head with it: sk_bucket_kernel tallies
                // counts below SK_LHIST in LDS bins and says in scalars[9] how far the others reach; the tail follows only if there is one
                return hipMemcpyAsync(ctx->h_hist, ctx->d_hist.p, (timed ? 65536 : SK_HIST_HEAD) * 8, hipMemcpyDeviceToHost, ctx->stream);
            };
            // what a captured pass depends on (kernel arguments are copied into the graph at capture)
            std::vector<u64> key = {(u64)k, (u64)p.m, (u64)p.w, (u64)p.B, (u64)p.l1, (u64)p.l2, p.cap1, (u64)p.cap2, (u64)(uintptr_t)p.bucket1, (u64)p.nx, (u64)(uintptr_t)p.cursor1,
                                    (u64)(uintptr_t)p.bucket2, (u64)(uintptr_t)p.cursor2, (u64)(uintptr_t)p.flag2, (u64)(uintptr_t)p.fallback.slots, p.fallback.mask,
                                    (u64)(uintptr_t)p.spill, p.spill_cap, (u64)(uintptr_t)p.scalars, (u64)(uintptr_t)p.hist, (u64)(uintptr_t)ctx->packed.p, (u64)(uintptr_t)mask,
                                    ctx->stream_len, nw64, (u64)g1, (u64)g3, (u64)blocks_per_sub, (u64)nb1, (u64)dig_bytes, (u64)meta_bytes, (u64)(uintptr_t)ctx->dist_keys.p,
                                    (u64)(uintptr_t)ctx->dist_counts.p, out_cap, (u64)(uintptr_t)ctx->h_hist, (u64)(uintptr_t)ctx->h_scalars, (u64)ctx->opt_bucket_form};
            const bool want_graph = ctx->opt_graph && !want_pos && !ctx->opt_verbose;
            ctx->stage_ms["graph_replay"] = 0.f;
            if (want_graph && ctx->sk_graph && key == ctx->sk_graph_key) {
                // the pass as it was captured: one launch; its duration (events around it) is the pass's kernel time -- the per-kernel
                // entries keep the values of the kernel-by-kernel pass over the same buffers
                HIPCHK(hipEventRecord(ctx->ev0, ctx->stream));
                HIPCHK(hipGraphLaunch(ctx->sk_graph, ctx->stream));
                HIPCHK(hipEventRecord(ctx->ev1, ctx->stream));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                float ms = 0;
                (void)hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
                ctx->stage_ms["graph_replay"] = 1.f;
                ctx->stage_ms["graph_ms"] = ms;
                memcpy(sc, ctx->h_scalars, 80);
                const u64 reach = sc[9] + 1;                          // bins [0, reach) may be non-zero (the LDS bins end at SK_LHIST <= SK_HIST_HEAD)
                if (reach > SK_HIST_HEAD) {
                    HIPCHK(hipMemcpyAsync(ctx->h_hist + SK_HIST_HEAD, (const u64 *)ctx->d_hist.p + SK_HIST_HEAD, (std::min<u64>(reach, 65536) - SK_HIST_HEAD) * 8, hipMemcpyDeviceToHost, ctx->stream));
                    HIPCHK(hipStreamSynchronize(ctx->stream));
                }
                const u64 now_hi = std::max<u64>(std::min<u64>(reach, 65536), SK_HIST_HEAD);
                if (ctx->h_hist_hi > now_hi) memset(ctx->h_hist + now_hi, 0, (ctx->h_hist_hi - now_hi) * 8);    // (what an earlier pass left beyond)
                ctx->h_hist_hi = now_hi;
            } else {
                HIPCHK(enqueue(true, sc));
                HIPCHK(hipStreamSynchronize(ctx->stream));
                ctx->h_hist_hi = 65536;
                for (int sl = 0; sl < 3; ++sl) { LazyTimer t(ctx, sl == 0 ? "part1" : sl == 1 ? "part2" : "bucket", sl, false); t.read(); }
                ctx->stage_ms["graph_ms"] = 0.f;
                if (want_graph) {                                     // the same sequence, captured for the next pass over these buffers
                    if (ctx->sk_graph) { (void)hipGraphExecDestroy(ctx->sk_graph); ctx->sk_graph = nullptr; }
                    ctx->sk_graph_key.clear();
                    hipGraph_t gr = nullptr;
                    hipError_t e = hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed);
                    if (e == hipSuccess) {
                        const hipError_t e1 = enqueue(false, ctx->h_scalars);
                        e = hipStreamEndCapture(ctx->stream, &gr);
                        if (e == hipSuccess) e = e1;
                    }
                    if (e == hipSuccess) e = hipGraphInstantiate(&ctx->sk_graph, gr, nullptr, nullptr, 0);
                    if (gr) (void)hipGraphDestroy(gr);
                    if (e == hipSuccess) ctx->sk_graph_key = key;
                    else { ctx->sk_graph = nullptr; (void)hipGetLastError(); }      // (no graph: the pass goes on kernel by kernel)
                }
            }
        }
        if (ctx->opt_verbose)
            fprintf(stderr, "[pbgpu] super-k-mer pass: k %u, 2^%u buckets (l1 %u), cap1 %llu cap2 %u: windows %llu, records %llu, distinct %llu, flagged buckets %llu, "
                            "spilled records %llu, overflow %llu; buffers %.3f s (%.1f GB held), pass %.3f s\n", k, B, p.l1, (unsigned long long)p.cap1, p.cap2, (unsigned long long)sc[0], (unsigned long long)sc[6],
                    (unsigned long long)sc[4], (unsigned long long)sc[7], (unsigned long long)sc[8], (unsigned long long)sc[1], alloc_s,
                    (double)(ctx->part_bucket1.bytes + ctx->part_bucket2.bytes + ctx->dist_keys.bytes + ctx->dist_counts.bytes + ctx->table.bytes + ctx->part_spill.bytes) / 1e9,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_attempt).count() - alloc_s);
#ifdef SK_DBG
        if (ctx->opt_verbose) {
            unsigned long long v[5] = {0, 0, 0, 0, 0};
            (void)hipMemcpyFromSymbol(&v[0], HIP_SYMBOL(sk_dbg_probe_iters), 8); (void)hipMemcpyFromSymbol(&v[1], HIP_SYMBOL(sk_dbg_flushes), 8);
            (void)hipMemcpyFromSymbol(&v[2], HIP_SYMBOL(sk_dbg_parked), 8); (void)hipMemcpyFromSymbol(&v[3], HIP_SYMBOL(sk_dbg_steps), 8);
            (void)hipMemcpyFromSymbol(&v[4], HIP_SYMBOL(sk_dbg_inserts), 8);
            {
                unsigned long long c1[16], t1 = 0;
                (void)hipMemcpyFromSymbol(c1, HIP_SYMBOL(sk1_dbg_cyc), 128);
                for (int i = 0; i < 16; ++i) t1 += c1[i];
                static const char *n1[16] = {"tile end: stage, clear, barrier", "barrier before the records leave", "hashes + store", "barrier 2", "a8 (neighbours, doubling, store)", "barrier 3",
                                             "window minima (3 x 17 reads)", "barrier 4", "run starts", "rank scan (2 barriers)", "reserve + first record build", "records leave, further builds, barrier",
                                             "list of starts", "barrier 5", "digit histogram", "barrier 6"};
                static const int order[16] = {2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 14, 15, 10, 1, 11, 0};
                for (int q = 0; q < 16; ++q) { const int i = order[q]; fprintf(stderr, "[pbgpu]   part1 wave cycles %-40s %5.1f %%\n", n1[i], 100.0 * (double)c1[i] / (double)t1); }
            }
            unsigned long long cyc[12];
            (void)hipMemcpyFromSymbol(cyc, HIP_SYMBOL(sk_dbg_cyc), 96);
            unsigned long long tot = 0;
            for (int i = 0; i < 12; ++i) tot += cyc[i];
            static const char *nm1[12] = {"bucket prologue", "wait for the records", "phase A", "barrier A", "scan + piece map", "keys + CAS + add", "parking", "clear + last flush", "barrier B", "emit", "", ""};
            static const char *nm2[12] = {"loop top (next bucket's size)", "phase A (all records, piece list)", "barrier A", "phase B rounds", "barrier B", "clear record table", "emit: rows + counts", "emit: barrier 1",
                                          "emit: scan + reservation + barrier 2", "emit: stores + histogram", "last flush of the parked keys", ""};
            const char **nm = ctx->opt_bucket_form == 2 ? nm2 : nm1;
            for (int i = 0; i < 11; ++i) fprintf(stderr, "[pbgpu]   wave cycles %-38s %5.1f %%\n", nm[i], 100.0 * (double)cyc[i] / (double)tot);
            { unsigned long long z[12] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(sk_dbg_cyc), z, 96); }
            fprintf(stderr, "[pbgpu]   dbg (cumulative): slow-probe iterations %llu, queue flushes %llu, parked keys %llu, piece steps %llu, window inserts %llu\n", v[0], v[1], v[2], v[3], v[4]);
        }
#endif
        if (ctx->opt_verbose && sc[7]) {
            std::vector<u32> fl(nb2), cu(nb2);
            (void)hipMemcpy(fl.data(), p.flag2, nb2 * 4, hipMemcpyDeviceToHost);
            (void)hipMemcpy(cu.data(), p.cursor2, nb2 * 4, hipMemcpyDeviceToHost);
            std::map<u32, std::pair<u64, u64>> by;
            u64 mx = 0;
            for (u64 i = 0; i < nb2; ++i) { by[fl[i]].first++; by[fl[i]].second += cu[i]; mx = std::max<u64>(mx, cu[i]); }
            for (auto &kv : by) fprintf(stderr, "[pbgpu]   flag %u: %llu buckets, %llu records\n", kv.first, (unsigned long long)kv.second.first, (unsigned long long)kv.second.second);
            fprintf(stderr, "[pbgpu]   largest bucket %llu records\n", (unsigned long long)mx);
        }
        // sc[7] buckets had more distinct keys than an LDS table holds.  Many of them: the input is not the high-coverage kind
        // the bucket count was chosen for -- repeat with 4x the buckets rather than push a large share through the global table.
        if (!forced_b && sc[7] > nb2 / 64 && B < SK_MAX_B) {
            B = std::min((unsigned)SK_MAX_B, B + 2);
            ctx->sk_extra_bits += 2;
            ctx->sk_extra_for = windows;
            continue;
        }
        if (!forced_b && sc[7] > nb2 / 4) return PB_OK;            // still mostly overflowing at 2^20 buckets: the direct path takes it
        if (sc[4] > out_cap && out_cap < out_bound) {              // the estimate above was too small: once more, with what this pass counted
            ctx->sk_out_need = sc[4];
            ctx->sk_out_for = windows;
            continue;
        }
        ctx->stage_ms["seg_compact"] = 0.f;
        const bool all_in_lds = sc[7] + sc[8] == 0;                // then h_hist already holds the whole histogram
        if (want_pos && !all_in_lds) return PB_OK;                 // records that left the LDS path carry no position: the caller takes the probing kernel
        if (!all_in_lds) {
            // the global table is sized NOW, from what the pass says goes through it (spilled records + the records of the flagged buckets,
            // at the pass's own windows per record), instead of failing at 2^20 slots and repeating the whole pass with 16 x the table:
            // a 5 Mbp genome at 1000x spills 1.3 % of its records on the first pass (one minimizer site per bucket: Poisson sizes)
            {
                const double wpr = sc[6] ? (double)sc[0] / (double)sc[6] : (double)p.w;
                const double need = ((double)sc[8] + (double)sc[7] * (double)p.cap2) * wpr * 1.5;
                u64 want = fcap;
                while ((double)want < need && want < (1ull << 34)) want <<= 1;
                if (want > fcap) {
                    if ((rc = ensure(ctx, ctx->table, want * 16)) != PB_OK) return rc;
                    if ((rc = ensure(ctx, ctx->dist_keys, (sc[4] + want) * 8, true)) != PB_OK) return rc;      // (no-ops while out_cap's head-room covers it)
                    if ((rc = ensure(ctx, ctx->dist_counts, (sc[4] + want) * 4, true)) != PB_OK) return rc;
                    fcap = want;
                    ctx->table_cap = fcap;
                    p.fallback = TableView<1>{(u64 *)ctx->table.p, fcap - 1};
                    ctx->sk_fcap = fcap;
                    ctx->sk_fcap_for = windows;
                }
            }
            StageTimer t4(ctx, "seg_compact");                     // (stage name kept for bench.py: flagged buckets through the global table)
            table_init_kernel<1><<<grid_for(fcap * 2, 256, 256 * 32), 256, 0, ctx->stream>>>(p.fallback.slots, fcap);
            sk_flagged_kernel<<<dim3(grid_for(nb2, 1, 1024), 16), 256, 0, ctx->stream>>>(p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 72, hipMemcpyDeviceToHost, ctx->stream));
            if ((rc = t4.stop()) != PB_OK) return rc;
        }
        if (sc[1] == 0) {
            ctx->n_instances = sc[0];
            ctx->n_distinct = sc[4];
            ctx->hist_done = sc[4];                 // sk_bucket_kernel tallied its records while it wrote them
            ctx->h_hist_valid = all_in_lds;
            ctx->sk_records = sc[6];
            ctx->sk_bits = B;
            ctx->sk_flagged = sc[7];
            ctx->sk_spilled = sc[8];
            ctx->stage_ms["sk_bucket_bits"] = (float)B;                    // pass statistics, read like the stage times
            ctx->stage_ms["sk_flagged_buckets"] = (float)sc[7];
            ctx->stage_ms["sk_spilled_records"] = (float)sc[8];
            ctx->stage_ms["sk_records_per_1k_windows"] = sc[0] ? (float)(1000.0 * (double)sc[6] / (double)sc[0]) : 0.f;
            break;
        }
        if (sc[8] > p.spill_cap) return PB_OK;   // a few minimizers carry most of the input (tiny k, low-complexity reads): the direct path takes it
        if (sc[8]) { room2 = 6.0; ctx->sk_room6_for = windows; }        // (records spilled AND the table sized for them overflowed: larger buckets, remembered for this stream)
        fcap <<= 4;                     // the fallback table overflowed: retry with a larger one (remembered for the next pass over this stream)
        ctx->sk_fcap = fcap;
        ctx->sk_fcap_for = windows;
        if (attempt == 11 || fcap > (1ull << 36)) { ctx->err = "fallback table overflow"; return PB_ENOMEM; }
    }
    // kernel time of the pass: the three kernels' event pairs, or -- a replayed graph -- the events around its one launch (the memsets, the
    // two boundaries between the kernels and the copy-out are then inside: what the stream was busy for)
    ctx->stage_ms["count"] = (ctx->stage_ms["graph_replay"] > 0.f ? ctx->stage_ms["graph_ms"] : ctx->stage_ms["part1"] + ctx->stage_ms["part2"] + ctx->stage_ms["bucket"]) +
                             ctx->stage_ms["seg_compact"];
    ctx->stage_ms["compact"] = 0.f;
    if (ctx->sk_flagged + ctx->sk_spilled > 0) {                    // otherwise nothing went into the global table
        if ((rc = table_to_arrays<1>(ctx, ctx->table_cap, true)) != PB_OK) return rc;
    }
    *done = true;
    return PB_OK;
}

// Super-k-mer pass for two-word keys (33 <= k <= 64; kernels_superk2.inc).  *done = false: the input is not the kind the pass is made for
// (a few minimizers carry most of it, or nearly every window is a distinct key even with 16 x the buckets): the caller takes the direct pass.
static int count_superk2(pb_ctx *ctx, unsigned k, bool *done)
{
    int rc;
    *done = false;
    const u64 lost = (u64)k * ctx->n_reads;
    const u64 windows = ctx->stream_len > lost + 4096 ? ctx->stream_len - lost : ctx->stream_len;
    Sk2Params p;
    memset(&p, 0, sizeof(p));
    p.k = k;
    p.m = k - (SK2_WIN - 1);
    // ~2 * SK2_SLOTS windows per final bucket: at sequencing depth that is a third of the table in distinct keys
    unsigned B = 8;
    while (((u64)(SK2_SLOTS * 2) << B) < windows && B < SK_MAX_B) ++B;
    const bool forced_b = ctx->sk_force_bits >= 2 && ctx->sk_force_bits <= SK_MAX_B;
    if (forced_b) B = (unsigned)ctx->sk_force_bits;
    u64 fcap = ctx->sk_fcap && ctx->sk_fcap_for == windows + k ? ctx->sk_fcap : 1ull << 20;
    for (int attempt = 0; attempt < 8; ++attempt) {
        p.B = B;
        p.l1 = std::max(B / 2, B > 9 ? B - 9 : 0u);
        p.l2 = B - p.l1;
        p.nx = PB_XCD;
        const double est_rec = (double)windows * 2.0 / (SK2_WIN + 1.0) + 2.0 * (double)ctx->n_reads;
        const u64 nb1 = (1ull << p.l1) * PB_XCD, nb2 = 1ull << B;
        p.cap1 = (u64)(est_rec / (double)nb1 * 1.5) + 4096;
        p.cap2 = (u32)std::min<double>(est_rec / (double)nb2 * 3.0 + 256.0, 1e9);
        p.spill_cap = std::max<u64>((u64)(est_rec / 16), 1ull << 16);
        if ((rc = ensure(ctx, ctx->part_bucket1, nb1 * p.cap1 * 24)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->part_bucket2, nb2 * (u64)p.cap2 * 24)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->part_meta, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->part_spill, p.spill_cap * 24)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->table, fcap * PB_SLOT_WORDS(2) * 8)) != PB_OK) return rc;
        ctx->table_cap = fcap;
        p.bucket1 = (u64 *)ctx->part_bucket1.p;
        p.bucket2 = (u64 *)ctx->part_bucket2.p;
        p.cursor1 = (u32 *)ctx->part_meta.p;
        p.cursor2 = p.cursor1 + nb1 * SK_CUR1_STRIDE;
        p.flag2 = p.cursor2 + nb2;
        p.spill = (u64 *)ctx->part_spill.p;
        p.fallback = TableView<2>{(u64 *)ctx->table.p, fcap - 1};
        p.scalars = (u64 *)ctx->d_scalars.p;
        const u64 out_cap = std::min<u64>(std::max<u64>(ctx->stream_len, 1), nb2 * (u64)SK2_SLOT_LIMIT);
        if ((rc = ensure(ctx, ctx->dist_keys, (out_cap + fcap) * 16)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->dist_counts, (out_cap + fcap) * 4)) != PB_OK) return rc;
        HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->part_meta.p, 0, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4, ctx->stream));
        table_init_kernel<2><<<grid_for(fcap * PB_SLOT_WORDS(2), 256, 256 * 32), 256, 0, ctx->stream>>>(p.fallback.slots, fcap);
        u64 sc[9];
        {
            LazyTimer t1(ctx, "part1", 0);
            const u64 n_tiles = (ctx->stream_len + SK2_OWN - 1) / SK2_OWN;
            const size_t dig_bytes = (size_t)2 * 4 * (1u << p.l1);
            const u64 nw64 = (ctx->stream_len + 63) / 64 + PB_STREAM_PAD_WORDS64;
            if (p.m <= 32) sk2_part1_kernel<true><<<grid_for(n_tiles, 1, 256 * 16), SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, cur_mask(ctx), ctx->stream_len, nw64, p);
            else sk2_part1_kernel<false><<<grid_for(n_tiles, 1, 256 * 16), SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, cur_mask(ctx), ctx->stream_len, nw64, p);
            t1.stop();
            LazyTimer t2(ctx, "part2", 1);
            const u64 exp_tiles = (u64)(est_rec / (double)nb1) / SK2P_RCAP + 1;
            const unsigned blocks_per_sub = (unsigned)std::min<u64>(std::max<u64>(exp_tiles / 4, 1), 64);
            sk2_part2_kernel<<<dim3(blocks_per_sub, (unsigned)nb1), SK_THREADS, 0, ctx->stream>>>(p);
            t2.stop();
            LazyTimer t3(ctx, "bucket", 2);
            if (ctx->opt_bucket_form == 2) sk2_bucket2_kernel<<<grid_for(nb2, 1, ctx->opt_bucket_blocks ? (unsigned)ctx->opt_bucket_blocks : 256 * 16), SK2_BUCKET_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap);
            else sk2_bucket_kernel<<<grid_for(nb2, 1, 256 * 16), SK2_BUCKET_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap);
            t3.stop();
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 72, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            t1.read(); t2.read(); t3.read();
        }
        if (ctx->opt_verbose)
            fprintf(stderr, "[pbgpu] super-k-mer pass (two-word keys): k %u, m %u, 2^%u buckets (l1 %u), cap1 %llu cap2 %u: windows %llu, records %llu, distinct %llu, abandoned buckets %llu, "
                            "spilled records %llu, overflow %llu\n", k, p.m, B, p.l1, (unsigned long long)p.cap1, p.cap2, (unsigned long long)sc[0], (unsigned long long)sc[6], (unsigned long long)sc[4],
                    (unsigned long long)sc[7], (unsigned long long)sc[8], (unsigned long long)sc[1]);
        if (sc[8] > p.spill_cap) break;                              // a few minimizers carry most of the input: the direct pass takes it
        // many buckets with more distinct keys than a table holds: not the coverage the bucket count was chosen for -- 4 x the buckets
        if (!forced_b && sc[7] > nb2 / 64 && B < SK_MAX_B) { B = std::min((unsigned)SK_MAX_B, B + 2); continue; }
        if (!forced_b && sc[7] > nb2 / 4) break;
        ctx->stage_ms["seg_compact"] = 0.f;
        if (sc[7] + sc[8] > 0) {
            StageTimer t4(ctx, "seg_compact");                       // (stage name as in the k <= 32 pass: flagged buckets through the global table)
            sk2_flagged_kernel<<<dim3(grid_for(nb2, 1, 1024), 16), 256, 0, ctx->stream>>>(p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 72, hipMemcpyDeviceToHost, ctx->stream));
            if ((rc = t4.stop()) != PB_OK) return rc;
        }
        if (sc[1] == 0) {
            ctx->n_instances = sc[0];
            ctx->n_distinct = sc[4];
            ctx->hist_done = 0;
            ctx->h_hist_valid = false;
            ctx->sk_records = sc[6];
            ctx->sk_bits = B;
            ctx->sk_flagged = sc[7];
            ctx->sk_spilled = sc[8];
            ctx->stage_ms["sk_bucket_bits"] = (float)B;
            ctx->stage_ms["sk_flagged_buckets"] = (float)sc[7];
            ctx->stage_ms["sk_spilled_records"] = (float)sc[8];
            ctx->stage_ms["sk_records_per_1k_windows"] = sc[0] ? (float)(1000.0 * (double)sc[6] / (double)sc[0]) : 0.f;
            ctx->stage_ms["compact"] = 0.f;
            ctx->stage_ms["count"] = ctx->stage_ms["part1"] + ctx->stage_ms["part2"] + ctx->stage_ms["bucket"] + ctx->stage_ms["seg_compact"];
            if (sc[7] + sc[8] > 0 && (rc = table_to_arrays<2>(ctx, ctx->table_cap, true)) != PB_OK) return rc;      // what went through the global table
            *done = true;
            return PB_OK;
        }
        fcap <<= 4;                                                   // the fallback table overflowed: once more with a larger one (remembered for this stream)
        ctx->sk_fcap = fcap;
        ctx->sk_fcap_for = windows + k;
        if (fcap > (1ull << 34)) break;
    }
    ctx->stage_ms["part1"] = ctx->stage_ms["part2"] = ctx->stage_ms["bucket"] = ctx->stage_ms["seg_compact"] = 0.f;
    return PB_OK;
}

// Super-k-mer pass for three-word keys (65 <= k <= 96; kernels_superk3.inc): the same host side with 32-byte records.  *done = false: the input is not the kind the pass is made for
// (a few minimizers carry most of it, or nearly every window is a distinct key even with 16 x the buckets): the caller takes the direct pass.
static int count_superk3(pb_ctx *ctx, unsigned k, bool *done)
{
    int rc;
    *done = false;
    const u64 lost = (u64)k * ctx->n_reads;
    const u64 windows = ctx->stream_len > lost + 4096 ? ctx->stream_len - lost : ctx->stream_len;
    SkwParams<3> p;
    memset(&p, 0, sizeof(p));
    p.k = k;
    p.m = k - (SK2_WIN - 1);
    // ~2 * SK2_SLOTS windows per final bucket: at sequencing depth that is a third of the table in distinct keys
    unsigned B = 8;
    while (((u64)(SK2_SLOTS * 3 / 2) << B) < windows && B < SK_MAX_B) ++B;      // (fewer windows per bucket than the two-word pass: at these k a third of the windows of a 100x set are distinct keys)
    const bool forced_b = ctx->sk_force_bits >= 2 && ctx->sk_force_bits <= SK_MAX_B;
    if (forced_b) B = (unsigned)ctx->sk_force_bits;
    u64 fcap = ctx->sk_fcap && ctx->sk_fcap_for == windows + k ? ctx->sk_fcap : 1ull << 20;
    for (int attempt = 0; attempt < 8; ++attempt) {
        p.B = B;
        p.l1 = std::max(B / 2, B > 9 ? B - 9 : 0u);
        p.l2 = B - p.l1;
        p.nx = PB_XCD;
        const double est_rec = (double)windows * 2.0 / (SK2_WIN + 1.0) + 2.0 * (double)ctx->n_reads;
        const u64 nb1 = (1ull << p.l1) * PB_XCD, nb2 = 1ull << B;
        p.cap1 = (u64)(est_rec / (double)nb1 * 1.5) + 4096;
        p.cap2 = (u32)std::min<double>(est_rec / (double)nb2 * 3.0 + 256.0, 1e9);
        p.spill_cap = std::max<u64>((u64)(est_rec / 16), 1ull << 16);
        if ((rc = ensure(ctx, ctx->part_bucket1, nb1 * p.cap1 * 32)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->part_bucket2, nb2 * (u64)p.cap2 * 32)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->part_meta, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->part_spill, p.spill_cap * 32)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->table, fcap * PB_SLOT_WORDS(3) * 8)) != PB_OK) return rc;
        ctx->table_cap = fcap;
        p.bucket1 = (u64 *)ctx->part_bucket1.p;
        p.bucket2 = (u64 *)ctx->part_bucket2.p;
        p.cursor1 = (u32 *)ctx->part_meta.p;
        p.cursor2 = p.cursor1 + nb1 * SK_CUR1_STRIDE;
        p.flag2 = p.cursor2 + nb2;
        p.spill = (u64 *)ctx->part_spill.p;
        p.fallback = TableView<3>{(u64 *)ctx->table.p, fcap - 1};
        p.scalars = (u64 *)ctx->d_scalars.p;
        const u64 out_cap = std::min<u64>(std::max<u64>(ctx->stream_len, 1), nb2 * (u64)SK2_SLOT_LIMIT);
        if ((rc = ensure(ctx, ctx->dist_keys, (out_cap + fcap) * 24)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->dist_counts, (out_cap + fcap) * 4)) != PB_OK) return rc;
        HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->part_meta.p, 0, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4, ctx->stream));
        table_init_kernel<3><<<grid_for(fcap * PB_SLOT_WORDS(3), 256, 256 * 32), 256, 0, ctx->stream>>>(p.fallback.slots, fcap);
        u64 sc[9];
        {
            LazyTimer t1(ctx, "part1", 0);
            const u64 n_tiles = (ctx->stream_len + SK2_OWN - 1) / SK2_OWN;
            const size_t dig_bytes = (size_t)2 * 4 * (1u << p.l1);
            const u64 nw64 = (ctx->stream_len + 63) / 64 + PB_STREAM_PAD_WORDS64;
            sk3_part1_kernel<<<grid_for(n_tiles, 1, 256 * 16), SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, cur_mask(ctx), ctx->stream_len, nw64, p);
            t1.stop();
            LazyTimer t2(ctx, "part2", 1);
            const u64 exp_tiles = (u64)(est_rec / (double)nb1) / SK3P_RCAP + 1;
            const unsigned blocks_per_sub = (unsigned)std::min<u64>(std::max<u64>(exp_tiles / 4, 1), 64);
            sk3_part2_kernel<<<dim3(blocks_per_sub, (unsigned)nb1), SK_THREADS, 0, ctx->stream>>>(p);
            t2.stop();
            LazyTimer t3(ctx, "bucket", 2);
            sk3_bucket_kernel<<<grid_for(nb2, 1, ctx->opt_bucket_blocks ? (unsigned)ctx->opt_bucket_blocks : 256 * 16), SK2_BUCKET_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap);
            t3.stop();
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 72, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            t1.read(); t2.read(); t3.read();
        }
        if (ctx->opt_verbose)
            fprintf(stderr, "[pbgpu] super-k-mer pass (three-word keys): k %u, m %u, 2^%u buckets (l1 %u), cap1 %llu cap2 %u: windows %llu, records %llu, distinct %llu, abandoned buckets %llu, "
                            "spilled records %llu, overflow %llu\n", k, p.m, B, p.l1, (unsigned long long)p.cap1, p.cap2, (unsigned long long)sc[0], (unsigned long long)sc[6], (unsigned long long)sc[4],
                    (unsigned long long)sc[7], (unsigned long long)sc[8], (unsigned long long)sc[1]);
        if (sc[8] > p.spill_cap) break;                              // a few minimizers carry most of the input: the direct pass takes it
        // many buckets with more distinct keys than a table holds: not the coverage the bucket count was chosen for -- 4 x the buckets
        if (!forced_b && sc[7] > nb2 / 64 && B < SK_MAX_B) { B = std::min((unsigned)SK_MAX_B, B + 2); continue; }
        if (!forced_b && sc[7] > nb2 / 4) break;
        ctx->stage_ms["seg_compact"] = 0.f;
        if (sc[7] + sc[8] > 0) {
            StageTimer t4(ctx, "seg_compact");                       // (stage name as in the k <= 32 pass: flagged buckets through the global table)
            sk3_flagged_kernel<<<dim3(grid_for(nb2, 1, 1024), 16), 256, 0, ctx->stream>>>(p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 72, hipMemcpyDeviceToHost, ctx->stream));
            if ((rc = t4.stop()) != PB_OK) return rc;
        }
        if (sc[1] == 0) {
            ctx->n_instances = sc[0];
            ctx->n_distinct = sc[4];
            ctx->hist_done = 0;
            ctx->h_hist_valid = false;
            ctx->sk_records = sc[6];
            ctx->sk_bits = B;
            ctx->sk_flagged = sc[7];
            ctx->sk_spilled = sc[8];
            ctx->stage_ms["sk_bucket_bits"] = (float)B;
            ctx->stage_ms["sk_flagged_buckets"] = (float)sc[7];
            ctx->stage_ms["sk_spilled_records"] = (float)sc[8];
            ctx->stage_ms["sk_records_per_1k_windows"] = sc[0] ? (float)(1000.0 * (double)sc[6] / (double)sc[0]) : 0.f;
            ctx->stage_ms["compact"] = 0.f;
            ctx->stage_ms["count"] = ctx->stage_ms["part1"] + ctx->stage_ms["part2"] + ctx->stage_ms["bucket"] + ctx->stage_ms["seg_compact"];
            if (sc[7] + sc[8] > 0 && (rc = table_to_arrays<3>(ctx, ctx->table_cap, true)) != PB_OK) return rc;      // what went through the global table
            *done = true;
            return PB_OK;
        }
        fcap <<= 4;                                                   // the fallback table overflowed: once more with a larger one (remembered for this stream)
        ctx->sk_fcap = fcap;
        ctx->sk_fcap_for = windows + k;
        if (fcap > (1ull << 34)) break;
    }
    ctx->stage_ms["part1"] = ctx->stage_ms["part2"] = ctx->stage_ms["bucket"] = ctx->stage_ms["seg_compact"] = 0.f;
    return PB_OK;
}

// scratch[2] (keys, n * W words) and scratch[3] (u32 counts) -> ctx->ret_keys / ret_counts, ascending by the 2k-bit key value
// (counter.h:916-951 std::sort); the sorted set becomes the context's retained set
static int sort_retained(pb_ctx *ctx, u64 n)
{
    int rc;
    const int W = ctx->W;
    DevBuf &ck = ctx->scratch[2], &cc = ctx->scratch[3];
    ctx->n_retained = n;
    if ((rc = ensure(ctx, ctx->ret_keys, std::max<u64>(n, 1) * W * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->ret_counts, std::max<u64>(n, 1) * 2)) != PB_OK) return rc;
    if (n == 0) return PB_OK;
    if (n >= (1ull << 32)) { ctx->err = "more than 2^32 retained k-mers on one GPU"; return PB_ERANGE; }
    // LSD radix sort of a permutation, one key word at a time (least significant word first)
    const u32 n_tiles = (u32)((n + RS_TILE - 1) / RS_TILE);
    DevBuf &sk_a = ctx->scratch[4], &sk_b = ctx->scratch[5], &pv = ctx->scratch[0], &cn = ctx->scratch[1];
    if ((rc = ensure(ctx, sk_a, n * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, sk_b, n * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, pv, n * 4 * 2)) != PB_OK) return rc;
    if ((rc = ensure(ctx, cn, ((u64)256 * n_tiles + (u64)256 * n_tiles / RS_SCAN_CHUNK + 16) * 4)) != PB_OK) return rc;   // digit counts + scan block sums
    u32 *perm_a = (u32 *)pv.p, *perm_b = perm_a + n;
    StageTimer tm(ctx, "sort");
    iota_kernel<<<grid_for(n, 256), 256, 0, ctx->stream>>>(perm_a, n);
    const unsigned total_bits = 2 * ctx->k;
    for (int w = 0; w < W; ++w) {
        const unsigned bits = std::min<unsigned>(64u, total_bits - 64u * (unsigned)w);
        gather_word_kernel<<<grid_for(n, 256), 256, 0, ctx->stream>>>((const u64 *)ck.p, perm_a, n, W, w, (u64 *)sk_a.p);
        if ((rc = radix_sort_pairs(ctx, (u64 *)sk_a.p, perm_a, (u64 *)sk_b.p, perm_b, n, bits, (u32 *)cn.p, n_tiles)) != PB_OK) return rc;
    }
    apply_perm_kernel<<<grid_for(n, 256), 256, 0, ctx->stream>>>((const u64 *)ck.p, (const u32 *)cc.p, perm_a, n, W,
                                                                (u64 *)ctx->ret_keys.p, (u16 *)ctx->ret_counts.p);
    HIPCHK(hipGetLastError());
    return tm.stop();
}

// ---- X1, second design: the records are exchanged BEFORE they are counted -------------------------------------------------
// The minimizer bucket of a k-mer is a function of the k-mer, so "rank = top bits of the bucket" is a partition of the key space as
// good as the reference's stripes: every rank cuts ITS reads into run records whose level-1 digit is the owner rank
// (pb_sk_owner_records), one all-to-all moves the records, and every rank runs the level-1 / level-2 scatter and the bucket
// kernel on what it received (pb_sk_count_records).  No export, no owner-side merge: copies of a run from different ranks meet in
// sk_bucket_kernel's record table like copies from different reads.
static unsigned sk_bits_for(u64 windows)
{
    unsigned B = 8;
    while (((u64)(SK_SLOTS * 3) << B) < windows && B < SK_MAX_B) ++B;
    return B;
}

extern "C" {

// bucket bits the super-k-mer pass would use for `windows` k-mer instances on one GPU (all ranks must agree on local_bits below)
unsigned pb_sk_bucket_bits(uint64_t windows) { return std::min(sk_bits_for(windows), 18u); }

// Stage 1: run records of this rank's reads, grouped by owner (n_owners a power of two >= 2; owner = top bits of the
// (log2(n_owners) + local_bits)-bit bucket).  counts[o] = records for owner o (16 bytes each); *n_windows = windows they hold.
int pb_sk_owner_records(pb_ctx *ctx, unsigned k, unsigned n_owners, unsigned local_bits, uint64_t *counts, uint64_t *n_windows)
{
    if (!ctx || !counts || k < 1 || k > 32 || n_owners < 2 || n_owners > PB_MAX_PARTS || (n_owners & (n_owners - 1)) || local_bits < 2 || local_bits > 18)
        return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    int rc;
    if ((rc = ensure(ctx, ctx->d_scalars, 16 * 8)) != PB_OK) return rc;
    unsigned ob = 0;
    while ((1u << ob) < n_owners) ++ob;
    if (ob + local_bits > 20) { ctx->err = "pb_sk_owner_records: owner bits + local bits must not exceed 20 (the record's bucket field)"; return PB_ERANGE; }
    const u64 lost = (u64)k * ctx->n_reads;
    const u64 windows = ctx->stream_len > lost + 4096 ? ctx->stream_len - lost : ctx->stream_len;
    SkParams p;
    memset(&p, 0, sizeof(p));
    p.k = k;
    p.m = k < SK_M ? k : SK_M;
    p.w = k - p.m + 1;
    p.B = ob + local_bits;
    p.l1 = ob;
    p.l2 = local_bits;
    // an owner's records are collected per XCD, like the level-1 sub-streams of the counting pass: with one cursor per owner every tile of
    // every block bumps the same few lines and the kernel runs at their atomic rate (measured: 1.55 instead of 0.8 ms)
    p.nx = PB_XCD;
    const u64 n_regions = (u64)n_owners * PB_XCD;
    const double est_rec = (double)windows * 2.0 / (p.w + 1.0) + 2.0 * (double)ctx->n_reads;
    p.cap1 = (u64)(est_rec / (double)n_regions * 1.5) + 16384;
    p.cap2 = 0;
    if ((rc = ensure(ctx, ctx->part_bucket1, n_regions * p.cap1 * SkStore16::BYTES)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->part_meta, (n_regions * SK_CUR1_STRIDE + 2ull * (1ull << p.B)) * 4)) != PB_OK) return rc;
    p.spill_cap = 1ull << 16;
    if ((rc = ensure(ctx, ctx->part_spill, p.spill_cap * SkStore16::BYTES)) != PB_OK) return rc;
    p.spill = ctx->part_spill.p;
    p.bucket1 = ctx->part_bucket1.p;
    p.cursor1 = (u32 *)ctx->part_meta.p;
    p.cursor2 = p.cursor1 + n_regions * SK_CUR1_STRIDE;
    p.flag2 = p.cursor2 + (1ull << p.B);                       // (only written when a record does not fit its owner's region)
    p.scalars = (u64 *)ctx->d_scalars.p;
    HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->part_meta.p, 0, (n_regions * SK_CUR1_STRIDE + 2ull * (1ull << p.B)) * 4, ctx->stream));
    {
        StageTimer t1(ctx, "part1");
        const u64 n_tiles = (ctx->stream_len + SK1_OWN - 1) / SK1_OWN;
        const size_t dig_bytes = (size_t)3 * 4 * (1u << p.l1);
        const u64 nw64 = (ctx->stream_len + 63) / 64 + PB_STREAM_PAD_WORDS64;
        const unsigned g1 = grid_for(n_tiles, 1, 256 * 16);
        if (ctx->stream_len >= k) {
            if (k == 32) sk_part1_kernel<32, false><<<g1, SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, cur_mask(ctx), ctx->stream_len, nw64, p);
            else sk_part1_kernel<0, false><<<g1, SK1_THREADS, dig_bytes, ctx->stream>>>((const u64 *)ctx->packed.p, cur_mask(ctx), ctx->stream_len, nw64, p);
        }
        HIPCHK(hipGetLastError());
        if ((rc = t1.stop()) != PB_OK) return rc;
    }
    u64 sc[10];
    std::vector<u32> cur_strided(n_regions * SK_CUR1_STRIDE);
    HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 80, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipMemcpyAsync(cur_strided.data(), p.cursor1, cur_strided.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (sc[8] || sc[1]) { ctx->err = "pb_sk_owner_records: an owner's region overflowed (a few minimizers carry most of the input)"; return PB_ERANGE; }
    ctx->sk_owner_cur.assign(n_regions, 0);
    for (unsigned o = 0; o < n_owners; ++o) {
        counts[o] = 0;
        for (unsigned x = 0; x < PB_XCD; ++x) counts[o] += ctx->sk_owner_cur[(u64)o * PB_XCD + x] = cur_strided[((u64)o * PB_XCD + x) * SK_CUR1_STRIDE];
        ctx->exp_cur[o] = counts[o];
    }
    ctx->exp_parts = n_owners;
    ctx->sk_owner_cap1 = p.cap1;
    ctx->n_instances = sc[0];
    if (n_windows) *n_windows = sc[0];
    return PB_OK;
}

// the records of pb_sk_owner_records, owner 0 first, into a caller device buffer of sum(counts) * 16 bytes
int pb_sk_owner_records_copy(pb_ctx *ctx, void *dev_records)
{
    if (!ctx || !dev_records || ctx->exp_parts == 0 || ctx->sk_owner_cap1 == 0) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    if (ctx->sk_owner_cur.size() != (size_t)ctx->exp_parts * PB_XCD) return PB_EINVAL;
    u64 at = 0;
    for (size_t r = 0; r < ctx->sk_owner_cur.size(); ++r) {                 // regions in (owner, XCD) order: an owner's records end up contiguous
        const u64 n = ctx->sk_owner_cur[r];
        if (n) HIPCHK(hipMemcpyAsync((char *)dev_records + at * 16, (const char *)ctx->part_bucket1.p + (u64)r * ctx->sk_owner_cap1 * 16, n * 16, hipMemcpyDeviceToDevice, ctx->stream));
        at += n;
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

// Stage 2: counts the n_records run records in dev_records (this rank's share of every rank's stage 1, any order): level-1 and level-2
// scatter on the low local_bits bucket bits (the owner bits above them are ignored), bucket kernel, histogram.  Outputs as pb_count's,
// for THIS rank's share of the key space; afterwards pb_retain / pb_export_* / pb_fetch_all work on it.
int pb_sk_count_records(pb_ctx *ctx, unsigned k, const void *dev_records, uint64_t n_records, unsigned local_bits, uint64_t *occ_hist, uint64_t *max_occ,
                        uint64_t *n_distinct)
{
    if (!ctx || k < 1 || k > 32 || (n_records && !dev_records) || local_bits < 2 || local_bits > 18) return PB_EINVAL;      // (both levels go through sk_part2: 9 + 9 bits)
    (void)hipSetDevice(ctx->dev);
    int rc;
    ctx->k = k;
    ctx->W = 1;
    ctx->n_retained = 0;
    ctx->n_distinct = 0;
    if ((rc = ensure(ctx, ctx->d_scalars, 16 * 8)) != PB_OK) return rc;
    SkParams p;
    memset(&p, 0, sizeof(p));
    p.k = k;
    p.m = k < SK_M ? k : SK_M;
    p.w = k - p.m + 1;
    const unsigned B = local_bits;
    p.B = B;
    p.l1 = B / 2;
    p.l2 = B - p.l1;
    p.nx = PB_XCD;
    const double est_rec = (double)std::max<u64>(n_records, 1);
    const u64 nb1 = (1ull << p.l1) * PB_XCD, nb2 = 1ull << B;
    p.cap1 = (u64)(est_rec / (double)nb1 * 1.5) + 4096;
    p.cap2 = (u32)std::min<double>(est_rec / (double)nb2 * 3.0 + 256.0, 1e9);
    // the level-1 sub-streams of this pass must not share storage with the incoming records: part_bucket1 may be the caller's source
    // (it is not: callers pass their own buffer), so a second buffer is used for them
    if ((rc = ensure(ctx, ctx->scratch[4], nb1 * p.cap1 * SkStore16::BYTES)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->part_bucket2, (nb2 * (u64)p.cap2 + 1024) * SkStore16::BYTES)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->part_meta, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4)) != PB_OK) return rc;
    p.spill_cap = std::max<u64>((u64)(est_rec / 16), 1ull << 16);
    if ((rc = ensure(ctx, ctx->part_spill, p.spill_cap * SkStore16::BYTES)) != PB_OK) return rc;
    p.spill = ctx->part_spill.p;
    p.bucket1 = ctx->scratch[4].p;
    p.bucket2 = ctx->part_bucket2.p;
    p.cursor1 = (u32 *)ctx->part_meta.p;
    p.cursor2 = p.cursor1 + nb1 * SK_CUR1_STRIDE;
    p.flag2 = p.cursor2 + nb2;
    const u64 out_cap = std::min<u64>(std::max<u64>(n_records * SK_RUN_MAXLEN, 1), nb2 * (u64)SK_SLOT_LIMIT);
    u64 fcap = 1ull << 20;
    for (int attempt = 0; attempt < 8; ++attempt) {
        if ((rc = ensure(ctx, ctx->dist_keys, (out_cap + fcap) * 8)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->dist_counts, (out_cap + fcap) * 4)) != PB_OK) return rc;
        if ((rc = ensure(ctx, ctx->table, fcap * 16)) != PB_OK) return rc;
        ctx->table_cap = fcap;
        p.fallback = TableView<1>{(u64 *)ctx->table.p, fcap - 1};
        p.scalars = (u64 *)ctx->d_scalars.p;
        if ((rc = ensure(ctx, ctx->d_hist, 65536 * 8)) != PB_OK) return rc;
        p.hist = (u64 *)ctx->d_hist.p;
        HIPCHK(hipMemsetAsync(ctx->d_hist.p, 0, 65536 * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->part_meta.p, 0, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4, ctx->stream));
        table_init_kernel<1><<<grid_for(fcap * 2, 256, 256 * 32), 256, 0, ctx->stream>>>(p.fallback.slots, fcap);
        StageTimer tm(ctx, "count_records");
        if (n_records) {
            const unsigned gf = (unsigned)std::min<u64>(std::max<u64>(n_records / SK2Q_RCAP / 2, 1), 256 * 12);
            sk_part2_raw16_kernel<false, true><<<dim3(gf, 1), SK_THREADS, 0, ctx->stream>>>(p, (const uint4 *)dev_records, n_records);
            const u64 exp_tiles = (u64)(est_rec / (double)nb1) / SK2Q_RCAP + 1;
            const unsigned blocks_per_sub = (unsigned)std::min<u64>(std::max<u64>(exp_tiles / 4, 1), 64);
            sk_part2_raw16_kernel<false><<<dim3(blocks_per_sub, (unsigned)nb1), SK_THREADS, 0, ctx->stream>>>(p);
            const unsigned g3 = grid_for(nb2, 1, 256 * 16);
            if (ctx->opt_bucket_form == 2) {
                if (k == 32) sk_bucket2_kernel<32, false><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, nullptr);
                else sk_bucket2_kernel<0, false><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, nullptr);
            }
            else if (k == 32) sk_bucket_kernel<32, false><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, nullptr);
            else sk_bucket_kernel<0, false><<<g3, SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, out_cap, nullptr);
        }
        HIPCHK(hipGetLastError());
        u64 sc[10];
        HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 80, hipMemcpyDeviceToHost, ctx->stream));
        if ((rc = tm.stop()) != PB_OK) return rc;
        if (sc[7] + sc[8]) {                                       // buckets that left the LDS path: every window of theirs through the global table
            sk_flagged_kernel<<<dim3(grid_for(nb2, 1, 1024), 16), 256, 0, ctx->stream>>>(p);
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 80, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
        }
        if (sc[1] == 0) {
            ctx->n_distinct = sc[4];
            ctx->hist_done = sc[4];
            ctx->h_hist_valid = false;
            ctx->sk_flagged = sc[7];
            ctx->sk_spilled = sc[8];
            ctx->stage_ms["sk_bucket_bits"] = (float)B;
            ctx->stage_ms["sk_flagged_buckets"] = (float)sc[7];
            ctx->stage_ms["sk_spilled_records"] = (float)sc[8];
            if (sc[7] + sc[8]) { if ((rc = table_to_arrays<1>(ctx, ctx->table_cap, true)) != PB_OK) return rc; }
            ctx->count_path = 2;
            if ((rc = finish_count(ctx, occ_hist, max_occ)) != PB_OK) return rc;
            if (n_distinct) *n_distinct = ctx->n_distinct;
            return PB_OK;
        }
        if (sc[8] > p.spill_cap) { ctx->err = "pb_sk_count_records: the spill list overflowed"; return PB_ERANGE; }
        fcap <<= 2;
    }
    ctx->err = "pb_sk_count_records: fallback table overflow";
    return PB_ENOMEM;
}

} // extern "C"

extern "C" {

int pb_count(pb_ctx *ctx, unsigned k, uint64_t *occ_hist, uint64_t *max_occ, uint64_t *n_instances, uint64_t *n_distinct)
{
    if (!ctx) return PB_EINVAL;
    if (k < 1 || k > PB_MAX_K) { ctx->err = "k out of range 1..160"; return PB_ERANGE; }
    (void)hipSetDevice(ctx->dev);
    ctx->k = k;
    ctx->W = (int)((k + 31) / 32);
    ctx->n_retained = 0;
    ctx->n_distinct = 0;
    int rc;
    if ((rc = ensure(ctx, ctx->d_scalars, 16 * 8)) != PB_OK) return rc;
    bool done = false;
    // small inputs take the direct pass unless pb_set_option("count_path") says otherwise
    const bool want_sk = k <= 96 && ctx->stream_len >= k &&
                         (ctx->opt_count_path ? ctx->opt_count_path == 2 : ctx->stream_len >= (1ull << 22));
    ctx->sk_records = 0;
    if (want_sk) {
        if ((rc = ctx->W == 1 ? count_superk(ctx, k, &done) : ctx->W == 2 ? count_superk2(ctx, k, &done) : count_superk3(ctx, k, &done)) != PB_OK) return rc;
    }
    if (!done) {
        switch (ctx->W) {
        case 1: rc = count_direct<1>(ctx, k); break;
        case 2: rc = count_direct<2>(ctx, k); break;
        case 3: rc = count_direct<3>(ctx, k); break;
        case 4: rc = count_direct<4>(ctx, k); break;
        default: rc = count_direct<5>(ctx, k); break;
        }
        if (rc != PB_OK) return rc;
    }
    ctx->count_path = done ? 2 : 1;
    if ((rc = finish_count(ctx, occ_hist, max_occ)) != PB_OK) return rc;
    if (n_instances) *n_instances = ctx->n_instances;
    if (n_distinct) *n_distinct = ctx->n_distinct;
    return PB_OK;
}

int pb_retain(pb_ctx *ctx, unsigned min_occ, uint64_t *n_keys)
{
    if (!ctx || ctx->W == 0) { if (ctx) ctx->err = "pb_retain before pb_count"; return PB_EINVAL; }
    (void)hipSetDevice(ctx->dev);
    int rc;
    const int W = ctx->W;
    DevBuf &ck = ctx->scratch[2], &cc = ctx->scratch[3];
    u64 n = 0;
    const u64 nd = ctx->n_distinct;
    if ((rc = ensure(ctx, ck, std::max<u64>(nd, 1) * W * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, cc, std::max<u64>(nd, 1) * 4)) != PB_OK) return rc;
    HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
    {
        StageTimer tm(ctx, "filter");
        if (nd)
            filter_array_kernel<<<grid_for(nd, 256 * FLT_RPT, 256 * 8), 256, 0, ctx->stream>>>((const u64 *)ctx->dist_keys.p, (const u32 *)ctx->dist_counts.p, nd, W,
                                                                                     min_occ, (u64 *)ck.p, (u32 *)cc.p, (u64 *)ctx->d_scalars.p);
        HIPCHK(hipGetLastError());
        if ((rc = tm.stop()) != PB_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(&n, ctx->d_scalars.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (n_keys) *n_keys = n;
    return sort_retained(ctx, n);
}

int pb_fetch_retained(pb_ctx *ctx, uint64_t *keys, uint16_t *counts)
{
    if (!ctx) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    if (ctx->n_retained == 0) return PB_OK;
    if (keys) HIPCHK(hipMemcpyAsync(keys, ctx->ret_keys.p, ctx->n_retained * ctx->W * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (counts) HIPCHK(hipMemcpyAsync(counts, ctx->ret_counts.p, ctx->n_retained * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

int pb_fetch_all(pb_ctx *ctx, uint64_t *keys, uint32_t *counts, uint64_t cap, uint64_t *n)
{
    if (!ctx || ctx->W == 0) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    const u64 m = ctx->n_distinct;
    if (n) *n = m;
    const u64 take = std::min<u64>(m, (u64)cap);
    if (take && keys) HIPCHK(hipMemcpyAsync(keys, ctx->dist_keys.p, take * ctx->W * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (take && counts) HIPCHK(hipMemcpyAsync(counts, ctx->dist_counts.p, take * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

int pb_last_kernel_ms(const pb_ctx *ctx, const char *stage, float *ms)
{
    if (!ctx || !stage || !ms) return PB_EINVAL;
    auto it = ctx->stage_ms.find(stage);
    if (it == ctx->stage_ms.end()) return PB_EINVAL;
    *ms = it->second;
    return PB_OK;
}

// ---- X1 ---------------------------------------------------------------------------------------------
int pb_export_begin(pb_ctx *ctx, int n_parts, uint64_t *part_counts)
{
    if (!ctx || !part_counts || n_parts < 1 || n_parts > PB_MAX_PARTS || ctx->W == 0) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    int rc;
    const int W = ctx->W;
    const u64 nd = ctx->n_distinct;
    DevBuf &pc = ctx->scratch[0];
    if ((rc = ensure(ctx, pc, PB_MAX_PARTS * 8 * 2)) != PB_OK) return rc;
    HIPCHK(hipMemsetAsync(pc.p, 0, PB_MAX_PARTS * 8 * 2, ctx->stream));
    StageTimer tm(ctx, "export");
    if (nd)
        part_count_kernel<<<grid_for(nd, 256, 256 * 8), 256, 0, ctx->stream>>>((const u64 *)ctx->dist_keys.p, (const u32 *)ctx->dist_counts.p, nd, W,
                                                                               (unsigned)n_parts, (u64 *)pc.p);
    u64 h[PB_MAX_PARTS];
    HIPCHK(hipMemcpyAsync(h, pc.p, n_parts * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    u64 cur[PB_MAX_PARTS], total = 0;
    for (int i = 0; i < n_parts; ++i) { cur[i] = total; total += h[i]; part_counts[i] = h[i]; }
    ctx->exp_total = total;
    ctx->exp_parts = (unsigned)n_parts;
    for (int i = 0; i < n_parts; ++i) ctx->exp_cur[i] = cur[i];
    HIPCHK(hipGetLastError());
    return tm.stop();
}

// Split export (see part_count2_kernel): per owner the number of count-1 records and of the others
int pb_export_split_begin(pb_ctx *ctx, int n_parts, uint64_t *single_counts, uint64_t *multi_counts)
{
    if (!ctx || !single_counts || !multi_counts || n_parts < 1 || n_parts > PB_MAX_PARTS || ctx->W == 0) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    int rc;
    const u64 nd = ctx->n_distinct;
    DevBuf &pc = ctx->scratch[0];
    // dest counts [128] | cursors [128] | key_start [128] | cnt_start [64]
    if ((rc = ensure(ctx, pc, (3 * 2 * PB_MAX_PARTS + PB_MAX_PARTS) * 8)) != PB_OK) return rc;
    HIPCHK(hipMemsetAsync(pc.p, 0, 2 * PB_MAX_PARTS * 8, ctx->stream));
    StageTimer tm(ctx, "export");
    if (nd)
        part_count2_kernel<<<grid_for(nd, 256, 256 * 8), 256, 0, ctx->stream>>>((const u64 *)ctx->dist_keys.p, (const u32 *)ctx->dist_counts.p, nd, ctx->W,
                                                                                (unsigned)n_parts, (u64 *)pc.p);
    u64 h[2 * PB_MAX_PARTS];
    HIPCHK(hipMemcpyAsync(h, pc.p, 2 * n_parts * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    u64 kpos = 0, cpos = 0;
    for (int o = 0; o < n_parts; ++o) {
        single_counts[o] = h[2 * o];
        multi_counts[o] = h[2 * o + 1];
        ctx->exp2_key_start[2 * o] = kpos;
        ctx->exp2_key_start[2 * o + 1] = kpos + h[2 * o];
        ctx->exp2_cnt_start[o] = cpos;
        kpos += h[2 * o] + h[2 * o + 1];
        cpos += h[2 * o + 1];
    }
    ctx->exp_total = kpos;
    ctx->exp2_parts = (unsigned)n_parts;
    HIPCHK(hipGetLastError());
    return tm.stop();
}

// dev_keys: room for every exported record (sum of both classes, W words each); dev_counts16: one u16 per record with count > 1
int pb_export_split_copy(pb_ctx *ctx, void *dev_keys, void *dev_counts16)
{
    if (!ctx || ctx->exp2_parts == 0) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    const u64 nd = ctx->n_distinct;
    if (ctx->exp_total && nd) {
        u64 *base = (u64 *)ctx->scratch[0].p;
        u64 *cursors = base + 2 * PB_MAX_PARTS, *key_start = cursors + 2 * PB_MAX_PARTS, *cnt_start = key_start + 2 * PB_MAX_PARTS;
        HIPCHK(hipMemcpyAsync(cursors, ctx->exp2_key_start, 2 * ctx->exp2_parts * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(key_start, ctx->exp2_key_start, 2 * ctx->exp2_parts * 8, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(cnt_start, ctx->exp2_cnt_start, ctx->exp2_parts * 8, hipMemcpyHostToDevice, ctx->stream));
        part_scatter2_kernel<<<grid_for(nd, 256 * PSC_RPT, 256 * 8), 256, 0, ctx->stream>>>((const u64 *)ctx->dist_keys.p, (const u32 *)ctx->dist_counts.p, nd, ctx->W,
                                                                                          ctx->exp2_parts, cursors, key_start, cnt_start, (u64 *)dev_keys,
                                                                                          (unsigned short *)dev_counts16);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

// the records are scattered straight into the caller's buffers (no staging copy)
int pb_export_copy(pb_ctx *ctx, void *dev_keys, void *dev_counts)
{
    if (!ctx || ctx->exp_parts == 0) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    const u64 nd = ctx->n_distinct;
    if (ctx->exp_total && nd) {
        u64 *cursors = (u64 *)ctx->scratch[0].p + PB_MAX_PARTS;
        HIPCHK(hipMemcpyAsync(cursors, ctx->exp_cur, ctx->exp_parts * 8, hipMemcpyHostToDevice, ctx->stream));
        part_scatter_kernel<<<grid_for(nd, 256 * PSC_RPT, 256 * 8), 256, 0, ctx->stream>>>((const u64 *)ctx->dist_keys.p, (const u32 *)ctx->dist_counts.p, nd, ctx->W,
                                                                                         ctx->exp_parts, cursors, (u64 *)dev_keys, (u32 *)dev_counts);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

} // extern "C"

template <int W>
static int merge_add_impl(pb_ctx *ctx, const void *dev_keys, const void *dev_counts, u64 n)
{
    int rc;
    TableView<W> tv{(u64 *)ctx->merge_table.p, ctx->merge_cap - 1};
    HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
    StageTimer tm(ctx, "merge");
    if (n) merge_kernel<W><<<grid_for(n, 256, 256 * 16), 256, 0, ctx->stream>>>((const u64 *)dev_keys, (const u32 *)dev_counts, n, tv, 1024u,
                                                                              (u64 *)ctx->d_scalars.p);
    HIPCHK(hipGetLastError());
    if ((rc = tm.stop()) != PB_OK) return rc;
    u64 sc[2];
    HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 16, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (sc[1]) { ctx->err = "merge table overflow (expected_records too small)"; return PB_ENOMEM; }
    ctx->merge_records += n;
    return PB_OK;
}

template <int W>
static int merge_begin_impl(pb_ctx *ctx, u64 cap)
{
    int rc;
    if ((rc = ensure(ctx, ctx->merge_table, cap * PB_SLOT_WORDS(W) * 8)) != PB_OK) return rc;
    ctx->merge_cap = cap;
    ctx->merge_records = 0;
    table_init_kernel<W><<<grid_for(cap * PB_SLOT_WORDS(W), 256, 256 * 32), 256, 0, ctx->stream>>>((u64 *)ctx->merge_table.p, cap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

#define PB_DISPATCH_W(ctx, fn, ...)                                   \
    ((ctx)->W == 1 ? fn<1>(__VA_ARGS__) : (ctx)->W == 2 ? fn<2>(__VA_ARGS__) : (ctx)->W == 3 ? fn<3>(__VA_ARGS__) \
     : (ctx)->W == 4 ? fn<4>(__VA_ARGS__) : fn<5>(__VA_ARGS__))

// owner-side merge through the partition + LDS machinery (k <= 32); *done = false: the caller uses the table path
// G = SkStoreM12: 12-byte records (counts below 2^(32-B)); SkStore16 otherwise
template <typename G>
static int merge_records_lds_impl(pb_ctx *ctx, const void *dev_keys, const void *dev_counts, u64 n, unsigned B, bool *done)
{
    int rc;
    *done = false;
    SkParams p;
    p.B = B;
    p.l1 = B / 2;
    p.l2 = B - p.l1;
    p.nx = PB_XCD;
    const u64 nb1 = (1ull << p.l1) * PB_XCD, nb2 = 1ull << B;
    p.cap1 = n / nb1 * 3 / 2 + 4096;
    p.cap2 = (u32)(n / nb2 * 2 + 256);
    if ((rc = ensure(ctx, ctx->part_bucket1, nb1 * p.cap1 * G::BYTES)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->part_bucket2, nb2 * (u64)p.cap2 * G::BYTES)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->part_meta, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4)) != PB_OK) return rc;
    p.spill_cap = 1ull << 16;
    if ((rc = ensure(ctx, ctx->part_spill, p.spill_cap * sizeof(SkRec))) != PB_OK) return rc;
    p.spill = ctx->part_spill.p;
    p.bucket1 = ctx->part_bucket1.p;
    p.bucket2 = ctx->part_bucket2.p;
    p.cursor1 = (u32 *)ctx->part_meta.p;
    p.cursor2 = p.cursor1 + nb1 * SK_CUR1_STRIDE;
    p.flag2 = p.cursor2 + nb2;
    p.fallback = TableView<1>{nullptr, 0};
    if ((rc = ensure(ctx, ctx->d_hist, 65536 * 8)) != PB_OK) return rc;
    p.hist = (u64 *)ctx->d_hist.p;
    HIPCHK(hipMemsetAsync(ctx->d_hist.p, 0, 65536 * 8, ctx->stream));
    p.scalars = (u64 *)ctx->d_scalars.p;
    if ((rc = ensure(ctx, ctx->dist_keys, n * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->dist_counts, n * 4)) != PB_OK) return rc;
    HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
    HIPCHK(hipMemsetAsync(ctx->part_meta.p, 0, (nb1 * SK_CUR1_STRIDE + 2 * nb2) * 4, ctx->stream));
    StageTimer tm(ctx, "merge");
    mg_part1_kernel<G><<<grid_for((n + SK2_RCAP - 1) / SK2_RCAP, 1, 256 * 8), SK_THREADS, 0, ctx->stream>>>((const u64 *)dev_keys, (const u32 *)dev_counts, n, p);
    const u64 exp_tiles = n / nb1 / SK2_RCAP + 1;
    const dim3 g2((unsigned)std::min<u64>(std::max<u64>(exp_tiles / 4, 1), 64), (unsigned)nb1);
    if (G::BYTES == 12) sk_part2_raw12_kernel<SkStoreM12><<<g2, SK_THREADS, 0, ctx->stream>>>(p);
    else sk_part2_kernel<SkStore16><<<g2, SK_THREADS, 0, ctx->stream>>>(p);
    mg_bucket_kernel<G><<<grid_for(nb2, 1, 256 * 16), SK_ROLL_THREADS, 0, ctx->stream>>>(p, (u64 *)ctx->dist_keys.p, (u32 *)ctx->dist_counts.p, n);
    HIPCHK(hipGetLastError());
    u64 sc[9];
    HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 72, hipMemcpyDeviceToHost, ctx->stream));
    if ((rc = tm.stop()) != PB_OK) return rc;
    if (sc[1] || sc[7] || sc[8]) return PB_OK;              // a region or a table overflowed (one key with a huge share): the table path takes it
    ctx->n_distinct = sc[4];
    ctx->hist_done = sc[4];                  // mg_bucket_kernel tallied every record it wrote
    ctx->n_retained = 0;
    *done = true;
    return PB_OK;
}

static int merge_records_lds(pb_ctx *ctx, const void *dev_keys, const void *dev_counts, u64 n, bool *done)
{
    *done = false;
    if (n < (1ull << 20)) return PB_OK;
    unsigned B = 8;
    while (((u64)1536 << B) < n && B < 20) ++B;            // hash-uniform buckets: mean <= 1536 distinct keys, the table takes 3840
    if (((u64)1536 << B) < n) return PB_OK;
    int rc;
    if (B <= 16 && !ctx->opt_merge_16b) {              // records of the exchange carry 16-bit counts: 12 bytes per record do
        if ((rc = merge_records_lds_impl<SkStoreM12>(ctx, dev_keys, dev_counts, n, B, done)) != PB_OK || *done) return rc;
    }
    return merge_records_lds_impl<SkStore16>(ctx, dev_keys, dev_counts, n, B, done);
}

extern "C" {

int pb_merge_records(pb_ctx *ctx, unsigned k, const void *dev_keys, const void *dev_counts, uint64_t n, uint64_t *occ_hist, uint64_t *max_occ,
                     uint64_t *n_distinct)
{
    if (!ctx || k < 1 || k > PB_MAX_K || (n && (!dev_keys || !dev_counts))) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    int rc;
    ctx->k = k;
    ctx->W = (int)((k + 31) / 32);
    if ((rc = ensure(ctx, ctx->d_scalars, 16 * 8)) != PB_OK) return rc;
    bool done = false;
    if (ctx->W == 1 && !ctx->opt_merge_table) {
        if ((rc = merge_records_lds(ctx, dev_keys, dev_counts, n, &done)) != PB_OK) return rc;
    }
    if (done) {
        if ((rc = finish_count(ctx, occ_hist, max_occ)) != PB_OK) return rc;
        if (n_distinct) *n_distinct = ctx->n_distinct;
        return PB_OK;
    }
    if ((rc = pb_merge_begin(ctx, k, std::max<u64>(n, 1))) != PB_OK) return rc;
    if ((rc = pb_merge_add(ctx, dev_keys, dev_counts, n)) != PB_OK) return rc;
    return pb_merge_end(ctx, occ_hist, max_occ, n_distinct);
}

} // extern "C"

extern "C" {

int pb_merge_begin(pb_ctx *ctx, unsigned k, uint64_t expected_records)
{
    if (!ctx || k < 1 || k > PB_MAX_K) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    int rc;
    ctx->k = k;
    ctx->W = (int)((k + 31) / 32);
    if ((rc = ensure(ctx, ctx->d_scalars, 16 * 8)) != PB_OK) return rc;
    u64 cap = 1ull << 16;
    while (cap < expected_records * 2) cap <<= 1;
    return PB_DISPATCH_W(ctx, merge_begin_impl, ctx, cap);
}

int pb_merge_add(pb_ctx *ctx, const void *dev_keys, const void *dev_counts, uint64_t n)
{
    if (!ctx || ctx->merge_cap == 0 || (n && (!dev_keys || !dev_counts))) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    return PB_DISPATCH_W(ctx, merge_add_impl, ctx, dev_keys, dev_counts, (u64)n);
}

int pb_merge_end(pb_ctx *ctx, uint64_t *occ_hist, uint64_t *max_occ, uint64_t *n_distinct)
{
    if (!ctx || ctx->merge_cap == 0) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    std::swap(ctx->table, ctx->merge_table);
    ctx->table_cap = ctx->merge_cap;
    ctx->merge_cap = 0;
    ctx->n_retained = 0;
    ctx->n_distinct = 0;
    int rc = PB_DISPATCH_W(ctx, table_to_arrays, ctx, std::min<u64>(ctx->merge_records, ctx->table_cap), false);
    if (rc != PB_OK) return rc;
    if ((rc = finish_count(ctx, occ_hist, max_occ)) != PB_OK) return rc;
    if (n_distinct) *n_distinct = ctx->n_distinct;
    return PB_OK;
}

} // extern "C"

#include "graph_multik.inc"
#include "chains.inc"
#include "kmer_occ_layout.inc"
#include "node_kernels.inc"
#include "dist.inc"
#include "raw_parse.inc"
#include "gap_graphs.inc"
