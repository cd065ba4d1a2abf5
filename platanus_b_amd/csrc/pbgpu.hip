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

// ------------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct pb_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
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
    DevBuf d_hist;                      // u64[65536]
    DevBuf d_scalars;                   // u64[16] scratch counters

    // retained (sorted) set
    DevBuf ret_keys, ret_counts;        // u64[n*W], u16[n]
    u64 n_retained = 0;

    // distinct (key,count) arrays of the last pb_count / pb_merge_end (device form of kmerFP)
    DevBuf dist_keys, dist_counts;      // u64[n*W], u32[n]
    int count_path = 0;                 // 1 = direct (v1), 2 = partitioned (v2)
    DevBuf part_bucket1, part_bucket2, part_meta, part_out_counts, part_segs;

    // owner exchange (X1)
    DevBuf exp_keys, exp_counts, merge_table;
    u64 exp_total = 0, merge_cap = 0, merge_records = 0;
    unsigned exp_parts = 0;

    // graph stage / multi-k rounds
    DevBuf index_table;                 // retained key -> index + 1
    DevBuf nmask_cur, alive, d_read_start;   // filtered read set: working mask copy, per-read flag, stream starts
    std::vector<u64> read_start;        // host copy: stream position of every read + final sentinel (= stream_len)
    size_t read_start_dev_n = 0;
    bool mask_filtered = false;
    DevBuf contig_packed, contig_mask, ctable;   // contig-derived next-k table (saveContig)
    u64 ctable_cap = 0;
    unsigned ctable_k = 0;
    DevBuf chain_work, chain_rec, chain_seq;     // K7 chain compaction
    DevBuf layout_work, layout_slots, layout_out, scan_sums, slot_blk;   // CT8 exact PREFIX_kmer_occ.bin layout
    u64 layout_records = 0;
    u64 n_chains = 0, chain_seq_words = 0;
    void *chain_ptrs[2] = {nullptr, nullptr};

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

static inline unsigned grid_for(u64 n, unsigned block, unsigned max_blocks = 256 * 16)
{
    u64 g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------------------------
// K1: pack reads.  One thread per 64 stream positions (two packed words + one mask word), so every
// output word has exactly one writer.  Input bases are read byte-wise; neighbouring lanes read
// neighbouring 64-byte segments.
// ------------------------------------------------------------------------------------------------------
__constant__ unsigned char c_char2bin[16] = {46, 0, 46, 1, 3, 46, 46, 2, 46, 46, 46, 46, 46, 46, 4, 46};   // common.h:256

__global__ void pack_reads_kernel(const unsigned char *__restrict__ bases, const u64 *__restrict__ off, u64 n_new_reads,
                                  u64 old_len, u64 new_len, u64 *__restrict__ packed, u64 *__restrict__ nmask,
                                  u64 first_word64, u64 n_word64, int ascii)
{
    const u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_word64) return;
    const u64 w64 = first_word64 + t;
    const u64 p0 = w64 * 64;
    u64 hi = 0, lo = 0, m = 0;
    u64 keep_hi = 0, keep_lo = 0, keep_m = 0;
    if (p0 < old_len) { keep_hi = packed[w64 * 2]; keep_lo = packed[w64 * 2 + 1]; keep_m = nmask[w64]; }
    // locate the read that holds the first new position of this word: largest r with start(r) <= p,
    // start(r) = old_len + off[r] + r  (one separator per read)
    u64 p = p0 < old_len ? old_len : p0;
    u64 r = 0;
    if (p < new_len) {
        u64 lo_r = 0, hi_r = n_new_reads;      // invariant: start(lo_r) <= p < start(hi_r)
        while (hi_r - lo_r > 1) {
            const u64 mid = (lo_r + hi_r) >> 1;
            if (old_len + off[mid] + mid <= p) lo_r = mid; else hi_r = mid;
        }
        r = lo_r;
    }
    u64 r_start = old_len + off[r] + r;
    u64 r_len = off[r + 1] - off[r];
    for (unsigned j = 0; j < 64; ++j) {
        const u64 q = p0 + j;
        unsigned code;        // 0..3, or 4 = unknown
        if (q < old_len) continue;
        if (q >= new_len) code = 4;
        else {
            while (q > r_start + r_len) { ++r; r_start = old_len + off[r] + r; r_len = off[r + 1] - off[r]; }
            if (q == r_start + r_len) code = 4;                                   // separator
            else {
                unsigned c = bases[off[r] + (q - r_start)];
                if (ascii) c = c_char2bin[c & 0xF];
                code = c > 3 ? 4 : c;
            }
        }
        if (code > 3) m |= 1ull << j;
        else if (j < 32) hi |= (u64)code << (62 - 2 * j);
        else lo |= (u64)code << (62 - 2 * (j - 32));
    }
    if (p0 < old_len) {
        const unsigned nk = (unsigned)(old_len - p0);      // positions [0, nk) are kept (1..63)
        const u64 km = (1ull << nk) - 1;
        m = (keep_m & km) | (m & ~km);
        if (nk >= 32) {
            hi = keep_hi;
            const unsigned n2 = nk - 32;
            const u64 kml = n2 ? (~0ull << (64 - 2 * n2)) : 0;
            lo = (keep_lo & kml) | (lo & ~kml);
        } else {
            const u64 kmh = ~0ull << (64 - 2 * nk);
            hi = (keep_hi & kmh) | (hi & ~kmh);
        }
    }
    packed[w64 * 2] = hi;
    packed[w64 * 2 + 1] = lo;
    nmask[w64] = m;
}

// ------------------------------------------------------------------------------------------------------
// K2+K3: one lane per stream position.  The k-mer is cut out of the 2-bit stream (no rolling state, so
// every lane is independent), canonicalised and added to the occurrence table.
// ------------------------------------------------------------------------------------------------------
template <int W>
__global__ void __launch_bounds__(256)
count_kernel(const u64 *__restrict__ packed, const u64 *__restrict__ nmask, u64 stream_len, unsigned k,
             TableView<W> table, u32 max_probe, u64 *__restrict__ scalars /* [0]=instances [1]=overflow */)
{
    u64 local = 0;
    bool ovf = false;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g + k <= stream_len; g += stride) {
        if (stream_window_has_n(nmask, g, k)) continue;
        const DKey<W> key = key_canon<W>(stream_kmer<W>(packed, g, k), k);
        if (!table_add<W>(table, key, 1u, max_probe)) ovf = true;
        ++local;
    }
    // wave reduction of the instance count (__shfl over 64 lanes), one atomic per wave
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&scalars[0], local);
    if (ovf) scalars[1] = 1;
}

template <int W>
__global__ void table_init_kernel(u64 *__restrict__ slots, u64 cap)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 n = cap * PB_SLOT_WORDS(W);
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (W == 1) slots[i] = (i & 1) ? 0ull : PB_EMPTY_KEY;
        else slots[i] = 0ull;
    }
}

// ------------------------------------------------------------------------------------------------------
// Result representation.  Whatever counting path ran, pb_count leaves the distinct k-mers as two device
// arrays: dist_keys u64[n*W] and dist_counts u32[n] (unsorted).  That is the device form of the reference's
// kmerFP temp file (counter.h:482-507); histogram, cutoff filter, sort and owner export all read it.
// ------------------------------------------------------------------------------------------------------

// table -> arrays.  One atomic per wave: lanes are ranked with __ballot + popcount.
template <int W>
__global__ void __launch_bounds__(256)
table_to_arrays_kernel(TableView<W> table, u64 *__restrict__ out_keys, u32 *__restrict__ out_counts, u64 cap_out,
                       u64 *__restrict__ counter)
{
    const u64 cap = table.mask + 1;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const unsigned lane = threadIdx.x & 63;
    const u64 n_iter = (cap + stride - 1) / stride;
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 it = 0; it < n_iter; ++it, s += stride) {
        bool take = false;
        u32 c = 0;
        const u64 *slot = nullptr;
        if (s < cap) {
            slot = table.slots + s * PB_SLOT_WORDS(W);
            bool occ;
            if (W == 1) { occ = slot[0] != PB_EMPTY_KEY; c = (u32)slot[1]; }
            else { const u64 meta = slot[W]; occ = (meta >> 32) != 0; c = (u32)meta; }
            take = occ && c != 0;
        }
        const u64 ballot = __ballot(take);
        if (ballot == 0) continue;
        u64 base = 0;
        const int leader = __ffsll((long long)ballot) - 1;
        if ((int)lane == leader) base = atomicAdd(counter, (u64)__popcll(ballot));
        base = __shfl(base, leader, 64);
        if (take) {
            const u64 pos = base + __popcll(ballot & ((1ull << lane) - 1));
            if (pos < cap_out) {
#pragma unroll
                for (int i = 0; i < W; ++i) out_keys[pos * W + i] = slot[i];
                out_counts[pos] = c;
            }
        }
    }
}

// K4a: occurrenceDistribution (counter.h:482-507) from the count array.  Bins < 4096 live in LDS (16 KB),
// the rare larger counts go straight to global memory.
#define HIST_LDS_BINS 4096
__global__ void __launch_bounds__(256)
hist_array_kernel(const u32 *__restrict__ counts, u64 n, u64 *__restrict__ hist)
{
    __shared__ u32 lh[HIST_LDS_BINS];
    for (unsigned i = threadIdx.x; i < HIST_LDS_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        u32 c = counts[i];
        if (c == 0) continue;
        if (c > PB_COUNT_MAX) c = PB_COUNT_MAX;
        if (c < HIST_LDS_BINS) atomicAdd(&lh[c], 1u);
        else atomicAdd(&hist[c], 1ull);
    }
    __syncthreads();
    for (unsigned i = threadIdx.x; i < HIST_LDS_BINS; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], (u64)lh[i]);
}

// K4b: keep records with min_occ <= min(count, 65534) (sortedKeyFromKmerFile's filter, counter.h:927-934)
__global__ void __launch_bounds__(256)
filter_array_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ counts, u64 n, int W, u32 min_occ,
                    u64 *__restrict__ out_keys, u32 *__restrict__ out_counts, u64 *__restrict__ counter)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const unsigned lane = threadIdx.x & 63;
    const u64 n_iter = (n + stride - 1) / stride;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 it = 0; it < n_iter; ++it, i += stride) {
        u32 c = i < n ? counts[i] : 0;
        if (c > PB_COUNT_MAX) c = PB_COUNT_MAX;
        const bool take = c != 0 && c >= min_occ;
        const u64 ballot = __ballot(take);
        if (ballot == 0) continue;
        u64 base = 0;
        const int leader = __ffsll((long long)ballot) - 1;
        if ((int)lane == leader) base = atomicAdd(counter, (u64)__popcll(ballot));
        base = __shfl(base, leader, 64);
        if (take) {
            const u64 pos = base + __popcll(ballot & ((1ull << lane) - 1));
            for (int w = 0; w < W; ++w) out_keys[pos * W + w] = keys[i * W + w];
            out_counts[pos] = c;
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// X1 (no reference counterpart: the reference's threads share one table).  Multi-GPU merge of per-GPU
// occurrence tables = owner exchange: every (key,count) record goes to the GPU that owns the key's
// stripe, owner = ((key & 1023) * n_parts) >> 10 -- the reference's own stripe function (counter.h:461).
//   part_count_kernel   : records per owner (LDS counters, one global add per owner per block)
//   part_scatter_kernel : bucketed copy; a block reserves one range per owner, lanes rank inside via LDS
//   merge_kernel        : owner side, table_add(key, count) -- counts are commutative and saturate at
//                         read-out (counter.h:468-470), so the union of owner shards is the -t 1 table
// ------------------------------------------------------------------------------------------------------
#define PB_MAX_PARTS 64
__global__ void __launch_bounds__(256)
part_count_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ counts, u64 n, int W, unsigned n_parts,
                  u64 *__restrict__ part_counts)
{
    __shared__ u32 lc[PB_MAX_PARTS];
    if (threadIdx.x < PB_MAX_PARTS) lc[threadIdx.x] = 0;
    __syncthreads();
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (counts[i] == 0) continue;
        atomicAdd(&lc[(unsigned)(((keys[i * W] & 1023) * n_parts) >> 10)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < n_parts && lc[threadIdx.x]) atomicAdd(&part_counts[threadIdx.x], (u64)lc[threadIdx.x]);
}

__global__ void __launch_bounds__(256)
part_scatter_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ counts, u64 n, int W, unsigned n_parts,
                    u64 *__restrict__ cursors /* start offsets, advanced */, u64 *__restrict__ out_keys,
                    u32 *__restrict__ out_counts)
{
    __shared__ u32 lc[PB_MAX_PARTS];
    __shared__ u64 lbase[PB_MAX_PARTS];
    const u64 stride = (u64)gridDim.x * blockDim.x;
    const u64 n_iter = (n + stride - 1) / stride;
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 it = 0; it < n_iter; ++it, i += stride) {
        if (threadIdx.x < PB_MAX_PARTS) lc[threadIdx.x] = 0;
        __syncthreads();
        const u32 c = i < n ? counts[i] : 0;
        const bool take = c != 0;
        unsigned owner = 0;
        u32 local = 0;
        if (take) { owner = (unsigned)(((keys[i * W] & 1023) * n_parts) >> 10); local = atomicAdd(&lc[owner], 1u); }
        __syncthreads();
        if (threadIdx.x < n_parts && lc[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (u64)lc[threadIdx.x]);
        __syncthreads();
        if (take) {
            const u64 pos = lbase[owner] + local;
            for (int w = 0; w < W; ++w) out_keys[pos * W + w] = keys[i * W + w];
            out_counts[pos] = c > PB_COUNT_MAX ? PB_COUNT_MAX : c;
        }
        __syncthreads();
    }
}

template <int W>
__global__ void __launch_bounds__(256)
merge_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ counts, u64 n, TableView<W> table, u32 max_probe,
             u64 *__restrict__ scalars)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    bool ovf = false;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        DKey<W> key;
#pragma unroll
        for (int w = 0; w < W; ++w) key.w[w] = keys[i * W + w];
        if (!table_add<W>(table, key, counts[i], max_probe)) ovf = true;
    }
    if (ovf) scalars[1] = 1;
}

// ------------------------------------------------------------------------------------------------------
// K2+K3, partitioned form (k <= 32).  The v1 kernel above does two random device-scope atomics per window on
// a multi-GB table and is bound by the atomic units (measured 18 ms for cfg2).  Here the random access is
// moved into LDS:
//   The partition buffers hold m = mix64(canonical key), a bijection, so the bucket index is simply the top l1+l2 bits
//   of m, the key is recovered with unmix64, and a bucket's LDS slot needs only the remaining bits.
//   part1_kernel  extract canonical k-mers from the packed stream and scatter them by hash digit 1 into
//                 2^l1 level-1 buckets.  A block stages a tile of keys in LDS sorted by digit (LDS counting
//                 sort), reserves one range per digit with a single global atomic, and copies every run with
//                 coalesced stores.
//   part2_kernel  same on each level-1 bucket with hash digit 2 -> 2^(l1+l2) final buckets of <= PB_BUCKET_CAP
//                 keys (mean <= 1536: hash partitioning makes sizes Poisson, capacity = mean + 13 sigma).
//   bucket_count_kernel  one block per final bucket: open-addressing occurrence hashing in LDS (64-bit LDS CAS on
//                 the key claims a slot, 32-bit LDS add counts), then the (key,count) records of the bucket are
//                 appended to the distinct arrays with one global atomic per block.
// A bucket that overflows its capacity (one k-mer with an enormous count, or a level-1 overflow) is flagged;
// everything of a flagged bucket is counted through the v1 global table instead (fallback, always correct).
// ------------------------------------------------------------------------------------------------------
#define PB_TILE 8192                 /* keys staged per block iteration: 64 KB of LDS                    */
#define PB_PART_THREADS 512          /* two 512-lane blocks per CU (79 KB of LDS each): their phases overlap          */
#define PB_TILE_ITEMS (PB_TILE / PB_PART_THREADS)
#define PB_MAX_DIGITS 1024
#define PB_BUCKET_CAP 2048           /* keys per final bucket                                             */
#define PB_BUCKET_SLOTS 4096         /* LDS table slots per final bucket: load <= 0.5 even if all distinct */

struct PartParams {
    unsigned l1, l2;                 // digit widths: level-1 buckets = 2^l1, final buckets = 2^(l1+l2)
    u64 cap1;                        // capacity (keys) of a level-1 sub-stream (bucket x XCD)
    u64 *bucket1;                    // [2^l1 * PB_XCD * cap1]
    u32 *cursor1;                    // [2^l1 * PB_XCD] keys appended so far (may exceed cap1: the excess went to the fallback)
    u64 *bucket2;                    // [2^(l1+l2) * PB_BUCKET_CAP]
    u32 *cursor2;                    // [2^(l1+l2)]
    u32 *flag2;                      // [2^(l1+l2)] 1 = some key of this final bucket went to the fallback table
    TableView<1> fallback;
    u64 *scalars;                    // [0] instances [1] fallback overflow
};

__device__ __forceinline__ u64 part_hash(u64 mixed_key) { return mixed_key; }     // buffers hold mix64(key): digits are its top bits

// Scatter `nvalid` keys held in registers (kreg[j], valid bit j of vmask) into 2^bits destinations.
// digit(key) = (part_hash(key) >> shift) & (2^bits - 1).  dst_of(d) gives base pointer, capacity and cursor.
template <typename DST>
__device__ __forceinline__ void tile_scatter(const u64 (&kreg)[PB_TILE_ITEMS], u32 vmask, unsigned shift, unsigned bits, DST dst,
                                             u64 *skeys, u32 *shist, u32 *soff, u32 *sbase)
{
    const unsigned nd = 1u << bits;
    for (unsigned i = threadIdx.x; i < nd; i += PB_PART_THREADS) shist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PB_TILE_ITEMS; ++j)
        if (vmask & (1u << j)) atomicAdd(&shist[(unsigned)(kreg[j] >> shift) & (nd - 1)], 1u);
    __syncthreads();
    // exclusive scan of shist[0..nd) -> soff (nd <= 1024: two bins per lane, wave scan + wave totals)
    {
        __shared__ u32 wtot[PB_PART_THREADS / 64];
        const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const unsigned i0 = threadIdx.x * 2;
        const u32 a = i0 < nd ? shist[i0] : 0, b = i0 + 1 < nd ? shist[i0 + 1] : 0;
        u32 x = a + b;
        for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o, 64); if ((int)lane >= o) x += y; }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();
        u32 run = x - (a + b);
        for (unsigned w = 0; w < wave; ++w) run += wtot[w];
        if (i0 < nd) { soff[i0] = run; sbase[i0] = run; }
        if (i0 + 1 < nd) { soff[i0 + 1] = run + a; sbase[i0 + 1] = run + a; }
    }
    __syncthreads();
    // place keys (sbase doubles as the running LDS cursor)
#pragma unroll
    for (int j = 0; j < PB_TILE_ITEMS; ++j)
        if (vmask & (1u << j)) skeys[atomicAdd(&sbase[(unsigned)(kreg[j] >> shift) & (nd - 1)], 1u)] = kreg[j];
    __syncthreads();
    // reserve one global range per non-empty digit
    for (unsigned d = threadIdx.x; d < nd; d += PB_PART_THREADS) {
        const u32 cnt = shist[d];
        sbase[d] = cnt ? dst.reserve(d, cnt) : 0;
    }
    __syncthreads();
    const u32 total = soff[nd - 1] + shist[nd - 1];
    for (u32 i = threadIdx.x; i < total; i += PB_PART_THREADS) {
        const u64 key = skeys[i];
        const unsigned d = (unsigned)(key >> shift) & (nd - 1);
        dst.store(d, sbase[d] + (i - soff[d]), key);
    }
    __syncthreads();
}

// Level-1 buckets are split into PB_XCD sub-streams, one per XCD: a block appends to the sub-stream of the XCD it
// runs on (HW_REG_XCC_ID; speed only -- any value is correct).  Appends to one sub-stream then all come through the
// same L2, whose write-back merges the ~100-byte runs of different blocks into full lines before they reach HBM;
// with one shared stream per bucket the eight non-coherent L2s each write partial lines of the same bucket tail
// (measured: part1 6.1 ms for 3.2 GB of keys against part2's 1.9 ms for the same bytes).
#define PB_XCD 8
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & (PB_XCD - 1); }   // HW_REG_XCC_ID[3:0]

struct Dst1 {
    PartParams p;
    unsigned x;
    __device__ __forceinline__ u32 reserve(unsigned d, u32 cnt) const { return atomicAdd(&p.cursor1[d * PB_XCD + x], cnt); }
    __device__ __forceinline__ void store(unsigned d, u32 pos, u64 key) const
    {
        if (pos < p.cap1) p.bucket1[((u64)d * PB_XCD + x) * p.cap1 + pos] = key;
        else {   // level-1 overflow: the key is counted in the fallback table and its final bucket is flagged
            p.flag2[(unsigned)(key >> (64 - p.l1 - p.l2))] = 1;
            DKey<1> k1; k1.w[0] = unmix64(key);
            if (!table_add<1>(p.fallback, k1, 1u, 1024u)) p.scalars[1] = 1;
        }
    }
};
struct Dst2 {
    PartParams p;
    unsigned b1;
    __device__ __forceinline__ u32 reserve(unsigned d, u32 cnt) const { return atomicAdd(&p.cursor2[(b1 << p.l2) + d], cnt); }
    __device__ __forceinline__ void store(unsigned d, u32 pos, u64 key) const
    {
        const unsigned b = (b1 << p.l2) + d;
        if (pos < PB_BUCKET_CAP) p.bucket2[(u64)b * PB_BUCKET_CAP + pos] = key;
        else {
            p.flag2[b] = 1;
            DKey<1> k1; k1.w[0] = unmix64(key);
            if (!table_add<1>(p.fallback, k1, 1u, 1024u)) p.scalars[1] = 1;
        }
    }
};

// k <= 32 window test on LDS-staged mask words: bits [g, g+k) of the mask stream
__device__ __forceinline__ bool lds_window_has_n(const u64 *__restrict__ sm, u32 rel, unsigned k)
{
    const u32 wi = rel >> 6, sh = rel & 63;
    const u64 lo = sm[wi] >> sh;
    const u64 hi = sh ? (sm[wi + 1] << (64 - sh)) : 0;
    const u64 m = k == 64 ? ~0ull : ((1ull << k) - 1);
    return ((lo | hi) & m) != 0;
}
__device__ __forceinline__ u64 lds_kmer(const u64 *__restrict__ sp, u32 rel, unsigned k)
{
    const u32 bit = 2 * rel, wi = bit >> 6, sh = bit & 63;
    const u64 hi = sp[wi];
    const u64 v = sh ? ((hi << sh) | (sp[wi + 1] >> (64 - sh))) : hi;
    return v >> (64 - 2 * k);
}

#define PB_TILE_PWORDS (PB_TILE / 32 + 2)      /* packed words covering a tile + one k-mer of overhang */
#define PB_TILE_MWORDS (PB_TILE / 64 + 3)

__global__ void __launch_bounds__(PB_PART_THREADS, 4)
part1_kernel(const u64 *__restrict__ packed, const u64 *__restrict__ nmask, u64 stream_len, u64 n_words64, unsigned k, PartParams p)
{
    __shared__ u64 skeys[PB_TILE];
    __shared__ u32 shist[PB_MAX_DIGITS], soff[PB_MAX_DIGITS], sbase[PB_MAX_DIGITS];
    __shared__ u64 spack[PB_TILE_PWORDS], smask[PB_TILE_MWORDS];
    const u64 n_tiles = (stream_len + PB_TILE - 1) / PB_TILE;
    const unsigned xcd = xcc_id();
    u64 local = 0;
    for (u64 tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const u64 g0 = tile * PB_TILE;                       // multiple of 64: word aligned in both streams
        // words past the allocation (last tile) read as all-unknown
        for (unsigned i = threadIdx.x; i < PB_TILE_PWORDS; i += PB_PART_THREADS) spack[i] = g0 / 32 + i < 2 * n_words64 ? packed[g0 / 32 + i] : 0ull;
        for (unsigned i = threadIdx.x; i < PB_TILE_MWORDS; i += PB_PART_THREADS) smask[i] = g0 / 64 + i < n_words64 ? nmask[g0 / 64 + i] : ~0ull;
        __syncthreads();
        // lane t owns PB_TILE_ITEMS consecutive positions: the first k-mer is cut out of the staged words, the others are
        // rolled (forward <<= 2 | base, reverse >>= 2 | complement: counter.h:418-422) -- ~6 ALU ops instead of ~80 each
        u64 kreg[PB_TILE_ITEMS];
        u32 vmask = 0;
        {
            const u32 rel0 = threadIdx.x * PB_TILE_ITEMS;
            const u64 kmask = k == 32 ? ~0ull : ((1ull << (2 * k)) - 1);
            const u64 wmask = k == 64 ? ~0ull : ((1ull << k) - 1);
            // unknown-base bits of positions [rel0, rel0 + 64)
            const u32 mwi = rel0 >> 6, msh = rel0 & 63;
            const u64 mwin = (smask[mwi] >> msh) | (msh ? (smask[mwi + 1] << (64 - msh)) : 0);
            u64 fwd = lds_kmer(spack, rel0, k);
            DKey<1> f0; f0.w[0] = fwd;
            u64 rev = key_revcomp<1>(f0, k).w[0];
#pragma unroll
            for (int j = 0; j < PB_TILE_ITEMS; ++j) {
                if (j > 0) {
                    const u32 pos = rel0 + j + k - 1;                        // new 3' base
                    const u64 c = (spack[pos >> 5] >> (62 - 2 * (pos & 31))) & 3;
                    fwd = ((fwd << 2) | c) & kmask;
                    rev = (rev >> 2) | ((3 - c) << (2 * (k - 1)));
                }
                kreg[j] = 0;
                if (((mwin >> j) & wmask) == 0) {
                    kreg[j] = mix64(fwd < rev ? fwd : rev);
                    vmask |= 1u << j;
                }
            }
        }
        local += __popc(vmask);
        tile_scatter(kreg, vmask, 64 - p.l1, p.l1, Dst1{p, xcd}, skeys, shist, soff, sbase);
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&p.scalars[0], local);
}

// grid = (tiles per level-1 sub-stream, 2^l1 * PB_XCD)
__global__ void __launch_bounds__(PB_PART_THREADS, 4)
part2_kernel(PartParams p)
{
    __shared__ u64 skeys[PB_TILE];
    __shared__ u32 shist[PB_MAX_DIGITS], soff[PB_MAX_DIGITS], sbase[PB_MAX_DIGITS];
    const unsigned sub = blockIdx.y;             // level-1 sub-stream = bucket * PB_XCD + xcd
    const unsigned b1 = sub / PB_XCD;
    u64 n = p.cursor1[sub];
    if (n > p.cap1) n = p.cap1;
    const u64 *src = p.bucket1 + (u64)sub * p.cap1;
    for (u64 tile = blockIdx.x; tile * PB_TILE < n; tile += gridDim.x) {
        u64 kreg[PB_TILE_ITEMS];
        u32 vmask = 0;
#pragma unroll
        for (int j = 0; j < PB_TILE_ITEMS; ++j) {
            const u64 i = tile * PB_TILE + (u64)j * PB_PART_THREADS + threadIdx.x;
            kreg[j] = 0;
            if (i < n) { kreg[j] = src[i]; vmask |= 1u << j; }
        }
        tile_scatter(kreg, vmask, 64 - p.l1 - p.l2, p.l2, Dst2{p, b1}, skeys, shist, soff, sbase);
    }
}

// exclusive scan of min(in[i], clamp) over n <= 2^20 entries (n a multiple of 1024) into u64 out[]; total -> *total
__global__ void __launch_bounds__(256)
scan_local_kernel(const u32 *__restrict__ in, u32 clamp, u64 *__restrict__ out, u64 *__restrict__ block_sums)
{
    __shared__ u32 wtot[4];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 i0 = (u64)blockIdx.x * 1024 + threadIdx.x * 4;
    u32 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const u32 c = in[i0 + q]; v[q] = c > clamp ? clamp : c; }
    const u32 sum4 = v[0] + v[1] + v[2] + v[3];
    u32 x = sum4;
    for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o, 64); if ((int)lane >= o) x += y; }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    u32 run = x - sum4;
    for (unsigned w = 0; w < wave; ++w) run += wtot[w];
#pragma unroll
    for (int q = 0; q < 4; ++q) { out[i0 + q] = run; run += v[q]; }
    if (threadIdx.x == 255) block_sums[blockIdx.x] = run;
}
__global__ void __launch_bounds__(1024)
scan_sums_kernel(u64 *__restrict__ block_sums, unsigned n_blocks, u64 *__restrict__ total)
{
    __shared__ u64 wtot[16];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 v = threadIdx.x < n_blocks ? block_sums[threadIdx.x] : 0;
    u64 x = v;
    for (int o = 1; o < 64; o <<= 1) { const u64 y = __shfl_up(x, o, 64); if ((int)lane >= o) x += y; }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    u64 run = x - v;
    for (unsigned w = 0; w < wave; ++w) run += wtot[w];
    if (threadIdx.x < n_blocks) block_sums[threadIdx.x] = run;
    if (threadIdx.x == n_blocks - 1) *total = run + v;
}
__global__ void __launch_bounds__(256)
scan_add_kernel(u64 *__restrict__ out, const u64 *__restrict__ block_sums)
{
    const u64 i0 = (u64)blockIdx.x * 1024 + threadIdx.x * 4;
    const u64 add = block_sums[blockIdx.x];
#pragma unroll
    for (int q = 0; q < 4; ++q) out[i0 + q] += add;
}

// One block per final bucket: LDS-resident open-addressing occurrence hashing.  The bucket's records are written at
// bucket_base[b] (exclusive prefix of the bucket sizes, an upper bound of the records before it), so no global
// atomic is on the block's critical path; n_occ[b] records are valid, a later pass closes the gaps.
__global__ void __launch_bounds__(256, 4)
bucket_count_kernel(PartParams p, u64 *__restrict__ out_keys, u32 *__restrict__ out_counts, const u64 *__restrict__ bucket_base,
                    u32 *__restrict__ n_occ_out)
{
    // one u64 per slot: (m << logp) | count.  The top logp bits of m are the bucket index, so they are implied; the count
    // of a k-mer inside a bucket is at most PB_BUCKET_CAP = 2048 < 2^12 <= 2^logp.  Insert = one LDS CAS (new key) or one
    // LDS 64-bit add (seen key); 32 KB of LDS per bucket -> 5 resident blocks per CU.
    __shared__ u64 lslot[PB_BUCKET_SLOTS];
    __shared__ u32 wtot[4];
    const unsigned logp = p.l1 + p.l2;
    const unsigned n_buckets = 1u << logp;
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int KPL = PB_BUCKET_CAP / 256;         // keys per lane
    constexpr int SPL = PB_BUCKET_SLOTS / 256;       // table slots per lane (contiguous: lane t owns slots [t*SPL, (t+1)*SPL))

    // software pipeline: the keys of the block's next bucket are loaded while the current one is emitted.  All KPL loads
    // are issued unconditionally (bucket regions have a fixed size), so they do not wait for the bucket's size to arrive.
    u64 kin[KPL];
    u32 n = 0, flagged = 0;
    u64 next_base = 0;
    auto fetch = [&](unsigned b) {
        const unsigned bb = b < n_buckets ? b : n_buckets - 1;
        const u64 *src = p.bucket2 + (u64)bb * PB_BUCKET_CAP;
#pragma unroll
        for (int j = 0; j < KPL; ++j) kin[j] = src[(u32)j * 256 + threadIdx.x];
        n = b < n_buckets ? p.cursor2[bb] : 0;
        flagged = p.flag2[bb];
        next_base = bucket_base[bb];
    };
    fetch(blockIdx.x);
    for (unsigned b = blockIdx.x; b < n_buckets; b += gridDim.x) {
        u32 cur_n = n;
        if (cur_n > PB_BUCKET_CAP) cur_n = PB_BUCKET_CAP;
        const bool cur_flagged = flagged != 0;
        const u64 gbase = next_base;
        if (cur_n == 0) { fetch(b + gridDim.x); continue; }          // n_occ_out[b] stays 0 (memset)
        if (cur_flagged) {           // flagged: count through the global fallback table
#pragma unroll
            for (int j = 0; j < KPL; ++j)
                if ((u32)j * 256 + threadIdx.x < cur_n) {
                    DKey<1> k1; k1.w[0] = unmix64(kin[j]);
                    if (!table_add<1>(p.fallback, k1, 1u, 1024u)) p.scalars[1] = 1;
                }
            fetch(b + gridDim.x);
            continue;
        }
        for (unsigned i = threadIdx.x; i < PB_BUCKET_SLOTS; i += 256) lslot[i] = ~0ull;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < KPL; ++j) {
            if ((u32)j * 256 + threadIdx.x >= cur_n) continue;
            const u64 want = kin[j] << logp;                         // key bits without the implied bucket index, count = 0
            u32 s = (u32)(kin[j] >> 3) & (PB_BUCKET_SLOTS - 1);      // low bits of m: independent of the bucket digits
            while (true) {
                u64 v = lslot[s];
                if (v == ~0ull) {
                    v = atomicCAS(&lslot[s], ~0ull, want | 1ull);
                    if (v == ~0ull) break;
                }
                if ((v >> logp) == (want >> logp)) { atomicAdd(&lslot[s], 1ull); break; }
                s = (s + 1) & (PB_BUCKET_SLOTS - 1);
            }
        }
        fetch(b + gridDim.x);                         // next bucket's loads fly during the emit below
        __syncthreads();
        // emit: every lane counts its occupied slots; a block-wide exclusive scan gives its offset in a dense LDS list
        // (the table is ~10 % full: the dense list makes the unmix + store work proportional to the records, not the slots)
        u32 mine = 0;
        u64 sl[SPL];
#pragma unroll
        for (int j = 0; j < SPL; ++j) { sl[j] = lslot[threadIdx.x * SPL + j]; mine += sl[j] != ~0ull; }
        u32 x = mine;
        for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o, 64); if ((int)lane >= o) x += y; }
        if (lane == 63) wtot[wave] = x;
        __syncthreads();                              // also: every lane has read its slots, the table may be overwritten
        u32 off = x - mine;
        for (unsigned w = 0; w < wave; ++w) off += wtot[w];
        const u32 total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
#pragma unroll
        for (int j = 0; j < SPL; ++j)
            if (sl[j] != ~0ull) lslot[off++] = sl[j];          // dense list reuses the table's storage (off <= own first slot)
        __syncthreads();
        const u64 top = (u64)b << (64 - logp);
        for (u32 i = threadIdx.x; i < total; i += 256) {
            const u64 v = lslot[i];
            out_keys[gbase + i] = unmix64((v >> logp) | top);
            out_counts[gbase + i] = (u32)(v & ((1ull << logp) - 1));
        }
        if (threadIdx.x == 0) n_occ_out[b] = total;
        __syncthreads();
    }
}

// close the gaps: bucket b's n_occ[b] records move from src[bucket_base[b]] to dst[dst_base[b]]; 16 lanes per bucket
__global__ void __launch_bounds__(256)
bucket_compact_kernel(const u64 *__restrict__ src_keys, const u32 *__restrict__ src_counts, const u64 *__restrict__ bucket_base,
                      const u32 *__restrict__ n_occ, const u64 *__restrict__ dst_base, unsigned n_buckets,
                      u64 *__restrict__ dst_keys, u32 *__restrict__ dst_counts)
{
    const unsigned sub = threadIdx.x & 15;
    for (u64 b = ((u64)blockIdx.x * blockDim.x + threadIdx.x) >> 4; b < n_buckets; b += ((u64)gridDim.x * blockDim.x) >> 4) {
        const u32 n = n_occ[b];
        const u64 sb = bucket_base[b], db = dst_base[b];
        for (u32 i = sub; i < n; i += 16) {
            dst_keys[db + i] = src_keys[sb + i];
            dst_counts[db + i] = src_counts[sb + i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// K5: stable LSD radix sort of (u64 key, u32 value) pairs, 8 bits per pass.
//   rs_count   : per-tile digit histogram              -> counts[digit * n_tiles + tile]
//   rs_scan    : exclusive scan of counts (one block)
//   rs_scatter : stable rank inside the tile via __ballot digit matching per wave, then scatter
// Tile = 256 lanes x RS_ITEMS; each wave owns a contiguous quarter of the tile so that tile order =
// (wave, round, lane) order and the scatter is stable.
// ------------------------------------------------------------------------------------------------------
#define RS_ITEMS 8
#define RS_TILE (256 * RS_ITEMS)

__global__ void __launch_bounds__(256)
rs_count_kernel(const u64 *__restrict__ keys, u64 n, unsigned shift, u32 *__restrict__ counts, u32 n_tiles)
{
    __shared__ u32 h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * RS_TILE;
    for (int j = 0; j < RS_ITEMS; ++j) {
        const u64 i = base + (u64)j * 256 + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255], 1u);
    }
    __syncthreads();
    counts[(u64)threadIdx.x * n_tiles + blockIdx.x] = h[threadIdx.x];
}

__global__ void __launch_bounds__(1024)
rs_scan_kernel(u32 *__restrict__ counts, u64 n)
{
    __shared__ u32 wsum[16];
    __shared__ u32 carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (u64 base = 0; base < n; base += 1024) {
        const u64 i = base + threadIdx.x;
        const u32 v = i < n ? counts[i] : 0;
        u32 x = v;
        for (int o = 1; o < 64; o <<= 1) { const u32 y = __shfl_up(x, o, 64); if ((int)lane >= o) x += y; }
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        u32 woff = 0;
        for (unsigned w = 0; w < wave; ++w) woff += wsum[w];
        const u32 c = carry;
        if (i < n) counts[i] = c + woff + x - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = c + woff + x;
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
rs_scatter_kernel(const u64 *__restrict__ keys_in, const u32 *__restrict__ vals_in, u64 *__restrict__ keys_out,
                  u32 *__restrict__ vals_out, u64 n, unsigned shift, const u32 *__restrict__ offsets, u32 n_tiles)
{
    __shared__ u32 wcnt[4][256];
    __shared__ u32 dbase[4][256];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (unsigned i = threadIdx.x; i < 1024; i += 256) (&wcnt[0][0])[i] = 0;
    __syncthreads();
    const u64 wbase = (u64)blockIdx.x * RS_TILE + (u64)wave * (RS_TILE / 4);
    u64 kreg[RS_ITEMS];
    u32 vreg[RS_ITEMS], rank[RS_ITEMS];
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const u64 i = wbase + (u64)j * 64 + lane;
        const bool valid = i < n;
        kreg[j] = valid ? keys_in[i] : 0;
        vreg[j] = valid ? vals_in[i] : 0;
        const unsigned d = (unsigned)(kreg[j] >> shift) & 255;
        u64 peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1;
            const u64 m = __ballot(bit);
            peers &= bit ? m : ~m;
        }
        const u32 prior = wcnt[wave][d];
        rank[j] = prior + (u32)__popcll(peers & ((1ull << lane) - 1));
        // all lanes of the wave have read `prior` before the leader's update (same wave, program order)
        if (valid && (int)lane == __ffsll((long long)peers) - 1) wcnt[wave][d] = prior + (u32)__popcll(peers);
    }
    __syncthreads();
    {
        const unsigned d = threadIdx.x;
        u32 run = offsets[(u64)d * n_tiles + blockIdx.x];
        for (int w = 0; w < 4; ++w) { dbase[w][d] = run; run += wcnt[w][d]; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RS_ITEMS; ++j) {
        const u64 i = wbase + (u64)j * 64 + lane;
        if (i < n) {
            const unsigned d = (unsigned)(kreg[j] >> shift) & 255;
            const u64 pos = (u64)dbase[wave][d] + rank[j];
            keys_out[pos] = kreg[j];
            vals_out[pos] = vreg[j];
        }
    }
}

__global__ void iota_kernel(u32 *__restrict__ v, u64 n)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) v[i] = (u32)i;
}
// sortkey[i] = keys[perm[i] * W + word]
__global__ void gather_word_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ perm, u64 n, int W, int word,
                                   u64 *__restrict__ out)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = keys[(u64)perm[i] * W + word];
}
// out_keys[i] = keys[perm[i]], out_counts[i] = (u16)counts[perm[i]]
__global__ void apply_perm_kernel(const u64 *__restrict__ keys, const u32 *__restrict__ counts, const u32 *__restrict__ perm,
                                  u64 n, int W, u64 *__restrict__ out_keys, u16 *__restrict__ out_counts)
{
    const u64 stride = (u64)gridDim.x * blockDim.x;
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const u64 p = perm[i];
        for (int w = 0; w < W; ++w) out_keys[i * W + w] = keys[p * W + w];
        out_counts[i] = (u16)counts[p];
    }
}

// sort pairs (a_keys, a_vals) of length n by the low `nbits` bits of the key; result ends in (a_keys, a_vals)
static int radix_sort_pairs(pb_ctx *ctx, u64 *a_keys, u32 *a_vals, u64 *b_keys, u32 *b_vals, u64 n, unsigned nbits,
                            u32 *counts, u32 n_tiles)
{
    u64 *ik = a_keys, *ok = b_keys;
    u32 *iv = a_vals, *ov = b_vals;
    unsigned passes = (nbits + 7) / 8;
    if (passes & 1) ++passes;           // an even number of passes leaves the result in (a_keys, a_vals)
    for (unsigned p = 0; p < passes; ++p) {
        const unsigned shift = p * 8;
        if (shift >= 64) {              // padding pass beyond the key width: plain copy
            HIPCHK(hipMemcpyAsync(ok, ik, n * 8, hipMemcpyDeviceToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(ov, iv, n * 4, hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            rs_count_kernel<<<n_tiles, 256, 0, ctx->stream>>>(ik, n, shift, counts, n_tiles);
            rs_scan_kernel<<<1, 1024, 0, ctx->stream>>>(counts, (u64)256 * n_tiles);
            rs_scatter_kernel<<<n_tiles, 256, 0, ctx->stream>>>(ik, iv, ok, ov, n, shift, counts, n_tiles);
        }
        std::swap(ik, ok);
        std::swap(iv, ov);
    }
    HIPCHK(hipGetLastError());
    return PB_OK;
}

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

void pb_destroy(pb_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->dev);
    (void)hipStreamSynchronize(ctx->stream);
    release(ctx->packed); release(ctx->nmask); release(ctx->table); release(ctx->d_hist); release(ctx->d_scalars);
    release(ctx->ret_keys); release(ctx->ret_counts); release(ctx->exp_keys); release(ctx->exp_counts); release(ctx->merge_table);
    release(ctx->dist_keys); release(ctx->dist_counts); release(ctx->part_bucket1); release(ctx->part_bucket2); release(ctx->part_meta); release(ctx->part_out_counts); release(ctx->part_segs);
    release(ctx->index_table); release(ctx->nmask_cur); release(ctx->alive); release(ctx->d_read_start);
    release(ctx->contig_packed); release(ctx->contig_mask); release(ctx->ctable);
    release(ctx->chain_work); release(ctx->chain_rec); release(ctx->chain_seq);
    release(ctx->layout_work); release(ctx->layout_slots); release(ctx->layout_out); release(ctx->scan_sums); release(ctx->slot_blk);
    for (auto &s : ctx->scratch) release(s);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *pb_last_error(const pb_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

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


int pb_reads_append(pb_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads, int ascii)
{
    if (!ctx || !offsets || (n_reads && !bases && offsets[n_reads] > 0)) return PB_EINVAL;
    if (n_reads == 0) return PB_OK;
    (void)hipSetDevice(ctx->dev);
    const u64 n_new_bases = offsets[n_reads] - offsets[0];
    for (u64 r = 0; r < n_reads; ++r) {
        const u64 len = offsets[r + 1] - offsets[r];
        if (len >= 500000) { ctx->err = "read of 500000 bases or more (common.cpp:28)"; return PB_ERANGE; }
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
    // staging copy of the host buffers
    DevBuf &d_bases = ctx->scratch[0], &d_off = ctx->scratch[1];
    if ((rc = ensure(ctx, d_bases, n_new_bases + 16)) != PB_OK) return rc;
    if ((rc = ensure(ctx, d_off, (n_reads + 1) * 8)) != PB_OK) return rc;
    std::vector<u64> rel(n_reads + 1);
    for (u64 r = 0; r <= n_reads; ++r) rel[r] = offsets[r] - offsets[0];
    HIPCHK(hipMemcpyAsync(d_bases.p, bases + offsets[0], n_new_bases, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(hipMemcpyAsync(d_off.p, rel.data(), (n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    StageTimer tm(ctx, "pack");
    const u64 first64 = old_len / 64;
    const u64 n64 = words64 - first64;          // includes the all-unknown padding words
    pack_reads_kernel<<<(unsigned)((n64 + 255) / 256), 256, 0, ctx->stream>>>(
        (const unsigned char *)d_bases.p, (const u64 *)d_off.p, n_reads, old_len, new_len, (u64 *)ctx->packed.p,
        (u64 *)ctx->nmask.p, first64, n64, ascii);
    HIPCHK(hipGetLastError());
    if ((rc = tm.stop()) != PB_OK) return rc;
    if (!ctx->read_start.empty()) ctx->read_start.pop_back();
    for (u64 r = 0; r < n_reads; ++r) ctx->read_start.push_back(old_len + rel[r] + r);
    ctx->read_start.push_back(new_len);
    ctx->read_start_dev_n = 0;
    ctx->mask_filtered = false;
    ctx->stream_len = new_len;
    ctx->n_reads += n_reads;
    ctx->n_bases += n_new_bases;
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
    HIPCHK(hipMemsetAsync(ctx->d_hist.p, 0, 65536 * 8, ctx->stream));
    {
        StageTimer tm(ctx, "hist");
        if (ctx->n_distinct)
            hist_array_kernel<<<grid_for(ctx->n_distinct, 256, 256 * 8), 256, 0, ctx->stream>>>((const u32 *)ctx->dist_counts.p, ctx->n_distinct,
                                                                                               (u64 *)ctx->d_hist.p);
        HIPCHK(hipGetLastError());
        if ((rc = tm.stop()) != PB_OK) return rc;
    }
    std::vector<u64> h(65536);
    HIPCHK(hipMemcpyAsync(h.data(), ctx->d_hist.p, 65536 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (occ_hist) for (int i = 0; i < PB_OCC_BINS; ++i) occ_hist[i] = h[i];
    if (max_occ) { *max_occ = 0; for (int i = PB_COUNT_MAX; i > 0; --i) if (h[i]) { *max_occ = (u64)i; break; } }
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
    table_to_arrays_kernel<W><<<grid_for(ctx->table_cap, 256, 256 * 8), 256, 0, ctx->stream>>>(
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
    const char *gc = getenv("PB_GRIDCAP");
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
            count_kernel<W><<<grid_for(ctx->stream_len, 256, gc ? atoi(gc) : 256 * 32), 256, 0, ctx->stream>>>(
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

// v2: two-level hash partitioning + LDS-resident counting (k <= 32)
static int count_partitioned(pb_ctx *ctx, unsigned k, bool *done)
{
    int rc;
    *done = false;
    const u64 windows = ctx->stream_len;          // upper bound on counted windows
    unsigned logp = 12;               // >= 12: a bucket's LDS slot keeps the count in the low logp bits (count <= 2048)
    while (((u64)1536 << logp) < windows && logp < 20) ++logp;
    if (((u64)1536 << logp) < windows) return PB_OK;          // too large for 2^20 buckets: caller uses v1
    PartParams p;
    p.l1 = (logp + 1) / 2;
    p.l2 = logp - p.l1;
    // a sub-stream holds 1/PB_XCD of a bucket if blocks are dealt evenly over the XCDs (observed: round-robin); 1.5x margin,
    // anything beyond goes through the fallback table
    p.cap1 = ((windows >> p.l1) / PB_XCD) * 3 / 2 + 8192;
    const u64 nb1 = (1ull << p.l1) * PB_XCD, nb2 = 1ull << logp;
    if ((rc = ensure(ctx, ctx->part_bucket1, nb1 * p.cap1 * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->part_bucket2, nb2 * PB_BUCKET_CAP * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->part_meta, (nb1 + 2 * nb2) * 4)) != PB_OK) return rc;
    p.bucket1 = (u64 *)ctx->part_bucket1.p;
    p.bucket2 = (u64 *)ctx->part_bucket2.p;
    p.cursor1 = (u32 *)ctx->part_meta.p;
    p.cursor2 = p.cursor1 + nb1;
    p.flag2 = p.cursor2 + nb2;
    // fallback table (v1 representation) for flagged buckets
    u64 fcap = 1ull << 20;
    for (int attempt = 0; attempt < 10; ++attempt) {
        if ((rc = ensure(ctx, ctx->table, fcap * 16)) != PB_OK) return rc;
        ctx->table_cap = fcap;
        p.fallback = TableView<1>{(u64 *)ctx->table.p, fcap - 1};
        p.scalars = (u64 *)ctx->d_scalars.p;
        HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
        HIPCHK(hipMemsetAsync(ctx->part_meta.p, 0, (nb1 + 2 * nb2) * 4, ctx->stream));
        {
            StageTimer ti(ctx, "table_init");
            table_init_kernel<1><<<grid_for(fcap * 2, 256, 256 * 32), 256, 0, ctx->stream>>>(p.fallback.slots, fcap);
            if ((rc = ti.stop()) != PB_OK) return rc;
        }
        const u64 cap_out = std::max<u64>(windows, 1);
        DevBuf &tmp_keys = ctx->part_bucket1;      // level-1 buckets are dead after part2: reuse them as the segmented output
        DevBuf &tmp_counts = ctx->part_out_counts;
        if ((rc = ensure(ctx, tmp_counts, cap_out * 4)) != PB_OK) return rc;
        // per-bucket offsets: [bucket_base u64 nb2][dst_base u64 nb2][block sums u64 1024][n_occ u32 nb2]
        if ((rc = ensure(ctx, ctx->part_segs, nb2 * 20 + 1024 * 8 + 64)) != PB_OK) return rc;
        u64 *bucket_base = (u64 *)ctx->part_segs.p, *dst_base = bucket_base + nb2, *blk_sums = dst_base + nb2;
        u32 *n_occ = (u32 *)(blk_sums + 1024);
        HIPCHK(hipMemsetAsync(n_occ, 0, nb2 * 4, ctx->stream));
        {
            StageTimer t1(ctx, "part1");
            const u64 n_tiles = (ctx->stream_len + PB_TILE - 1) / PB_TILE;
            part1_kernel<<<grid_for(n_tiles, 1, 256 * 8), PB_PART_THREADS, 0, ctx->stream>>>((const u64 *)ctx->packed.p, cur_mask(ctx),
                                                                                ctx->stream_len, (ctx->stream_len + 63) / 64 + PB_STREAM_PAD_WORDS64, k, p);
            HIPCHK(hipGetLastError());
            if ((rc = t1.stop()) != PB_OK) return rc;
        }
        {
            StageTimer t2(ctx, "part2");
            const unsigned tiles_per_b1 = (unsigned)std::min<u64>((p.cap1 + PB_TILE - 1) / PB_TILE, 65535);
            part2_kernel<<<dim3(tiles_per_b1, (unsigned)nb1), PB_PART_THREADS, 0, ctx->stream>>>(p);   // nb1 counts sub-streams
            HIPCHK(hipGetLastError());
            if ((rc = t2.stop()) != PB_OK) return rc;
        }
        const unsigned scan_blocks = (unsigned)(nb2 / 1024);
        {
            StageTimer t3(ctx, "bucket");
            scan_local_kernel<<<scan_blocks, 256, 0, ctx->stream>>>(p.cursor2, PB_BUCKET_CAP, bucket_base, blk_sums);
            scan_sums_kernel<<<1, 1024, 0, ctx->stream>>>(blk_sums, scan_blocks, (u64 *)ctx->d_scalars.p + 3);
            scan_add_kernel<<<scan_blocks, 256, 0, ctx->stream>>>(bucket_base, blk_sums);
            bucket_count_kernel<<<grid_for(nb2, 1, 256 * 64), 256, 0, ctx->stream>>>(p, (u64 *)tmp_keys.p, (u32 *)tmp_counts.p, bucket_base, n_occ);
            HIPCHK(hipGetLastError());
            if ((rc = t3.stop()) != PB_OK) return rc;
        }
        u64 sc[5];
        {
            StageTimer t4(ctx, "seg_compact");
            scan_local_kernel<<<scan_blocks, 256, 0, ctx->stream>>>(n_occ, 0xFFFFFFFFu, dst_base, blk_sums);
            scan_sums_kernel<<<1, 1024, 0, ctx->stream>>>(blk_sums, scan_blocks, (u64 *)ctx->d_scalars.p + 4);
            scan_add_kernel<<<scan_blocks, 256, 0, ctx->stream>>>(dst_base, blk_sums);
            HIPCHK(hipMemcpyAsync(sc, ctx->d_scalars.p, 40, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(hipStreamSynchronize(ctx->stream));
            if (sc[1] == 0) {
                const u64 total = sc[4];
                if ((rc = ensure(ctx, ctx->dist_keys, std::max<u64>(total + fcap, 1) * 8)) != PB_OK) return rc;
                if ((rc = ensure(ctx, ctx->dist_counts, std::max<u64>(total + fcap, 1) * 4)) != PB_OK) return rc;
                bucket_compact_kernel<<<grid_for(nb2 * 16, 256, 256 * 16), 256, 0, ctx->stream>>>(
                    (const u64 *)tmp_keys.p, (const u32 *)tmp_counts.p, bucket_base, n_occ, dst_base, (unsigned)nb2, (u64 *)ctx->dist_keys.p,
                    (u32 *)ctx->dist_counts.p);
                HIPCHK(hipGetLastError());
            }
            if ((rc = t4.stop()) != PB_OK) return rc;
        }
        if (sc[1] == 0) {
            ctx->n_instances = sc[0];
            ctx->n_distinct = sc[4];
            break;
        }
        fcap <<= 2;                     // the fallback table itself overflowed: retry with a larger one
        if (attempt == 9) { ctx->err = "fallback table overflow"; return PB_ENOMEM; }
    }
    ctx->stage_ms["count"] = ctx->stage_ms["part1"] + ctx->stage_ms["part2"] + ctx->stage_ms["bucket"] + ctx->stage_ms["seg_compact"];
    // records counted through the fallback table are appended
    if ((rc = table_to_arrays<1>(ctx, ctx->table_cap, true)) != PB_OK) return rc;
    *done = true;
    return PB_OK;
}

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
    const char *force = getenv("PB_COUNT_PATH");       // "direct" | "partitioned" (development / A-B runs)
    const bool want_part = ctx->W == 1 && ctx->stream_len >= k &&
                           (force ? strcmp(force, "partitioned") == 0 : ctx->stream_len >= (1ull << 22));
    if (want_part && !(force && strcmp(force, "direct") == 0)) {
        if ((rc = count_partitioned(ctx, k, &done)) != PB_OK) return rc;
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
            filter_array_kernel<<<grid_for(nd, 256, 256 * 8), 256, 0, ctx->stream>>>((const u64 *)ctx->dist_keys.p, (const u32 *)ctx->dist_counts.p, nd, W,
                                                                                     min_occ, (u64 *)ck.p, (u32 *)cc.p, (u64 *)ctx->d_scalars.p);
        HIPCHK(hipGetLastError());
        if ((rc = tm.stop()) != PB_OK) return rc;
    }
    HIPCHK(hipMemcpyAsync(&n, ctx->d_scalars.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->n_retained = n;
    if ((rc = ensure(ctx, ctx->ret_keys, std::max<u64>(n, 1) * W * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->ret_counts, std::max<u64>(n, 1) * 2)) != PB_OK) return rc;
    if (n_keys) *n_keys = n;
    if (n == 0) return PB_OK;
    if (n >= (1ull << 32)) { ctx->err = "more than 2^32 retained k-mers on one GPU"; return PB_ERANGE; }
    // LSD radix sort of a permutation, one key word at a time (least significant word first)
    const u32 n_tiles = (u32)((n + RS_TILE - 1) / RS_TILE);
    DevBuf &sk_a = ctx->scratch[4], &sk_b = ctx->scratch[5], &pv = ctx->scratch[0], &cn = ctx->scratch[1];
    if ((rc = ensure(ctx, sk_a, n * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, sk_b, n * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, pv, n * 4 * 2)) != PB_OK) return rc;
    if ((rc = ensure(ctx, cn, (u64)256 * n_tiles * 4)) != PB_OK) return rc;
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
    if ((rc = tm.stop()) != PB_OK) return rc;
    return PB_OK;
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
    if ((rc = ensure(ctx, ctx->exp_keys, std::max<u64>(total, 1) * W * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->exp_counts, std::max<u64>(total, 1) * 4)) != PB_OK) return rc;
    u64 *cursors = (u64 *)pc.p + PB_MAX_PARTS;
    HIPCHK(hipMemcpyAsync(cursors, cur, n_parts * 8, hipMemcpyHostToDevice, ctx->stream));
    if (nd)
        part_scatter_kernel<<<grid_for(nd, 256, 256 * 8), 256, 0, ctx->stream>>>((const u64 *)ctx->dist_keys.p, (const u32 *)ctx->dist_counts.p, nd, W,
                                                                                 (unsigned)n_parts, cursors, (u64 *)ctx->exp_keys.p, (u32 *)ctx->exp_counts.p);
    HIPCHK(hipGetLastError());
    return tm.stop();
}

int pb_export_copy(pb_ctx *ctx, void *dev_keys, void *dev_counts)
{
    if (!ctx || ctx->exp_parts == 0) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    if (ctx->exp_total) {
        HIPCHK(hipMemcpyAsync(dev_keys, ctx->exp_keys.p, ctx->exp_total * ctx->W * 8, hipMemcpyDeviceToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(dev_counts, ctx->exp_counts.p, ctx->exp_total * 4, hipMemcpyDeviceToDevice, ctx->stream));
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

#include "pb_graph.inc"
