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

    // owner exchange (X1)
    DevBuf exp_keys, exp_counts, merge_table;
    u64 exp_total = 0, merge_cap = 0;
    unsigned exp_parts = 0;

    // scratch
    DevBuf scratch[6];
};

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
// K4a: histogram of saturated counts.  Bins < 4096 live in LDS (16 KB), the rare larger counts go
// straight to global memory.  hist[65535] is used as the "distinct keys" counter (bin 65535 is never a
// count: saturation is at 65534, counter.h:468-470).
// ------------------------------------------------------------------------------------------------------
#define HIST_LDS_BINS 4096
template <int W>
__global__ void __launch_bounds__(256)
hist_kernel(TableView<W> table, u64 *__restrict__ hist)
{
    __shared__ u32 lh[HIST_LDS_BINS];
    for (unsigned i = threadIdx.x; i < HIST_LDS_BINS; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    const u64 cap = table.mask + 1;
    const u64 stride = (u64)gridDim.x * blockDim.x;
    u64 distinct = 0;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += stride) {
        const u64 *slot = table.slots + s * PB_SLOT_WORDS(W);
        bool occ;
        u32 c;
        if (W == 1) { occ = slot[0] != PB_EMPTY_KEY; c = (u32)slot[1]; }
        else { const u64 meta = slot[W]; occ = (meta >> 32) != 0; c = (u32)meta; }
        if (!occ || c == 0) continue;
        if (c > PB_COUNT_MAX) c = PB_COUNT_MAX;
        ++distinct;
        if (c < HIST_LDS_BINS) atomicAdd(&lh[c], 1u);
        else atomicAdd(&hist[c], 1ull);
    }
    for (int o = 32; o > 0; o >>= 1) distinct += __shfl_down(distinct, o, 64);
    if ((threadIdx.x & 63) == 0 && distinct) atomicAdd(&hist[65535], distinct);
    __syncthreads();
    for (unsigned i = threadIdx.x; i < HIST_LDS_BINS; i += blockDim.x)
        if (lh[i]) atomicAdd(&hist[i], (u64)lh[i]);
}

// K4b: stream compaction of slots with min_occ <= count.  Output order is arbitrary (the sort follows).
// One atomic per wave: lanes are ranked with __ballot + popcount.
template <int W>
__global__ void __launch_bounds__(256)
compact_kernel(TableView<W> table, u32 min_occ, u64 *__restrict__ out_keys, u32 *__restrict__ out_counts,
               u64 cap_out, u64 *__restrict__ counter)
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
            if (c > PB_COUNT_MAX) c = PB_COUNT_MAX;
            take = occ && c != 0 && c >= min_occ;
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
template <int W>
__device__ __forceinline__ bool slot_record(const TableView<W> &t, u64 s, u32 &c, const u64 *&slot)
{
    slot = t.slots + s * PB_SLOT_WORDS(W);
    bool occ;
    if (W == 1) { occ = slot[0] != PB_EMPTY_KEY; c = (u32)slot[1]; }
    else { const u64 meta = slot[W]; occ = (meta >> 32) != 0; c = (u32)meta; }
    return occ && c != 0;
}

template <int W>
__global__ void __launch_bounds__(256)
part_count_kernel(TableView<W> table, unsigned n_parts, u64 *__restrict__ part_counts)
{
    __shared__ u32 lc[PB_MAX_PARTS];
    if (threadIdx.x < PB_MAX_PARTS) lc[threadIdx.x] = 0;
    __syncthreads();
    const u64 cap = table.mask + 1, stride = (u64)gridDim.x * blockDim.x;
    for (u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x; s < cap; s += stride) {
        u32 c; const u64 *slot;
        if (!slot_record<W>(table, s, c, slot)) continue;
        atomicAdd(&lc[(unsigned)(((slot[0] & 1023) * n_parts) >> 10)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < n_parts && lc[threadIdx.x]) atomicAdd(&part_counts[threadIdx.x], (u64)lc[threadIdx.x]);
}

template <int W>
__global__ void __launch_bounds__(256)
part_scatter_kernel(TableView<W> table, unsigned n_parts, u64 *__restrict__ cursors /* start offsets, advanced */,
                    u64 *__restrict__ out_keys, u32 *__restrict__ out_counts)
{
    __shared__ u32 lc[PB_MAX_PARTS];
    __shared__ u64 lbase[PB_MAX_PARTS];
    const u64 cap = table.mask + 1, stride = (u64)gridDim.x * blockDim.x;
    const u64 n_iter = (cap + stride - 1) / stride;
    u64 s = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    for (u64 it = 0; it < n_iter; ++it, s += stride) {
        if (threadIdx.x < PB_MAX_PARTS) lc[threadIdx.x] = 0;
        __syncthreads();
        u32 c = 0; const u64 *slot = nullptr;
        const bool take = s < cap && slot_record<W>(table, s, c, slot);
        unsigned owner = 0; u32 local = 0;
        if (take) { owner = (unsigned)(((slot[0] & 1023) * n_parts) >> 10); local = atomicAdd(&lc[owner], 1u); }
        __syncthreads();
        if (threadIdx.x < n_parts && lc[threadIdx.x]) lbase[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (u64)lc[threadIdx.x]);
        __syncthreads();
        if (take) {
            const u64 pos = lbase[owner] + local;
#pragma unroll
            for (int i = 0; i < W; ++i) out_keys[pos * W + i] = slot[i];
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

#define PB_STREAM_PAD_WORDS64 8      /* 512 positions of padding: a 160-mer window never leaves the buffer */

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
    ctx->stream_len = new_len;
    ctx->n_reads += n_reads;
    ctx->n_bases += n_new_bases;
    return PB_OK;
}

} // extern "C"

template <int W>
static int count_impl(pb_ctx *ctx, unsigned k)
{
    int rc;
    if ((rc = ensure(ctx, ctx->d_scalars, 16 * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->d_hist, 65536 * 8)) != PB_OK) return rc;
    // capacity: a power of two >= half the number of windows (distinct k-mers are a fraction of the
    // instances at any useful coverage); doubled and redone if a probe sequence ever overflows.
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
            count_kernel<W><<<grid_for(ctx->stream_len, 256, getenv("PB_GRIDCAP") ? atoi(getenv("PB_GRIDCAP")) : 256 * 32), 256, 0, ctx->stream>>>(
                (const u64 *)ctx->packed.p, (const u64 *)ctx->nmask.p, ctx->stream_len, k, tv, 256u, (u64 *)ctx->d_scalars.p);
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
    {
        TableView<W> tv{(u64 *)ctx->table.p, ctx->table_cap - 1};
        HIPCHK(hipMemsetAsync(ctx->d_hist.p, 0, 65536 * 8, ctx->stream));
        StageTimer tm(ctx, "hist");
        hist_kernel<W><<<grid_for(ctx->table_cap, 256, 256 * 8), 256, 0, ctx->stream>>>(tv, (u64 *)ctx->d_hist.p);
        HIPCHK(hipGetLastError());
        if ((rc = tm.stop()) != PB_OK) return rc;
    }
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
    int rc;
    switch (ctx->W) {
    case 1: rc = count_impl<1>(ctx, k); break;
    case 2: rc = count_impl<2>(ctx, k); break;
    case 3: rc = count_impl<3>(ctx, k); break;
    case 4: rc = count_impl<4>(ctx, k); break;
    default: rc = count_impl<5>(ctx, k); break;
    }
    if (rc != PB_OK) return rc;
    std::vector<u64> h(65536);
    HIPCHK(hipMemcpyAsync(h.data(), ctx->d_hist.p, 65536 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->n_distinct = h[65535];
    if (occ_hist) { for (int i = 0; i < PB_OCC_BINS; ++i) occ_hist[i] = h[i]; }
    if (max_occ) { *max_occ = 0; for (int i = PB_COUNT_MAX; i > 0; --i) if (h[i]) { *max_occ = (u64)i; break; } }
    if (n_instances) *n_instances = ctx->n_instances;
    if (n_distinct) *n_distinct = ctx->n_distinct;
    return PB_OK;
}

} // extern "C"

template <int W>
static int compact_impl(pb_ctx *ctx, unsigned min_occ, DevBuf &keys, DevBuf &counts, u64 *n_out)
{
    int rc;
    const u64 cap_out = std::max<u64>(ctx->n_distinct, 1);
    if ((rc = ensure(ctx, keys, cap_out * W * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, counts, cap_out * 4)) != PB_OK) return rc;
    TableView<W> tv{(u64 *)ctx->table.p, ctx->table_cap - 1};
    HIPCHK(hipMemsetAsync(ctx->d_scalars.p, 0, 16 * 8, ctx->stream));
    StageTimer tm(ctx, "compact");
    compact_kernel<W><<<grid_for(ctx->table_cap, 256, 256 * 8), 256, 0, ctx->stream>>>(
        tv, min_occ, (u64 *)keys.p, (u32 *)counts.p, cap_out, (u64 *)ctx->d_scalars.p);
    HIPCHK(hipGetLastError());
    if ((rc = tm.stop()) != PB_OK) return rc;
    HIPCHK(hipMemcpyAsync(n_out, ctx->d_scalars.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

static int compact_dispatch(pb_ctx *ctx, unsigned min_occ, DevBuf &keys, DevBuf &counts, u64 *n_out)
{
    switch (ctx->W) {
    case 1: return compact_impl<1>(ctx, min_occ, keys, counts, n_out);
    case 2: return compact_impl<2>(ctx, min_occ, keys, counts, n_out);
    case 3: return compact_impl<3>(ctx, min_occ, keys, counts, n_out);
    case 4: return compact_impl<4>(ctx, min_occ, keys, counts, n_out);
    default: return compact_impl<5>(ctx, min_occ, keys, counts, n_out);
    }
}

extern "C" {

int pb_retain(pb_ctx *ctx, unsigned min_occ, uint64_t *n_keys)
{
    if (!ctx || ctx->W == 0 || !ctx->table.p) { if (ctx) ctx->err = "pb_retain before pb_count"; return PB_EINVAL; }
    (void)hipSetDevice(ctx->dev);
    int rc;
    const int W = ctx->W;
    DevBuf &ck = ctx->scratch[2], &cc = ctx->scratch[3];
    u64 n = 0;
    if ((rc = compact_dispatch(ctx, min_occ, ck, cc, &n)) != PB_OK) return rc;
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
    if (!ctx || ctx->W == 0 || !ctx->table.p) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    int rc;
    DevBuf &ck = ctx->scratch[2], &cc = ctx->scratch[3];
    u64 m = 0;
    if ((rc = compact_dispatch(ctx, 1, ck, cc, &m)) != PB_OK) return rc;
    if (n) *n = m;
    const u64 take = std::min<u64>(m, (u64)cap);
    if (take && keys) HIPCHK(hipMemcpyAsync(keys, ck.p, take * ctx->W * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (take && counts) HIPCHK(hipMemcpyAsync(counts, cc.p, take * 4, hipMemcpyDeviceToHost, ctx->stream));
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

} // extern "C"

template <int W>
static int export_impl(pb_ctx *ctx, unsigned n_parts, uint64_t *part_counts)
{
    int rc;
    TableView<W> tv{(u64 *)ctx->table.p, ctx->table_cap - 1};
    if ((rc = ensure(ctx, ctx->d_scalars, 16 * 8)) != PB_OK) return rc;
    DevBuf &pc = ctx->scratch[0];
    if ((rc = ensure(ctx, pc, PB_MAX_PARTS * 8 * 2)) != PB_OK) return rc;
    HIPCHK(hipMemsetAsync(pc.p, 0, PB_MAX_PARTS * 8 * 2, ctx->stream));
    StageTimer tm(ctx, "export");
    part_count_kernel<W><<<grid_for(ctx->table_cap, 256, 256 * 8), 256, 0, ctx->stream>>>(tv, n_parts, (u64 *)pc.p);
    u64 h[PB_MAX_PARTS];
    HIPCHK(hipMemcpyAsync(h, pc.p, n_parts * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    u64 cur[PB_MAX_PARTS], total = 0;
    for (unsigned i = 0; i < n_parts; ++i) { cur[i] = total; total += h[i]; part_counts[i] = h[i]; }
    ctx->exp_total = total;
    ctx->exp_parts = n_parts;
    if ((rc = ensure(ctx, ctx->exp_keys, std::max<u64>(total, 1) * W * 8)) != PB_OK) return rc;
    if ((rc = ensure(ctx, ctx->exp_counts, std::max<u64>(total, 1) * 4)) != PB_OK) return rc;
    u64 *cursors = (u64 *)pc.p + PB_MAX_PARTS;
    HIPCHK(hipMemcpyAsync(cursors, cur, n_parts * 8, hipMemcpyHostToDevice, ctx->stream));
    part_scatter_kernel<W><<<grid_for(ctx->table_cap, 256, 256 * 8), 256, 0, ctx->stream>>>(tv, n_parts, cursors, (u64 *)ctx->exp_keys.p,
                                                                                         (u32 *)ctx->exp_counts.p);
    HIPCHK(hipGetLastError());
    return tm.stop();
}

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
    return PB_OK;
}

template <int W>
static int merge_begin_impl(pb_ctx *ctx, u64 cap)
{
    int rc;
    if ((rc = ensure(ctx, ctx->merge_table, cap * PB_SLOT_WORDS(W) * 8)) != PB_OK) return rc;
    ctx->merge_cap = cap;
    table_init_kernel<W><<<grid_for(cap * PB_SLOT_WORDS(W), 256, 256 * 32), 256, 0, ctx->stream>>>((u64 *)ctx->merge_table.p, cap);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PB_OK;
}

template <int W>
static int hist_only_impl(pb_ctx *ctx)
{
    int rc;
    if ((rc = ensure(ctx, ctx->d_hist, 65536 * 8)) != PB_OK) return rc;
    TableView<W> tv{(u64 *)ctx->table.p, ctx->table_cap - 1};
    HIPCHK(hipMemsetAsync(ctx->d_hist.p, 0, 65536 * 8, ctx->stream));
    StageTimer tm(ctx, "hist");
    hist_kernel<W><<<grid_for(ctx->table_cap, 256, 256 * 8), 256, 0, ctx->stream>>>(tv, (u64 *)ctx->d_hist.p);
    HIPCHK(hipGetLastError());
    return tm.stop();
}

#define PB_DISPATCH_W(ctx, fn, ...)                                   \
    ((ctx)->W == 1 ? fn<1>(__VA_ARGS__) : (ctx)->W == 2 ? fn<2>(__VA_ARGS__) : (ctx)->W == 3 ? fn<3>(__VA_ARGS__) \
     : (ctx)->W == 4 ? fn<4>(__VA_ARGS__) : fn<5>(__VA_ARGS__))

extern "C" {

int pb_export_begin(pb_ctx *ctx, int n_parts, uint64_t *part_counts)
{
    if (!ctx || !part_counts || n_parts < 1 || n_parts > PB_MAX_PARTS || ctx->W == 0 || !ctx->table.p) return PB_EINVAL;
    (void)hipSetDevice(ctx->dev);
    return PB_DISPATCH_W(ctx, export_impl, ctx, (unsigned)n_parts, part_counts);
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
    int rc = PB_DISPATCH_W(ctx, hist_only_impl, ctx);
    if (rc != PB_OK) return rc;
    std::vector<u64> h(65536);
    HIPCHK(hipMemcpyAsync(h.data(), ctx->d_hist.p, 65536 * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    ctx->n_distinct = h[65535];
    if (occ_hist) for (int i = 0; i < PB_OCC_BINS; ++i) occ_hist[i] = h[i];
    if (max_occ) { *max_occ = 0; for (int i = PB_COUNT_MAX; i > 0; --i) if (h[i]) { *max_occ = (u64)i; break; } }
    if (n_distinct) *n_distinct = ctx->n_distinct;
    return PB_OK;
}

} // extern "C"
