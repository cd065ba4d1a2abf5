// Device-side k-mer primitives for gfx950 (wave64).  Restates the reference's key types as a 2k-bit
// integer in W little-endian 64-bit words (kmer.h:84-310, binstr.h:292-721); nothing is compiled from
// the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef unsigned short u16;

#define PB_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull   // T^32 is never canonical (its reverse complement A^32 = 0 is smaller)

template <int W>
struct DKey {
    u64 w[W];
};

template <int W>
__device__ __forceinline__ bool key_eq(const DKey<W> &a, const DKey<W> &b)
{
    bool e = true;
#pragma unroll
    for (int i = 0; i < W; ++i) e &= (a.w[i] == b.w[i]);
    return e;
}
template <int W>
__device__ __forceinline__ bool key_lt(const DKey<W> &a, const DKey<W> &b)   // binstr.h:460-466
{
#pragma unroll
    for (int i = W - 1; i > 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i];
    return a.w[0] < b.w[0];
}

__device__ __forceinline__ u64 revcom_u64(u64 t)                              // kmer.h:53-60
{
    t = ~t;
    t = ((t & 0x3333333333333333ull) << 2) | ((t & 0xCCCCCCCCCCCCCCCCull) >> 2);
    t = ((t & 0x0F0F0F0F0F0F0F0Full) << 4) | ((t & 0xF0F0F0F0F0F0F0F0ull) >> 4);
    t = ((t & 0x00FF00FF00FF00FFull) << 8) | ((t & 0xFF00FF00FF00FF00ull) >> 8);
    t = ((t & 0x0000FFFF0000FFFFull) << 16) | ((t & 0xFFFF0000FFFF0000ull) >> 16);
    return (t << 32) | (t >> 32);
}

// reverse complement of the k-mer in the low 2k bits; n = words actually used = ceil(k/32) <= W
template <int W>
__device__ __forceinline__ DKey<W> key_revcomp(const DKey<W> &f, unsigned k)
{
    const int n = (int)((k + 31) >> 5);
    u64 tmp[W + 1];
#pragma unroll
    for (int i = 0; i < W; ++i) tmp[i] = (i < n) ? revcom_u64(f.w[n - 1 - i]) : 0;
    tmp[W] = 0;
    const unsigned j = (k & 31) * 2;
    DKey<W> r;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        if (i >= n) r.w[i] = 0;
        else if (j == 0) r.w[i] = tmp[i];
        else r.w[i] = (tmp[i] >> (64 - j)) | ((i + 1 < n) ? (tmp[i + 1] << j) : 0);
    }
    return r;
}

template <int W>
__device__ __forceinline__ DKey<W> key_canon(const DKey<W> &f, unsigned k)    // counter.h:429
{
    const DKey<W> r = key_revcomp<W>(f, k);
    return key_lt<W>(r, f) ? r : f;
}

template <int W>
__device__ __forceinline__ void key_mask(DKey<W> &a, unsigned k)
{
    const unsigned top = (k - 1) >> 5;
#pragma unroll
    for (int i = 0; i < W; ++i)
        if ((unsigned)i > top) a.w[i] = 0;
    if (k & 31) {
#pragma unroll
        for (int i = 0; i < W; ++i)
            if ((unsigned)i == top) a.w[i] &= (1ull << (2 * (k & 31))) - 1;
    }
}
// successor: append base b at the 3' end; predecessor: prepend base b at the 5' end
template <int W>
__device__ __forceinline__ DKey<W> key_succ(const DKey<W> &s, unsigned k, unsigned b)
{
    DKey<W> r;
#pragma unroll
    for (int i = W - 1; i > 0; --i) r.w[i] = (s.w[i] << 2) | (s.w[i - 1] >> 62);
    r.w[0] = (s.w[0] << 2) | b;
    key_mask<W>(r, k);
    return r;
}
template <int W>
__device__ __forceinline__ DKey<W> key_pred(const DKey<W> &s, unsigned k, unsigned b)
{
    DKey<W> r;
#pragma unroll
    for (int i = 0; i < W - 1; ++i) r.w[i] = (s.w[i] >> 2) | (s.w[i + 1] << 62);
    r.w[W - 1] = s.w[W - 1] >> 2;
    const unsigned pos = k - 1;
#pragma unroll
    for (int i = 0; i < W; ++i)
        if ((unsigned)i == (pos >> 5)) r.w[i] |= (u64)b << ((pos & 31) * 2);
    return r;
}

__device__ __forceinline__ u64 mix64(u64 x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
// inverse of mix64 (every step of the murmur3 finaliser is a bijection on 64 bits)
__device__ __forceinline__ u64 unmix64(u64 x)
{
    x ^= x >> 33; x *= 0x9cb4b2f8129337dbull;
    x ^= x >> 33; x *= 0x4f74430c22a54005ull;
    x ^= x >> 33;
    return x;
}
template <int W>
__device__ __forceinline__ u64 key_hash(const DKey<W> &a)
{
    u64 h = a.w[0];
#pragma unroll
    for (int i = 1; i < W; ++i) h = h * 0x9E3779B97F4A7C15ull + a.w[i];
    return mix64(h);
}

// ---- packed read stream -------------------------------------------------------------------------------
// Reads are concatenated with ONE unknown-base separator after every read, so a k-mer window that crosses
// a read boundary covers an unknown base and is skipped by the same test that skips windows over an N
// (counter.h:423-428).  Base j of the stream sits at bits [62-2*(j%32), 63-2*(j%32)] of packed[j/32]
// (first base in the top bits, as in Kmer31); bit (j%64) of nmask[j/64] is 1 for N / separator / padding.

// 64 bits of the stream starting at bit position `bit` (big-endian bit order)
__device__ __forceinline__ u64 stream_get64(const u64 *__restrict__ packed, u64 bit)
{
    const u64 wi = bit >> 6;
    const unsigned s = (unsigned)(bit & 63);
    const u64 hi = packed[wi];
    if (s == 0) return hi;
    return (hi << s) | (packed[wi + 1] >> (64 - s));
}

// forward k-mer starting at stream position g
template <int W>
__device__ __forceinline__ DKey<W> stream_kmer(const u64 *__restrict__ packed, u64 g, unsigned k)
{
    DKey<W> key;
    const u64 b0 = 2 * g, b1 = 2 * g + 2 * (u64)k;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const u64 lo_end = b1 - 64ull * i;                 // one past the last bit of word i
        if (lo_end <= b0) { key.w[i] = 0; continue; }
        if (lo_end - b0 >= 64) key.w[i] = stream_get64(packed, lo_end - 64);
        else key.w[i] = stream_get64(packed, b0) >> (64 - (lo_end - b0));
    }
    return key;
}

// true if any of the k mask bits starting at stream position g is set (k <= 160)
__device__ __forceinline__ bool stream_window_has_n(const u64 *__restrict__ nmask, u64 g, unsigned k)
{
    u64 wi = g >> 6;
    const unsigned s = (unsigned)(g & 63);
    unsigned remaining = k;
    // first (partial) word
    u64 w = nmask[wi] >> s;
    unsigned avail = 64 - s;
    while (true) {
        if (remaining <= avail) {
            const u64 m = remaining == 64 ? ~0ull : ((1ull << remaining) - 1);
            return (w & m) != 0;
        }
        if (w) return true;
        remaining -= avail;
        ++wi;
        w = nmask[wi];
        avail = 64;
    }
}

// ---- device occurrence table --------------------------------------------------------------------------
// Open addressing, linear probing, power-of-two capacity.  Slot = W key words + 1 meta word
// (meta low 32 bits = count, high 32 bits = state for W > 1: 0 empty, 1 being written, 2 ready).
// For W == 1 the key word itself is the state (PB_EMPTY_KEY = empty), so an insert is one 64-bit CAS
// plus one 32-bit add that hit the same 16-byte slot.
template <int W> struct TableView {
    u64 *slots;
    u64 mask;            // capacity - 1
};
#define PB_SLOT_WORDS(W) ((W) + 1)

// W > 1 slot protocol.  EVERY access to a slot that other workgroups may touch inside the same launch is a
// device-scope atomic RMW: on MI355X those execute at the memory side, while ordinary (even sc1) loads and
// stores go through the issuing XCD's own L2, and the eight L2s are not coherent with each other -- mixing
// the two on one line produced torn keys (measured, round 1).  Returns 1 = slot holds `key` (claimed now or
// earlier), 0 = slot holds another key, 2 = another lane is still writing it (retry).
template <int W>
__device__ __forceinline__ int slot_claim_or_match(u64 *slot, const DKey<W> &key)
{
    u32 *state = (u32 *)(slot + W) + 1;
    const u32 st = atomicCAS(state, 0u, 1u);
    if (st == 0) {
        u64 old = 0;
#pragma unroll
        for (int i = 0; i < W; ++i) old |= atomicExch(slot + i, key.w[i]);
        // `old` is zero (fresh slot); folding it into the published value makes the publish wait for
        // the returned key-word atomics, i.e. for their completion at the memory side
        atomicExch(state, 2u | (u32)(old >> 40));
        return 1;
    }
    if (st == 1) return 2;
    bool eq = true;
#pragma unroll
    for (int i = 0; i < W; ++i) eq &= (atomicOr(slot + i, 0ull) == key.w[i]);
    return eq ? 1 : 0;
}

// returns false if the probe limit was hit (table too full -> caller doubles the capacity and redoes)
template <int W>
__device__ __forceinline__ bool table_add(const TableView<W> &t, const DKey<W> &key, u32 amount, u32 max_probe)
{
    u64 s = key_hash<W>(key) & t.mask;
    if (W == 1) {
        for (u32 p = 0; p < max_probe; ++p) {
            u64 *slot = t.slots + s * 2;
            u64 prev = *(volatile u64 *)slot;
            if (prev == PB_EMPTY_KEY) prev = atomicCAS(slot, PB_EMPTY_KEY, key.w[0]);
            if (prev == PB_EMPTY_KEY || prev == key.w[0]) {
                atomicAdd((u32 *)(slot + 1), amount);
                return true;
            }
            s = (s + 1) & t.mask;
        }
        return false;
    } else {
        u32 p = 0;
        while (p < max_probe) {
            u64 *slot = t.slots + s * PB_SLOT_WORDS(W);
            const int st = slot_claim_or_match<W>(slot, key);
            if (st == 1) { atomicAdd((u32 *)(slot + W), amount); return true; }
            if (st == 2) { __builtin_amdgcn_s_sleep(1); continue; }     // being written: retry the same slot
            s = (s + 1) & t.mask;
            ++p;
        }
        return false;
    }
}

// atomicMax on the count (contig-derived coverage, graph.h:1762-1767); same claiming protocol
template <int W>
__device__ __forceinline__ bool table_max(const TableView<W> &t, const DKey<W> &key, u32 value, u32 max_probe)
{
    u64 s = key_hash<W>(key) & t.mask;
    if (W == 1) {
        for (u32 p = 0; p < max_probe; ++p) {
            u64 *slot = t.slots + s * 2;
            u64 prev = *(volatile u64 *)slot;
            if (prev == PB_EMPTY_KEY) prev = atomicCAS(slot, PB_EMPTY_KEY, key.w[0]);
            if (prev == PB_EMPTY_KEY || prev == key.w[0]) {
                atomicMax((u32 *)(slot + 1), value);
                return true;
            }
            s = (s + 1) & t.mask;
        }
        return false;
    } else {
        u32 p = 0;
        while (p < max_probe) {
            u64 *slot = t.slots + s * PB_SLOT_WORDS(W);
            const int st = slot_claim_or_match<W>(slot, key);
            if (st == 1) { atomicMax((u32 *)(slot + W), value); return true; }
            if (st == 2) { __builtin_amdgcn_s_sleep(1); continue; }
            s = (s + 1) & t.mask;
            ++p;
        }
        return false;
    }
}

// read-only lookup on a table no kernel is writing: count of key, 0 if absent
template <int W>
__device__ __forceinline__ u32 table_find(const TableView<W> &t, const DKey<W> &key)
{
    u64 s = key_hash<W>(key) & t.mask;
    while (true) {
        const u64 *slot = t.slots + s * PB_SLOT_WORDS(W);
        if (W == 1) {
            const u64 kk = slot[0];
            if (kk == key.w[0]) return (u32)slot[1];
            if (kk == PB_EMPTY_KEY) return 0;
        } else {
            const u64 meta = slot[W];
            if ((meta >> 32) == 0) return 0;
            bool eq = true;
#pragma unroll
            for (int i = 0; i < W; ++i) eq &= (slot[i] == key.w[i]);
            if (eq) return (u32)meta;
        }
        s = (s + 1) & t.mask;
    }
}
