// TEST INFRASTRUCTURE ONLY -- oracle restatement of the reference's DoubleHash<KEY,u16>
// (doubleHash.h:34-295): power-of-two open addressing, value 0 == empty slot,
//   h1   = sum_w (w + (w >> indexLength) + (w >> shifter)) & (N-1)          doubleHash.h:119-130
//   step = sum_w (~w ^ (w >> indexLength) ^ (w >> shifter)) | 1             doubleHash.h:135-146
//   indexLength = bit length of N (= log2 N + 1), shifter = indexLength >= 32 ? 0 : 2*indexLength
//                                                                            doubleHash.h:98-115
// The slot a key lands in is part of the observable result (PREFIX_kmer_occ.bin, doubleHash.h:266-278).
#pragma once
#include "../../platanus_b_amd/host/kmer.hpp"
#include <vector>
#include <stdexcept>

namespace pbo {
using namespace pbh;

template <int W>
struct DHash {
    u64 nmask = 0;
    unsigned index_len = 0, shifter = 0;
    std::vector<Key<W>> keys;
    std::vector<u16> vals;

    static unsigned calc_length(u64 len)
    {
        for (unsigned i = 1; i < 64; ++i) if ((len >> i) == 0) return i;
        return 64;
    }
    void resize(u64 n)
    {
        if (n & (n - 1)) throw std::runtime_error("DHash: size is not a power of two");
        keys.assign(n, Key<W>());
        vals.assign(n, 0);
        nmask = n - 1;
        index_len = calc_length(n);
        shifter = index_len >= 32 ? 0 : 2 * index_len;
    }
    void release()
    {
        std::vector<Key<W>>().swap(keys);
        std::vector<u16>().swap(vals);
        nmask = 0; index_len = 0; shifter = 0;
    }
    u64 size() const { return keys.size(); }
    u64 h1(const Key<W> &k) const
    {
        u64 v = 0;
        for (int i = 0; i < W; ++i) v += k.w[i] + (k.w[i] >> index_len) + (k.w[i] >> shifter);
        return v & nmask;
    }
    u64 step(const Key<W> &k) const
    {
        u64 v = 0;
        for (int i = 0; i < W; ++i) v += ~k.w[i] ^ (k.w[i] >> index_len) ^ (k.w[i] >> shifter);
        return v | 1;
    }
    // doubleHash.h:170-184: slot of key, or the first empty slot on its probe sequence
    u64 find_any(const Key<W> &k) const
    {
        u64 s = h1(k);
        if (!vals[s] || keys[s] == k) return s;
        const u64 st = step(k);
        s = (s + st) & nmask;
        while (vals[s]) {
            if (keys[s] == k) return s;
            s = (s + st) & nmask;
        }
        return s;
    }
    // doubleHash.h:202-218: gives up after `times` probes and returns whatever slot it stands on
    u64 find_times_any(const Key<W> &k, unsigned times) const
    {
        unsigned i = 1;
        u64 s = h1(k);
        if (!vals[s] || keys[s] == k) return s;
        const u64 st = step(k);
        s = (s + st) & nmask;
        while (vals[s] && i < times) {
            ++i;
            if (keys[s] == k) return s;
            s = (s + st) & nmask;
        }
        return s;
    }
    u16 value(const Key<W> &k) const { return vals[find_any(k)]; }          // counter.h:94-97
    void put(const Key<W> &k, u16 v)                                        // operator[] = (doubleHash.h:221-226)
    {
        const u64 s = find_any(k);
        keys[s] = k;
        vals[s] = v;
    }
};

// sizeof(std::pair<KEY, unsigned short>) the reference divides the -m budget by
// (counter.h:300, assemble.cpp:537-597,769; measured: SURVEY Appendix C)
inline u64 ref_pair_bytes(int W)
{
    static const u64 t[6] = {0, 16, 48, 56, 64, 72};
    return t[W];
}

// counter.h:300-309 / assemble.cpp:537-546,769-775: largest power of two <= memory / sizeof(pair),
// computed through double log/pow exactly as the reference does.
u64 table_size_for_memory(u64 memory_bytes, int W);
// counter.h:621-622 / assemble.cpp:766-767: 2^(floor(log2(n / 0.9)) + 1)
u64 table_size_for_keys(u64 n);

} // namespace pbo
