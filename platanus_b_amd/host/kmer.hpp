// Host-side k-mer primitives (shared by the product's C++ host and, for the same arithmetic, by the oracle).
// Restates (does not copy) the reference's key types:
//   * 2-bit code A=0 C=1 G=2 T=3, N=4, other -> 46            reference common.h:256-257
//   * Kmer31  (k<=32, one u64; base i of the k-mer, 0 = 5' end, at bit 2*(k-1-i))   kmer.h:84-191
//   * KmerN<BinstrNN> (W = ceil(k/32) words, word 0 = least significant 32 bases)   kmer.h:193-310,
//     binstr.h:292-721; operator< compares from the most significant word down (binstr.h:460-466),
//     i.e. the order of the 2k-bit integer.
// Here every width is one template: Key<W> is a 2k-bit little-endian-word integer.
#pragma once
#include <cstdint>
#include <cstring>
#include <functional>

namespace pbh {

typedef uint64_t u64;
typedef uint16_t u16;

// common.h:256  ".\x0.\x1\x3..\x2......\x4"[c & 0xF]
inline unsigned char char2bin(char c)
{
    static const unsigned char t[16] = {46, 0, 46, 1, 3, 46, 46, 2, 46, 46, 46, 46, 46, 46, 4, 46};
    return t[c & 0xF];
}
inline char bin2char(unsigned b) { return "ACGTN"[b]; }

inline int words_for_k(unsigned k) { return (int)((k + 31) / 32); }

template <int W>
struct Key {
    u64 w[W];

    Key() { for (int i = 0; i < W; ++i) w[i] = 0; }
    bool operator==(const Key &o) const
    {
        for (int i = 0; i < W; ++i) if (w[i] != o.w[i]) return false;
        return true;
    }
    bool operator!=(const Key &o) const { return !(*this == o); }
    bool operator<(const Key &o) const          // binstr.h:460-466 (most significant word first)
    {
        for (int i = W - 1; i > 0; --i) if (w[i] != o.w[i]) return w[i] < o.w[i];
        return w[0] < o.w[0];
    }
    unsigned get(unsigned pos) const { return (unsigned)((w[pos >> 5] >> ((pos & 31) * 2)) & 3); }
    void set(unsigned pos, unsigned v)
    {
        w[pos >> 5] = (w[pos >> 5] & ~(3ull << ((pos & 31) * 2))) | ((u64)v << ((pos & 31) * 2));
    }
    // keep only the low 2k bits
    void mask(unsigned k)
    {
        const unsigned top = (k - 1) >> 5;
        for (unsigned i = top + 1; i < (unsigned)W; ++i) w[i] = 0;
        if (k & 31) w[top] &= (1ull << (2 * (k & 31))) - 1;
    }
    void shl2(unsigned k)                         // <<= 2 then mask (counter.h:418-419, binstr.h:434-436)
    {
        for (int i = W - 1; i > 0; --i) w[i] = (w[i] << 2) | (w[i - 1] >> 62);
        w[0] <<= 2;
        mask(k);
    }
    void shr2()
    {
        for (int i = 0; i < W - 1; ++i) w[i] = (w[i] >> 2) | (w[i + 1] << 62);
        w[W - 1] >>= 2;
    }
};

// std::hash<u64> is the identity in libstdc++; BinstrNN::hasher is the XOR of all words
// (binstr.h:535-538,595-598,655-658,716-719).  Container iteration order depends on it.
template <int W>
struct KeyHash {
    size_t operator()(const Key<W> &k) const
    {
        u64 h = 0;
        for (int i = 0; i < W; ++i) h ^= k.w[i];
        return (size_t)h;
    }
};

inline u64 revcom_u64(u64 t)                      // kmer.h:53-60
{
    t = ~t;
    t = ((t & 0x3333333333333333ull) << 2) | ((t & 0xCCCCCCCCCCCCCCCCull) >> 2);
    t = ((t & 0x0F0F0F0F0F0F0F0Full) << 4) | ((t & 0xF0F0F0F0F0F0F0F0ull) >> 4);
    t = ((t & 0x00FF00FF00FF00FFull) << 8) | ((t & 0xFF00FF00FF00FF00ull) >> 8);
    t = ((t & 0x0000FFFF0000FFFFull) << 16) | ((t & 0xFFFF0000FFFF0000ull) >> 16);
    return (t << 32) | (t >> 32);
}

// reverse complement of a k-mer held in the low 2k bits (kmer.h:139-145, 262-285)
template <int W>
inline Key<W> revcomp(const Key<W> &f, unsigned k)
{
    const int n = words_for_k(k);
    u64 tmp[W + 1];
    for (int i = 0; i < n; ++i) tmp[i] = revcom_u64(f.w[n - 1 - i]);
    tmp[n] = 0;
    const unsigned j = (k & 31) * 2;
    Key<W> r;
    if (j == 0) {
        for (int i = 0; i < n; ++i) r.w[i] = tmp[i];
    } else {
        for (int i = 0; i < n; ++i) r.w[i] = (tmp[i] >> (64 - j)) | (i + 1 < n ? (tmp[i + 1] << j) : 0);
    }
    return r;
}

// forward/reverse pair that is rolled along a read (counter.h:413-422)
template <int W>
struct Kmer {
    Key<W> fwd, rev;
    unsigned k;
    explicit Kmer(unsigned k_) : k(k_) {}
    void push_right(unsigned b)                   // append base at the 3' end
    {
        fwd.shl2(k);
        fwd.set(0, b);
        rev.shr2();
        rev.set(k - 1, 3 ^ b);
    }
    void push_left(unsigned b)                    // prepend base at the 5' end
    {
        fwd.shr2();
        fwd.set(k - 1, b);
        rev.shl2(k);
        rev.set(0, 3 ^ b);
    }
    const Key<W> &canon() const { return rev < fwd ? rev : fwd; }   // std::min(fwd, rev)
    void set_forward(const Key<W> &f) { fwd = f; rev = revcomp(f, k); }
};

} // namespace pbh
