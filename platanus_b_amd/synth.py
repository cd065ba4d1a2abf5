"""Seeded synthetic Illumina-like read sets.

Uniform-random ACGT genome, paired-end reads of fixed length, fragment length ~N(350,30) clipped to
>= read length, uniform start, random strand, independent substitution errors (uniform over the three
other bases), optional N rate.  All randomness comes from a counter-based splitmix64 so the same
(seed, parameters) give the same bytes on every numpy version -- the golden fixtures under
tests/golden/ depend on that.

Nothing here is taken from the reference: Platanus_B ships no read simulator.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 counters."""
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def rand_u64(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n pseudo-random uint64 from (seed, stream); element i depends only on (seed, stream, offset+i)."""
    base = splitmix64(np.array([(seed * 0x100 + stream) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        return splitmix64(idx * np.uint64(0xD1342543DE82EF95) + base)


def rand_unit(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 in [0,1)."""
    return (rand_u64(seed, stream, n, offset) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def make_genome(genome_len: int, seed: int) -> np.ndarray:
    """uint8 codes 0..3 (A,C,G,T)."""
    return (rand_u64(seed, 0, genome_len) >> np.uint64(62)).astype(np.uint8)


def make_haplotype2(genome: np.ndarray, seed: int, het: float) -> np.ndarray:
    """Second haplotype: the genome with substitutions at rate `het` (bubbles in the de Bruijn graph)."""
    n = genome.size
    hit = rand_unit(seed, 40, n) < het
    sub = (rand_u64(seed, 41, n) % np.uint64(3)).astype(np.uint8)
    h2 = genome.copy()
    h2[hit] = (h2[hit] + 1 + sub[hit]) & 3
    return h2


def make_reads(genome_len: int, coverage: float, seed: int, read_len: int = 150, err: float = 0.005,
               n_rate: float = 0.0, frag_mean: float = 350.0, frag_sd: float = 30.0,
               genome: np.ndarray | None = None, pair_lo: int = 0, pair_hi: int | None = None, het: float = 0.0):
    """Return (r1, r2): uint8 [n_pairs, read_len] code matrices (0..3, 4 = N).

    n_pairs = floor(G*cov / (2*read_len)).  pair_lo/pair_hi select a sub-range of pairs (used to shard the
    same read set over ranks, and to stream large sets without holding them whole).
    """
    if genome is None:
        genome = make_genome(genome_len, seed)
    n_total = int(genome_len * coverage) // (2 * read_len)
    if pair_hi is None:
        pair_hi = n_total
    pair_hi = min(pair_hi, n_total)
    n = max(pair_hi - pair_lo, 0)
    # fragment length: Irwin-Hall(12) - 6 ~ N(0,1)
    z = np.zeros(n, dtype=np.float64)
    for j in range(12):
        z += rand_unit(seed, 16 + j, n, pair_lo)
    z -= 6.0
    flen = np.rint(frag_mean + frag_sd * z).astype(np.int64)
    flen = np.clip(flen, read_len, genome_len)
    start = (rand_unit(seed, 1, n, pair_lo) * (genome_len - flen + 1)).astype(np.int64)
    strand = (rand_u64(seed, 2, n, pair_lo) >> np.uint64(63)).astype(bool)
    ar = np.arange(read_len, dtype=np.int64)
    src = genome
    hap = np.zeros(n, dtype=np.int64)
    if het > 0:                                   # odd pairs are sampled from the second haplotype
        src = np.concatenate([genome, make_haplotype2(genome, seed, het)])
        hap = ((np.arange(pair_lo, pair_lo + n, dtype=np.int64) & 1) * genome_len)
    left = src[(start + hap)[:, None] + ar[None, :]]                              # 5' end of fragment, forward
    right = 3 - src[(start + hap + flen - 1)[:, None] - ar[None, :]]              # revcomp of 3' end
    r1 = np.where(strand[:, None], right, left).astype(np.uint8)
    r2 = np.where(strand[:, None], left, right).astype(np.uint8)
    for which, r in ((0, r1), (1, r2)):
        flat = r.reshape(-1)
        off = pair_lo * read_len
        if err > 0:
            u = rand_unit(seed, 3 + which, flat.size, off)
            hit = u < err
            sub = (rand_u64(seed, 5 + which, flat.size, off) % np.uint64(3)).astype(np.uint8)
            flat[hit] = (flat[hit] + 1 + sub[hit]) & 3
        if n_rate > 0:
            u = rand_unit(seed, 7 + which, flat.size, off)
            flat[u < n_rate] = 4
    return r1, r2


_CODE2ASCII = np.frombuffer(b"ACGTN", dtype=np.uint8)


def write_fastq(path: str, codes: np.ndarray, mate: int, first_index: int = 0, append: bool = False) -> None:
    """4-line FASTQ, fixed quality 'I', fixed-width names so the file is written as one byte matrix."""
    n, L = codes.shape
    name_w = 1 + 1 + 10 + 2                      # '@' 'r' 10 digits '/' mate
    rec = name_w + 1 + L + 1 + 2 + L + 1
    buf = np.empty((n, rec), dtype=np.uint8)
    idx = np.arange(first_index, first_index + n, dtype=np.int64)
    buf[:, 0] = ord("@")
    buf[:, 1] = ord("r")
    for d in range(10):
        buf[:, 2 + d] = (idx // 10 ** (9 - d)) % 10 + ord("0")
    buf[:, 12] = ord("/")
    buf[:, 13] = ord("0") + mate
    buf[:, 14] = ord("\n")
    buf[:, 15:15 + L] = _CODE2ASCII[codes]
    buf[:, 15 + L] = ord("\n")
    buf[:, 16 + L] = ord("+")
    buf[:, 17 + L] = ord("\n")
    buf[:, 18 + L:18 + 2 * L] = ord("I")
    buf[:, 18 + 2 * L] = ord("\n")
    with open(path, "ab" if append else "wb") as f:
        buf.tofile(f)


def write_pair_files(prefix: str, genome_len: int, coverage: float, seed: int, chunk_pairs: int = 200_000, **kw):
    """Write prefix_1.fq / prefix_2.fq in chunks; returns (n_pairs, read_len)."""
    read_len = kw.get("read_len", 150)
    n_total = int(genome_len * coverage) // (2 * read_len)
    genome = make_genome(genome_len, seed)
    first = True
    for lo in range(0, max(n_total, 1), chunk_pairs):
        r1, r2 = make_reads(genome_len, coverage, seed, genome=genome, pair_lo=lo, pair_hi=lo + chunk_pairs, **kw)
        write_fastq(prefix + "_1.fq", r1, 1, lo, append=not first)
        write_fastq(prefix + "_2.fq", r2, 2, lo, append=not first)
        first = False
    return n_total, read_len


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("prefix")
    ap.add_argument("--genome", type=int, default=50_000)
    ap.add_argument("--cov", type=float, default=50.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--err", type=float, default=0.005)
    ap.add_argument("--n-rate", type=float, default=0.0)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--het", type=float, default=0.0)
    a = ap.parse_args()
    n, L = write_pair_files(a.prefix, a.genome, a.cov, a.seed, err=a.err, n_rate=a.n_rate, read_len=a.read_len, het=a.het)
    print(f"{n} pairs x 2 x {L} bp -> {a.prefix}_1.fq {a.prefix}_2.fq")
