"""Synthetic gaps for the gap-closing graph tests: per gap a short random region (sometimes with an internal repeat), reads of it with
substitution errors and unknown bases, sometimes a sequence shorter than k.  Seeded: the golden dumps under tests/golden/gapclose/ were made
by the REFERENCE's own functions (oracle/_ref/gapclose_ref) on exactly these inputs (tests/golden/make_gapclose_golden.py)."""
import numpy as np

# name -> (k, min_coverage, gaps, seed, region length, read length, coverage, error rate, N rate)
CASES = {
    "k21_c2": (21, 2, 40, 121, 520, 100, 12, 0.01, 0.002),
    "k24_c1": (24, 1, 30, 124, 540, 100, 10, 0.01, 0.002),
    "k32_c2": (32, 2, 40, 132, 600, 100, 12, 0.01, 0.002),
    "k31_c3_deep": (31, 3, 12, 131, 1500, 150, 60, 0.02, 0.001),
    "k45_c2": (45, 2, 40, 145, 700, 150, 12, 0.01, 0.002),
    "k64_c2": (64, 2, 30, 164, 800, 150, 14, 0.008, 0.001),
    "k75_c2": (75, 2, 30, 175, 900, 150, 16, 0.008, 0.001),
    "k100_c2": (100, 2, 20, 200, 1200, 250, 16, 0.006, 0.001),
    "k140_c1": (140, 1, 16, 240, 1500, 250, 14, 0.004, 0.0005),
}


def make_gaps(name):
    k, mc, n_gaps, seed, glen, rl, cov, err, n_rate = CASES[name]
    rng = np.random.default_rng(seed)
    seqs, first = [], [0]
    for g in range(n_gaps):
        G = int(rng.integers(max(k + 5, glen // 3), glen))
        genome = rng.integers(0, 4, G).astype(np.uint8)
        if rng.random() < 0.35 and G > 2 * k + 60:
            a = int(rng.integers(0, G - (2 * k + 40)))
            genome[a + k + 20:a + 2 * k + 40] = genome[a:a + k + 20]            # a repeat longer than k inside the gap
        if rng.random() < 0.15:
            a = int(rng.integers(0, G - 30))
            genome[a:a + 30] = 3 if rng.random() < 0.5 else 0                     # a homopolymer run
        for _ in range(max(1, int(G * cov / rl))):
            L = int(min(rl, G))
            s = int(rng.integers(0, G - L + 1))
            r = genome[s:s + L].copy()
            e = rng.random(L) < err
            r[e] = (r[e] + 1 + rng.integers(0, 3, int(e.sum()))) & 3
            r[rng.random(L) < n_rate] = 4
            seqs.append(r)
        if rng.random() < 0.2:
            seqs.append(rng.integers(0, 4, max(1, k - 3)).astype(np.uint8))       # shorter than k: not counted
        if rng.random() < 0.1:
            seqs.append(np.full(k + 4, 4, dtype=np.uint8))                        # unknown bases only
        first.append(len(seqs))
    if n_gaps > 3:                                                               # one gap without any sequence
        first.insert(3, first[2])
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    return k, mc, np.array(first, dtype=np.uint64), off, np.concatenate(seqs)


# (k_low, k_high, min_coverage, case whose gaps are used): GapCloseDBG<k_low, k_high>::gapAssemble; (20, 80) is gap_close's own pair
# (gapClose.cpp:41-43: BRUIJN_MIN_KMER, BRUIJN_MAX_KMER, MIN_COVERAGE = 2)
ASSEMBLE_CASES = {
    "asm_20_80": (20, 80, 2, "k100_c2"),
    "asm_20_80_b": (20, 80, 2, "k75_c2"),
    "asm_25_60": (25, 60, 2, "k45_c2"),
    "asm_28_45": (28, 45, 2, "k31_c3_deep"),
    "asm_32_100": (32, 100, 2, "k100_c2"),
    "asm_40_90": (40, 90, 3, "k75_c2"),
}


# gaps WITH flanks (GAP::headSeq / tailSeq / start / end) for the closing decisions (gapCloseDBG.h:479-741): name -> (k_low, k_high, seed, gaps)
CLOSE_CASES = {
    "close_20_80": (20, 80, 301, 60),
    "close_25_60": (25, 60, 302, 40),
    "close_40_90": (40, 90, 303, 30),
}


def make_gaps_with_flanks(name):
    """A region = [left contig end][gap][right contig start]; reads from the whole region (20-40x, 150 bases, 0.5 % errors); the flanks are the
    last / first 40-120 bases of the contigs, sometimes with an error, sometimes too short (< 32: not enabled), sometimes the gap lies at a
    scaffold's end (start = 0 or the right end: extendEdge).  Some gaps get too few reads to be closed (partial closing or nothing)."""
    import struct
    kl, kh, seed, n_gaps = CLOSE_CASES[name]
    rng = np.random.default_rng(seed)
    seqs, first, flanks = [], [0], b""
    for g in range(n_gaps):
        hl, tl = int(rng.integers(40, 120)), int(rng.integers(40, 120))
        if rng.random() < 0.1:
            hl = int(rng.integers(5, 31))
        G = int(rng.integers(30, 900))
        region = rng.integers(0, 4, 200 + hl + G + tl + 200).astype(np.uint8)
        lo, hi = 200, 200 + hl + G + tl
        cov = float(rng.choice([3, 12, 25, 40]))
        n_reads = max(2, int((hi - lo + 300) * cov / 150))
        holes = rng.random() < 0.25                                   # a stretch of the gap that no read covers
        for _ in range(n_reads):
            L = 150
            s = int(rng.integers(lo - 150, hi + 1))
            s = max(0, min(s, len(region) - L))
            if holes and s < lo + hl + G // 2 < s + L:
                continue
            r = region[s:s + L].copy()
            e = rng.random(L) < 0.005
            r[e] = (r[e] + 1 + rng.integers(0, 3, int(e.sum()))) & 3
            seqs.append(r)
        first.append(len(seqs))
        head = region[lo:lo + hl].copy()
        tail = region[lo + hl + G:hi].copy()
        if rng.random() < 0.2:
            head[int(rng.integers(0, len(head)))] ^= 1                # a mismatch inside a flank
        u = rng.random()
        at_end = u < 0.25
        start = 0 if u < 0.12 else 5000 + g
        end = start + G + int(rng.integers(-20, 20))
        flanks += struct.pack("<iiBII", start, end, 1 if at_end else 0, len(head), len(tail)) + head.tobytes() + tail.tobytes()
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    return kl, kh, np.array(first, dtype=np.uint64), off, np.concatenate(seqs), flanks
