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
