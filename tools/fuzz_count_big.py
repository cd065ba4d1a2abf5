"""Development aid: the differential test of tools/fuzz_count.py at sizes where the library chooses the super-k-mer pass and its bucket
count by itself (30-400 M bases): genomes with long tandem repeats, low-complexity stretches and duplicated segments (heavy minimizers,
flagged buckets, the spill list), mixed coverage.  Super-k-mer pass vs direct pass, both on the GPU.
usage: python tools/fuzz_count_big.py [n_cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from platanus_b_amd.binding import GpuCounter
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for case in range(n_cases):
        t0 = time.time()
        G = int(rng.integers(300_000, 4_000_000))
        genome = rng.integers(0, 4, G).astype(np.uint8)
        notes = []
        for _ in range(int(rng.integers(0, 4))):                 # tandem repeats / low complexity
            a = int(rng.integers(0, G - 60_000))
            span = int(rng.integers(2_000, 50_000))
            unit = rng.integers(0, 4, int(rng.choice([1, 2, 3, 7, 31, 150]))).astype(np.uint8)
            genome[a:a + span] = np.resize(unit, span)
            notes.append(f"tandem{len(unit)}x{span}")
        for _ in range(int(rng.integers(0, 3))):                 # duplicated segment (optionally reverse-complemented)
            span = int(rng.integers(5_000, 100_000))
            a, b = int(rng.integers(0, G - span)), int(rng.integers(0, G - span))
            seg = genome[a:a + span].copy()
            genome[b:b + span] = (3 - seg)[::-1] if rng.random() < 0.5 else seg
            notes.append(f"dup{span}")
        cov = float(rng.choice([3, 15, 60, 120]))
        err = float(rng.choice([0.0, 0.005, 0.02]))
        n_rate = float(rng.choice([0.0, 0.0005]))
        k = int(rng.choice([32, 32, 31, 25, 21, 17, 33, 41, 48, 55, 64, 65, 75, 83, 96]))
        L = int(rng.choice([100, 150, 250]))
        n_reads = min(int(G * cov / L), 400_000_000 // L)
        g = GpuCounter(0)
        total = 0
        for lo in range(0, n_reads, 500_000):
            m = min(500_000, n_reads - lo)
            s = rng.integers(0, G - L + 1, m)
            r = genome[s[:, None] + np.arange(L)[None, :]]
            flip = rng.random(m) < 0.5
            r[flip] = (3 - r[flip])[:, ::-1]
            if err > 0:
                e = rng.random(r.shape) < err
                r[e] = (r[e] + 1 + rng.integers(0, 3, int(e.sum()))) & 3
            if n_rate > 0:
                r[rng.random(r.shape) < n_rate] = 4
            g.reads_append_matrix(np.ascontiguousarray(r))
            total += m * L
        res = {}
        for path in ("superk", "direct"):
            g.set_option("count_path", 1 if path == "direct" else 2)
            h = g.make_kmer_read_distribution(k).copy()
            took = g.kernel_ms("part1") > 0
            stats = (int(g.kernel_ms("sk_bucket_bits")), int(g.kernel_ms("sk_flagged_buckets")), int(g.kernel_ms("sk_spilled_records"))) if took else None
            g.retain(1)
            keys, counts = g.fetch_retained()
            res[path] = (h, keys, counts, g.n_instances, took, stats)
        a, b = res["superk"], res["direct"]
        ok = a[3] == b[3] and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        bad += not ok
        print(f"{'ok' if ok else 'MISMATCH'} case {case}: G={G} cov={cov} err={err} n={n_rate} k={k} L={L} bases={total} windows={a[3]} distinct={len(a[2])} "
              f"superk={a[4]} (bits, flagged, spilled)={a[5]} {' '.join(notes)} [{time.time() - t0:.0f} s]", flush=True)
        del g
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
