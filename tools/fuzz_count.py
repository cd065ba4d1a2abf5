"""Development aid: differential test of the super-k-mer counting pass against the direct (global table) pass on random
inputs -- random genome size, coverage, error and N rates, read lengths (mixed), k in 1..96 (two-word keys above 32: kernels_superk2.inc, three-word keys above 64: kernels_superk3.inc), bucket bits.  Both run on the
GPU, so hundreds of cases take a minute.  usage: python tools/fuzz_count.py [n_cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from platanus_b_amd.binding import GpuCounter
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for case in range(n_cases):
        G = int(rng.integers(2_000, 300_000))
        genome = rng.integers(0, 4, G).astype(np.uint8)
        if rng.random() < 0.3:                                  # low-complexity insert
            a = int(rng.integers(0, G - 500))
            unit = rng.integers(0, 4, int(rng.integers(1, 6))).astype(np.uint8)
            genome[a:a + 400] = np.resize(unit, 400)
        cov = float(rng.choice([2, 10, 40, 120]))
        err = float(rng.choice([0.0, 0.005, 0.03]))
        n_rate = float(rng.choice([0.0, 0.001, 0.02]))
        u = rng.random()
        k = int(rng.integers(1, 33)) if u < 0.45 else int(rng.integers(33, 65)) if u < 0.75 else int(rng.integers(65, 97))
        n_bases, reads = 0, []
        while n_bases < G * cov and n_bases < 6_000_000:
            L = int(rng.choice([int(rng.integers(1, 60)), 100, 150, 151, 250, int(rng.integers(300, 3000))]))
            L = min(L, G)
            s = int(rng.integers(0, G - L + 1))
            r = genome[s:s + L].copy()
            if rng.random() < 0.5:
                r = (3 - r)[::-1]
            e = rng.random(L) < err
            r[e] = (r[e] + 1 + rng.integers(0, 3, int(e.sum()))) & 3
            r[rng.random(L) < n_rate] = 4
            reads.append(r)
            n_bases += L
        lens = np.array([len(r) for r in reads], dtype=np.uint64)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        codes = np.concatenate(reads)
        res = {}
        skb = str(int(rng.integers(2, 21))) if rng.random() < 0.5 else None

        def run(path):
            g = GpuCounter(0)
            g.set_option("count_path", 1 if path == "direct" else 2)
            if path == "superk" and skb:
                g.set_option("sk_bucket_bits", int(skb))
            g.reads_append(codes, off)
            hist = g.make_kmer_read_distribution(k)
            took = g.kernel_ms("part1") > 0
            g.retain(1)
            keys, counts = g.fetch_retained()
            out = (hist.copy(), keys.copy(), counts.copy(), g.n_instances, took)
            g.close()
            return out

        for path in ("direct", "superk"):
            res[path] = run(path)
        d, s_ = res["direct"], res["superk"]
        ok = np.array_equal(d[0], s_[0]) and np.array_equal(d[1], s_[1]) and np.array_equal(d[2], s_[2]) and d[3] == s_[3]
        if not ok:
            bad += 1
            for path in ("direct", "superk"):                  # which side is unstable, and how?
                for rep in range(3):
                    o = run(path)
                    print(f"    rerun {path} {rep}: equals first run of direct {np.array_equal(o[1], d[1]) and np.array_equal(o[2], d[2]) and np.array_equal(o[0], d[0])}, of superk "
                          f"{np.array_equal(o[1], s_[1]) and np.array_equal(o[2], s_[2]) and np.array_equal(o[0], s_[0])}  n_keys {len(o[1])} skb={skb}")
            ka, kb = d[1].reshape(len(d[2]), -1)[:, 0], s_[1].reshape(len(s_[2]), -1)[:, 0]
            only_a, only_b = np.setdiff1d(ka, kb), np.setdiff1d(kb, ka)
            print(f"    keys only in direct {len(only_a)} {only_a[:4]}, only in superk {len(only_b)} {only_b[:4]}")
            if len(ka) == len(kb) and np.array_equal(ka, kb):
                diff = np.nonzero(d[2] != s_[2])[0]
                print(f"    same keys; counts differ at {len(diff)} places: {[(hex(int(ka[i])), int(d[2][i]), int(s_[2][i])) for i in diff[:6]]}")
            print(f"MISMATCH case {case}: G={G} cov={cov} err={err} n={n_rate} k={k} reads={len(reads)} instances {d[3]} vs {s_[3]} distinct {len(d[1])} vs {len(s_[1])}", flush=True)
        elif case % 10 == 0:
            print(f"case {case}: ok (G={G} cov={cov} k={k} reads={len(reads)} instances={d[3]} distinct={len(d[1])} superk={s_[4]})", flush=True)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
