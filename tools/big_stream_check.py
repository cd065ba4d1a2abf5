"""Development aid: a read stream of more than 2^32 positions on one GPU (45 Mbp x 100x PE150 = 4.5 G bases, 3.57 G windows):
checks the 64-bit index arithmetic of the counting pass through size-independent properties and the two count paths against each
other.  usage: python tools/big_stream_check.py [genome_bp] [coverage]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from platanus_b_amd import synth
    from platanus_b_amd.binding import GpuCounter
    G = int(sys.argv[1]) if len(sys.argv) > 1 else 45_000_000
    cov = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
    genome = synth.make_genome(G, 3)
    n_pairs = int(G * cov) // 300
    g = GpuCounter(0)
    t0 = time.time()
    for lo in range(0, n_pairs, 500_000):
        r1, r2 = synth.make_reads(G, cov, 3, genome=genome, pair_lo=lo, pair_hi=min(lo + 500_000, n_pairs))
        g.reads_append_matrix(r1)
        g.reads_append_matrix(r2)
        if (lo // 500_000) % 5 == 0:
            print(f"  {lo + len(r1)} / {n_pairs} pairs resident, {time.time() - t0:.0f} s", flush=True)
    n_reads, n_bases = g.reads_info()
    print(f"{n_reads} reads, {n_bases} bases (2^32 = {1 << 32})", flush=True)
    res = {}
    for path in ("superk", "direct"):
        g.set_option("count_path", 1 if path == "direct" else 2)
        t0 = time.time()
        h = g.make_kmer_read_distribution(32).copy()
        dt = time.time() - t0
        took = "super-k-mer" if g.kernel_ms("part1") > 0 else "direct"
        assert g.n_instances == n_reads * 119, (g.n_instances, n_reads * 119)
        assert int(h.sum()) == g.n_distinct
        assert int((np.arange(65535, dtype=np.uint64) * h).sum()) == g.n_instances
        n = g.retain(3)
        keys, counts = g.fetch_retained()
        assert (keys[1:, 0] > keys[:-1, 0]).all()
        res[path] = (h, keys, counts)
        print(f"count_path={path}: took the {took} pass, {dt * 1e3:.0f} ms wall, kernels {g.kernel_ms('count'):.1f} ms, "
              f"{g.n_instances} windows, {g.n_distinct} distinct, {n} retained (>= 3), bucket bits {g.kernel_ms('sk_bucket_bits'):.0f}, "
              f"flagged {g.kernel_ms('sk_flagged_buckets'):.0f}", flush=True)
    assert np.array_equal(res["superk"][0], res["direct"][0])
    assert np.array_equal(res["superk"][1], res["direct"][1]) and np.array_equal(res["superk"][2], res["direct"][2])
    print("both passes agree; all properties hold")


if __name__ == "__main__":
    main()
