"""Maximum-size check on one GPU: BASELINE configs[4]'s read set (5 Mbp genome x 1000x = 5 G windows, more than 2^32 stream positions)
whole on one device.  No oracle finishes at this size: the super-k-mer pass (2^19 buckets, 2^10 level-1 digits per XCD) must give the
histogram of the direct pass (one global table, another code path entirely) and the size-independent identities.
usage: python tools/big_one_gpu.py [--genome 5000000] [--cov 1000]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from platanus_b_amd import synth
    from platanus_b_amd.binding import GpuCounter
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=5_000_000)
    ap.add_argument("--cov", type=float, default=1000.0)
    ap.add_argument("--k", type=int, default=32)
    a = ap.parse_args()
    g = GpuCounter(0)
    g.set_option("verbose", 1)
    genome = synth.make_genome(a.genome, 1)
    pairs = int(a.genome * a.cov) // 300
    t0 = time.time()
    for lo in range(0, pairs, 250_000):
        r1, r2 = synth.make_reads(a.genome, a.cov, 1, genome=genome, pair_lo=lo, pair_hi=min(lo + 250_000, pairs))
        g.reads_append_matrix(r1)
        g.reads_append_matrix(r2)
        if (lo // 250_000) % 10 == 0:
            print(f"  appended {lo + 250_000} of {pairs} pairs, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    out = {"genome": a.genome, "cov": a.cov, "k": a.k, "pairs": pairs}
    res = {}
    for name, path in (("superk", 2), ("direct", 1)):
        g.set_option("count_path", path)
        ts = []
        for _ in range(2):
            t = time.perf_counter()
            h = g.make_kmer_read_distribution(a.k)
            ts.append((time.perf_counter() - t) * 1e3)
        h = np.asarray(h, dtype=np.uint64)
        res[name] = h
        out[name] = {"ms_wall": [round(x, 2) for x in ts], "kernels_ms": round(g.kernel_ms("count"), 3), "windows": int(g.n_instances), "distinct": int(g.n_distinct),
                     "max_occ": int(g.max_occ)}
        if path == 2:
            out[name].update(bucket_bits=int(g.kernel_ms("sk_bucket_bits")), flagged_buckets=int(g.kernel_ms("sk_flagged_buckets")),
                             spilled_records=int(g.kernel_ms("sk_spilled_records")), part1_ms=round(g.kernel_ms("part1"), 3), part2_ms=round(g.kernel_ms("part2"), 3),
                             bucket_ms=round(g.kernel_ms("bucket"), 3), took_superk_path=bool(g.kernel_ms("part1") > 0))
        print(json.dumps({name: out[name]}), file=sys.stderr, flush=True)
    out["histograms_identical"] = bool(np.array_equal(res["superk"], res["direct"]))
    h = res["superk"]
    idx = np.arange(len(h), dtype=np.uint64)
    out["sum_hist_equals_distinct"] = bool(int(h.sum()) == out["superk"]["distinct"])
    # (counts saturate at the last bin: the weighted sum is a lower bound there)
    out["weighted_sum_le_windows"] = bool(int((h * idx).sum()) <= out["superk"]["windows"])
    out["windows_identical"] = out["superk"]["windows"] == out["direct"]["windows"]
    print(json.dumps(out))
    ok = out["histograms_identical"] and out["sum_hist_equals_distinct"] and out["windows_identical"] and out["weighted_sum_le_windows"]
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
