"""Development aid: time the counting pass with parts of its kernels removed (PB_ABLATE), or with other env switches.

usage: python tools/ablate.py [--genome N] [--lib path/to/variant.so] NAME=VALUE[,NAME=VALUE...] ...
Every argument is one configuration (comma-separated env assignments, "-" = none); each is run 3 times on the
bench workload and the per-stage kernel times of the last run are printed.  Results under PB_ABLATE are invalid.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    from platanus_b_amd import synth
    from platanus_b_amd.binding import GpuCounter
    args = sys.argv[1:]
    genome_len = 5_000_000
    while args and args[0] in ("--genome", "--lib"):
        if args[0] == "--genome":
            genome_len = int(args[1])
        else:                               # a variant build of the library (development only)
            from platanus_b_amd import binding
            binding.LIB_PATH = os.path.abspath(args[1])
            print("library", binding.LIB_PATH, flush=True)
        args = args[2:]
    g = GpuCounter(0)
    genome = synth.make_genome(genome_len, 1)
    pairs = int(genome_len * 100.0) // 300
    for lo in range(0, pairs, 250_000):
        r1, r2 = synth.make_reads(genome_len, 100.0, 1, genome=genome, pair_lo=lo, pair_hi=min(lo + 250_000, pairs))
        g.reads_append_matrix(r1); g.reads_append_matrix(r2)
    g.make_kmer_read_distribution(32)
    print("instances", g.n_instances, "distinct", g.n_distinct, flush=True)
    stages = ["part1", "part2", "bucket", "seg_compact", "count"]
    for cfg in args or ["-"]:
        sets = [] if cfg == "-" else [kv.split("=", 1) for kv in cfg.split(",")]
        for k, v in sets:
            os.environ[k] = v
        for _ in range(3):
            t0 = time.perf_counter()
            g.make_kmer_read_distribution(32)
            wall = (time.perf_counter() - t0) * 1e3
        extra = " ".join(f"{s[3:]}={g.kernel_ms(s):g}" for s in ("sk_bucket_bits", "sk_flagged_buckets", "sk_records_per_1k_windows"))
        print(f"{cfg:40s} " + " ".join(f"{s}={g.kernel_ms(s):7.3f}" for s in stages) + f" wall={wall:7.2f} distinct={g.n_distinct} {extra}", flush=True)
        for k, _ in sets:
            del os.environ[k]


if __name__ == "__main__":
    main()
