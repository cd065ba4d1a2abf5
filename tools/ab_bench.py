"""Development aid: stage times of the counting pass on BASELINE configs[1] for one or more builds of the library.
usage: python tools/ab_bench.py [--genome N] [--cov C] [--check] lib1.so [lib2.so ...]     (each library in its own process)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(lib, genome, cov, check, opts):
    import numpy as np
    from platanus_b_amd import binding, synth
    binding.LIB_PATH = os.path.abspath(lib)
    g = binding.GpuCounter(0)
    for name, v in opts:
        g.set_option(name, v)
    gen = synth.make_genome(genome, 1)
    n_pairs = int(genome * cov) // 300
    for lo in range(0, n_pairs, 250_000):
        r1, r2 = synth.make_reads(genome, cov, 1, genome=gen, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs))
        g.reads_append_matrix(r1)
        g.reads_append_matrix(r2)
    acc = {}
    for it in range(8):
        h = g.make_kmer_read_distribution(32)
        if it >= 2:
            for st in ("part1", "part2", "bucket", "seg_compact", "count"):
                acc.setdefault(st, []).append(g.kernel_ms(st))
    out = {st: round(float(np.mean(v)), 4) for st, v in acc.items()}
    out.update(lib=lib, bits=g.kernel_ms("sk_bucket_bits"), flagged=g.kernel_ms("sk_flagged_buckets"), spilled=g.kernel_ms("sk_spilled_records"),
               rec_per_1k=round(g.kernel_ms("sk_records_per_1k_windows"), 2), distinct=g.n_distinct, windows=g.n_instances)
    if check:
        g.set_option("count_path", 1)
        h2 = g.make_kmer_read_distribution(32)
        out["hist_equals_direct"] = bool(np.array_equal(h, h2))
    print(json.dumps(out), flush=True)


def main():
    a = sys.argv[1:]
    genome, cov, check, opts = 5_000_000, 100.0, False, []
    libs = []
    while a:
        x = a.pop(0)
        if x == "--genome": genome = int(a.pop(0))
        elif x == "--cov": cov = float(a.pop(0))
        elif x == "--check": check = True
        elif x == "--opt": n, v = a.pop(0).split("="); opts.append((n, int(v)))
        elif x == "--one": return one(a.pop(0), genome, cov, check, opts)
        else: libs.append(x)
    for lib in libs:
        cmd = [sys.executable, __file__, "--genome", str(genome), "--cov", str(cov)] + (["--check"] if check else [])
        for n, v in opts:
            cmd += ["--opt", f"{n}={v}"]
        subprocess.run(cmd + ["--one", lib], timeout=600)


if __name__ == "__main__":
    main()
