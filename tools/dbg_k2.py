"""Development aid: the two-word super-k-mer pass against the direct pass on one input (verbose), usage: python tools/dbg_k2.py k [genome] [cov]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platanus_b_amd import binding, synth
k = int(sys.argv[1]); G = int(sys.argv[2]) if len(sys.argv) > 2 else 60_000; cov = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
gen = synth.make_genome(G, 1); n_pairs = int(G * cov) // 300
res = {}
for name, path in (("superk", 2), ("direct", 1)):
    g = binding.GpuCounter(0); g.set_option("verbose", 1); g.set_option("count_path", path)
    for lo in range(0, n_pairs, 250_000):
        r1, r2 = synth.make_reads(G, cov, 1, genome=gen, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs), n_rate=0.002)
        g.reads_append_matrix(r1); g.reads_append_matrix(r2)
    for _ in range(3):
        h = np.asarray(g.make_kmer_read_distribution(k))
    res[name] = (h, g.n_instances, g.n_distinct)
    print(name, "windows", g.n_instances, "distinct", g.n_distinct, "part1/part2/bucket/count ms", g.kernel_ms("part1"), g.kernel_ms("part2"), g.kernel_ms("bucket"), g.kernel_ms("count"), flush=True)
print("identical histograms:", bool(np.array_equal(res["superk"][0], res["direct"][0])), "first bins", res["superk"][0][:6], res["direct"][0][:6])
