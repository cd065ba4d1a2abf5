import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platanus_b_amd import binding, synth
binding.LIB_PATH = os.path.abspath(sys.argv[1])
g = binding.GpuCounter(0); g.set_option("verbose", 1)
G, cov = 5_000_000, 100.0
gen = synth.make_genome(G, 1); n_pairs = int(G * cov) // 300
for lo in range(0, n_pairs, 250_000):
    r1, r2 = synth.make_reads(G, cov, 1, genome=gen, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs))
    g.reads_append_matrix(r1); g.reads_append_matrix(r2)
g.make_kmer_read_distribution(32)
print("bucket ms", g.kernel_ms("bucket"))
