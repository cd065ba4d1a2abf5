"""Development aid: stage times of the two-word super-k-mer pass for several builds (usage: python tools/dbg_k2t.py k lib.so ...)"""
import sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--one":
    from platanus_b_amd import binding, synth
    k = int(sys.argv[2]); binding.LIB_PATH = os.path.abspath(sys.argv[3])
    G, cov = 5_000_000, 100.0
    gen = synth.make_genome(G, 1); n_pairs = int(G * cov) // 300
    g = binding.GpuCounter(0); g.set_option("count_path", 2)
    if os.environ.get("BITS"): g.set_option("sk_bucket_bits", int(os.environ["BITS"]))
    for lo in range(0, n_pairs, 250_000):
        r1, r2 = synth.make_reads(G, cov, 1, genome=gen, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs))
        g.reads_append_matrix(r1); g.reads_append_matrix(r2)
    for _ in range(3):
        try:
            g.make_kmer_read_distribution(k)
        except Exception as e:
            pass
    print(sys.argv[3], "part1 %.3f part2 %.3f bucket %.3f flagged-path %.3f count %.3f | bits %d flagged %d spilled %d" % (g.kernel_ms("part1"), g.kernel_ms("part2"), g.kernel_ms("bucket"), g.kernel_ms("seg_compact"), g.kernel_ms("count"), g.kernel_ms("sk_bucket_bits"), g.kernel_ms("sk_flagged_buckets"), g.kernel_ms("sk_spilled_records")), flush=True)
else:
    for lib in sys.argv[2:]:
        subprocess.run([sys.executable, __file__, "--one", sys.argv[1], lib])
