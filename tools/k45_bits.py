"""development: the two-word pass (k = 45) on the bench workload at several bucket counts, both forms of its bucket kernel.
usage: python tools/k45_bits.py [--k K] [--forms 1,2] [--blocks N] [bits ...]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platanus_b_amd import synth
from platanus_b_amd import binding
from platanus_b_amd.binding import GpuCounter
if os.environ.get("PB_LIB"):
    binding.LIB_PATH = os.path.abspath(os.environ["PB_LIB"])       # (a development build of the library)
g = GpuCounter(0)
gen = synth.make_genome(5_000_000, 1)
n_pairs = 5_000_000 * 100 // 300
for lo in range(0, n_pairs, 250_000):
    r1, r2 = synth.make_reads(5_000_000, 100.0, 1, genome=gen, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs))
    g.reads_append_matrix(r1); g.reads_append_matrix(r2)
g.set_option("count_path", 2)
ref = None
args = sys.argv[1:]
K = int(args.pop(args.index("--k") + 1)) if "--k" in args else 45
if "--k" in args: args.remove("--k")
forms = [int(x) for x in args.pop(args.index("--forms") + 1).split(",")] if "--forms" in args else [1, 2]
if "--forms" in args: args.remove("--forms")
if "--blocks" in args:
    g.set_option("sk_bucket_blocks", int(args.pop(args.index("--blocks") + 1))); args.remove("--blocks")
for form, bits in [(f, int(x)) for x in (args or ["0", "16", "17", "18"]) for f in forms]:
    g.set_option("sk_bucket_bits", bits)
    g.set_option("sk_bucket_form", form)
    ms = []
    for it in range(4):
        h = g.make_kmer_read_distribution(K)
        if it: ms.append((g.kernel_ms("part1"), g.kernel_ms("part2"), g.kernel_ms("bucket"), g.kernel_ms("count")))
    if ref is None: ref = h.copy()
    m = np.mean(ms, axis=0)
    print(f"k {K} form {form} bits {bits} (used {int(g.kernel_ms('sk_bucket_bits'))}): part1 {m[0]:.3f} part2 {m[1]:.3f} bucket {m[2]:.3f} count {m[3]:.3f} ms; flagged {int(g.kernel_ms('sk_flagged_buckets'))} spilled {int(g.kernel_ms('sk_spilled_records'))} same histogram {bool(np.array_equal(h, ref))}", flush=True)
