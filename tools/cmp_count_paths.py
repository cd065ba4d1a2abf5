"""development: the two forms of the two-word bucket kernel against each other (33 <= k <= 64), or the three-word super-k-mer pass
against the direct pass (65 <= k <= 96), on a set large enough for many buckets per block.  usage: python tools/cmp_count_paths.py [genome bp] [k]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platanus_b_amd import synth
from platanus_b_amd.binding import GpuCounter
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 45
g = GpuCounter(0)
gen = synth.make_genome(G, 1)
n_pairs = G * 100 // 300
for lo in range(0, n_pairs, 250_000):
    r1, r2 = synth.make_reads(G, 100.0, 1, genome=gen, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs))
    g.reads_append_matrix(r1); g.reads_append_matrix(r2)
def ms(name):
    try:
        return round(g.kernel_ms(name), 3)
    except Exception:
        return None


res = {}
for form in (1, 2):
    if K <= 64:
        g.set_option("count_path", 2)
        g.set_option("sk_bucket_form", form)
    else:
        g.set_option("count_path", form)                  # 1 = direct pass, 2 = super-k-mer pass
    h = g.make_kmer_read_distribution(K)
    nd = g.n_distinct
    g.retain(1)
    k, c = g.fetch_retained()
    res[form] = (h.copy(), nd, k.copy(), c.copy())
    print("form" if K <= 64 else "count_path", form, "bits", ms("sk_bucket_bits"), "kernel ms", ms("count"), "part1/part2/bucket", ms("part1"), ms("part2"), ms("bucket"), "distinct", nd, "retained", len(c), "instances", g.n_instances, "sum counts", int(c.astype(np.uint64).sum()), flush=True)
h1, n1, k1, c1 = res[1]
h2, n2, k2, c2 = res[2]
print("hist equal", np.array_equal(h1, h2), "keys equal", k1.shape == k2.shape and np.array_equal(k1, k2), "counts equal", c1.shape == c2.shape and np.array_equal(c1, c2))
d = np.nonzero(h1 != h2)[0]
print("bins that differ", d[:20], [(int(h1[i]), int(h2[i])) for i in d[:20]])
if k1.shape == k2.shape and np.array_equal(k1, k2):
    bad = np.nonzero(c1 != c2)[0]
    print("keys with other counts", len(bad))
    for i in bad[:10]:
        print(" ", [hex(int(x)) for x in k1[i]], int(c1[i]), int(c2[i]))
else:
    v1 = {tuple(int(x) for x in r) for r in k1.reshape(len(c1), -1)} if len(c1) < 5_000_000 else None
    if v1 is not None:
        v2 = {tuple(int(x) for x in r) for r in k2.reshape(len(c2), -1)}
        print("only in form 1:", len(v1 - v2), [tuple(hex(x) for x in t) for t in list(v1 - v2)[:5]])
        print("only in form 2:", len(v2 - v1), [tuple(hex(x) for x in t) for t in list(v2 - v1)[:5]])
