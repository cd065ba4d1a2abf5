"""Quick GPU-vs-oracle parity sweep over key widths (development aid; the real tests live in tests/)."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platanus_b_amd import synth
from platanus_b_amd.binding import GpuCounter
from oracle.pyoracle import OracleCounter

r1, r2 = synth.make_reads(50_000, 50, seed=1, n_rate=0.001)
codes = np.concatenate([r1, r2]); n, L = codes.shape
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
g0 = GpuCounter(0); print(g0.device_info())
for k in (32, 21, 42, 75, 108, 150):
    oc = OracleCounter(); oc.count(codes.reshape(-1), off, k)
    oh, omx = oc.occ_hist()
    g = GpuCounter(0); g.reads_append_matrix(r1); g.reads_append_matrix(r2)
    gh = g.make_kmer_read_distribution(k)
    ok_h = np.array_equal(gh, oh) and g.max_occ == omx
    ok_r = True
    for cut in (1, 3):
        okeys, ocnt = oc.retain(cut)
        g.retain(cut); gk, gc = g.fetch_retained()
        ok_r &= np.array_equal(gk, okeys) and np.array_equal(gc, ocnt)
    print(k, "hist", ok_h, "retain", ok_r, "inst", g.n_instances, "distinct", g.n_distinct, "count_ms", g.kernel_ms("count"),
          "sort_ms", g.kernel_ms("sort"), flush=True)
    assert ok_h and ok_r
print("ALL OK")
