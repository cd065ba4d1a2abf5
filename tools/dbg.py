import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from platanus_b_amd import synth
from platanus_b_amd.binding import GpuCounter
from oracle.pyoracle import OracleCounter
r1, r2 = synth.make_reads(50_000, 50, seed=1)
codes = np.concatenate([r1, r2]); k = 42
n, L = codes.shape
off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
oc = OracleCounter(); oc.count(codes.reshape(-1), off, k)
okeys, ocnt = oc.retain(1)
g = GpuCounter(0); g.reads_append_matrix(codes)
g.make_kmer_read_distribution(k)
ak, ac = g.fetch_all()
so = dict(zip(map(tuple, okeys.tolist()), ocnt.tolist()))
tk = list(map(tuple, ak.tolist()))
from collections import Counter as PC
pc = PC(tk)
dups = sum(1 for v in pc.values() if v > 1)
only_gpu = [t for t in pc if t not in so]
print("table: n", len(tk), "unique", len(pc), "dup keys", dups, "only gpu", len(only_gpu), "oracle n", len(so), "n_distinct", g.n_distinct)
w0 = set(int(x) for x in okeys[:, 0]); w1 = set(int(x) for x in okeys[:, 1])
mix = sum(1 for t in only_gpu if t[0] in w0 and t[1] in w1)
print("only-gpu that are mixtures of valid words:", mix)
for t in only_gpu[:5]: print([hex(x) for x in t])
# count totals
tot = {}
for t, c in zip(tk, ac.tolist()): tot[t] = tot.get(t, 0) + c
bad = sum(1 for t in so if tot.get(t, 0) != so[t])
print("keys whose summed count differs:", bad, "sum all", sum(tot.values()), "instances", g.n_instances)
g.retain(1); gk, gc = g.fetch_retained()
st = list(map(tuple, gk.tolist()))
print("sorted output equals sorted(table)?", sorted(tk, key=lambda t: (t[1], t[0])) == st)
