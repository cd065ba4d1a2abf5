"""development / measurement: the batched gap-closing graphs (pb_gap_graphs_build + _fetch) on a synthetic scaffold's worth of gaps, beside
the CPU restatement (oracle, one thread) and -- where oracle/_ref/gapclose_ref travelled -- the REFERENCE's own two functions gap by gap.
usage (GPU box): python tools/gap_graphs_time.py [n_gaps] [k] [min_cov]      -> one JSON line"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from oracle import pyoracle
    from platanus_b_amd.binding import GpuCounter
    n_gaps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    mc = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    rng = np.random.default_rng(9)
    seqs, first = [], [0]
    for _ in range(n_gaps):                                   # a gap of 200..1500 bp, 20x of 150-base reads with 1 % errors
        G = int(rng.integers(200, 1500))
        genome = rng.integers(0, 4, G).astype(np.uint8)
        for _ in range(max(1, G * 20 // 150)):
            L = min(150, G)
            s = int(rng.integers(0, G - L + 1))
            r = genome[s:s + L].copy()
            e = rng.random(L) < 0.01
            r[e] = (r[e] + 1) & 3
            seqs.append(r)
        first.append(len(seqs))
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    codes = np.concatenate(seqs)
    first = np.array(first, dtype=np.uint64)
    g = GpuCounter(0)
    g.gap_graphs(k, mc, first[:3], off[:int(first[2]) + 1], codes)          # warm-up (module load)
    t0 = time.perf_counter()
    got = g.gap_graphs(k, mc, first, off, codes)
    t_dev = time.perf_counter() - t0
    k_ms = g.kernel_ms("gap_graphs")
    t0 = time.perf_counter()
    want, _ = pyoracle.gap_graphs(k, mc, first, off, codes)
    t_cpu = time.perf_counter() - t0
    same = all(np.array_equal(got[f], want[f]) for f in want)
    out = {"gaps": n_gaps, "k": k, "min_coverage": mc, "sequences": len(seqs), "bases": int(off[-1]), "nodes": int(len(want["kind"])),
           "device_call_s": t_dev, "device_kernels_ms": k_ms, "cpu_restatement_s_one_thread": t_cpu, "identical_to_the_cpu_restatement": bool(same)}
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "gapclose_ref")):
        sub = min(n_gaps, 4000)                               # (a bounded share for the reference: it also prints its tables)
        t0 = time.perf_counter()
        pyoracle.gap_graphs_reference(k, mc, first[:sub + 1], off[:int(first[sub]) + 1], codes)
        out["reference_functions_s"] = {"gaps": sub, "wall_s_incl_process_start_and_table_dump": time.perf_counter() - t0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
