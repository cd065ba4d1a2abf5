"""Development aid: differential test of pb_merge_records -- the partition + LDS-table merge of (key,count) records against the
global-table merge (pb_set_option "merge_table") on random record lists: duplicates, zero counts, heavy hitters, k = 1..32.
usage: python tools/fuzz_merge.py [n_cases] [seed]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from platanus_b_amd.binding import GpuCounter
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for case in range(n_cases):
        k = int(rng.integers(8, 33))
        n_keys = int(rng.integers(200_000, 3_000_000))
        space = 1 << (2 * k) if k < 32 else 1 << 63
        keys = rng.integers(0, space, n_keys, dtype=np.uint64) if k < 32 else rng.integers(0, 1 << 63, n_keys, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n_keys, dtype=np.uint64)
        copies = rng.choice([1, 1, 1, 2, 3, 8], n_keys)
        rk = np.repeat(keys, copies)
        rc = rng.integers(0, 50, len(rk)).astype(np.int64)
        rc[rng.random(len(rk)) < 0.05] = 0                                    # zero counts are legal and ignored
        if rng.random() < 0.5:                                                 # a heavy hitter: one key in thousands of records
            hot = keys[0]
            rk = np.concatenate([rk, np.full(5000, hot, dtype=np.uint64)])
            rc = np.concatenate([rc, rng.integers(1, 30, 5000)])
        perm = rng.permutation(len(rk))
        rk, rc = rk[perm], rc[perm]
        if len(rk) < (1 << 20):
            continue
        tk = torch.from_numpy(rk.view(np.int64)).cuda()
        tc = torch.from_numpy(rc.astype(np.int32)).cuda()
        res = []
        for table in (False, True):
            g = GpuCounter(0)
            g.set_option("merge_table", 1 if table else 0)
            hist = np.zeros(65535, dtype=np.uint64)
            mx, nd = C.c_uint64(), C.c_uint64()
            torch.cuda.synchronize()
            g._chk(g.lib.pb_merge_records(g.h, k, C.c_void_p(tk.data_ptr()), C.c_void_p(tc.data_ptr()), len(rk), hist.ctypes.data, C.byref(mx), C.byref(nd)))
            used_lds = g.kernel_ms("merge") > 0 and not table
            g.k = k
            g.retain(1)
            gk, gc = g.fetch_retained()
            res.append((hist.copy(), gk.copy(), gc.copy(), nd.value, used_lds))
            g.close()
        a, b = res
        ok = np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3]
        if not ok:
            bad += 1
            print(f"MISMATCH case {case}: k={k} records={len(rk)} distinct {a[3]} vs {b[3]}", flush=True)
        else:
            print(f"case {case}: ok (k={k} records={len(rk)} distinct={a[3]} lds_path={a[4]})", flush=True)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
