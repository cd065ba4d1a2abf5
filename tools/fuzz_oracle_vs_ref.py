"""Development aid (CPU, build container only): the oracle CLI against the REAL reference binary (oracle/_ref/platanus_b, built by
oracle/Makefile from /root/reference) on random small read sets with random `assemble` options, including the ones the golden
fixtures do not vary (-c -d -K -a -e).  usage: python tools/fuzz_oracle_vs_ref.py [n_cases] [seed]"""
import filecmp
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC = os.path.join(ROOT, "oracle", "pb_oracle")
REF = os.path.join(ROOT, "oracle", "_ref", "platanus_b")


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    asc = np.frombuffer(b"ACGTN", dtype=np.uint8)
    bad = 0
    for case in range(n_cases):
        G = int(rng.integers(3_000, 25_000))
        g = rng.integers(0, 4, G).astype(np.uint8)
        for _ in range(int(rng.integers(0, 5))):
            L = int(rng.integers(40, 500))
            a, b = int(rng.integers(0, G - L)), int(rng.integers(0, G - L))
            g[b:b + L] = g[a:a + L] if rng.random() < 0.5 else (3 - g[a:a + L])[::-1]
        cov, err, RL = float(rng.choice([20, 40, 60])), float(rng.choice([0.0, 0.004, 0.012])), int(rng.choice([100, 150]))
        n_reads = int(G * cov) // RL
        with tempfile.TemporaryDirectory() as td:
            with open(os.path.join(td, "in.fq"), "wb") as f:
                for i in range(n_reads):
                    s = int(rng.integers(0, G - RL + 1))
                    r = g[s:s + RL].copy()
                    if rng.random() < 0.5:
                        r = (3 - r)[::-1].copy()
                    e = rng.random(RL) < err
                    r[e] = (r[e] + 1 + rng.integers(0, 3, int(e.sum()))) & 3
                    f.write(b"@r%d\n" % i + asc[r].tobytes() + b"\n+\n" + b"I" * RL + b"\n")
            args = ["-k", str(int(rng.choice([21, 25, 32, 33, 45]))), "-s", str(int(rng.choice([0, 5, 10, 20]))), "-u", str(rng.choice(["0", "1"])), "-t", "1", "-m", "1"]
            for opt, vals, p in (("-n", ["1", "2", "3"], 0.3), ("-c", ["1", "2", "3", "4"], 0.4), ("-d", ["0.3", "0.5", "0.8"], 0.4), ("-K", ["0.3", "0.5", "0.7"], 0.4),
                                 ("-a", ["5", "10", "20"], 0.3), ("-e", ["20", "45"], 0.3)):
                if rng.random() < p:
                    args += [opt, str(rng.choice(vals))]
            res = {}
            for who, b in (("oracle", ORC), ("reference", REF)):
                d = os.path.join(td, who)
                os.mkdir(d)
                p = subprocess.run([b, "assemble", *args, "-o", "out", "-f", "../in.fq"], cwd=d, capture_output=True, text=True, timeout=900)
                res[who] = (p.returncode, d)
            ok = res["oracle"][0] == res["reference"][0]
            if ok and res["reference"][0] == 0:
                for fn in os.listdir(res["reference"][1]):
                    f2 = os.path.join(res["oracle"][1], fn)
                    ok = ok and os.path.exists(f2) and filecmp.cmp(os.path.join(res["reference"][1], fn), f2, shallow=False)
            bad += not ok
            print(f"{'ok' if ok else 'MISMATCH'} case {case}: G={G} cov={cov} err={err} RL={RL} rc={res['reference'][0]} args={' '.join(args)}", flush=True)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
