"""Development aid: differential test of the whole `assemble` CLI (GPU build) against the REAL reference binary (oracle/_ref/platanus_b,
compiled from /root/reference by oracle/Makefile; it travels to the GPU box) -- or, where that binary is absent, against the oracle's
CLI, which shares the node-level host headers with the product -- on random small inputs with repeats, inverted repeats and tandem
repeats (junctions, hairpins, cycles in the graph) and random options; a third of the cases run the GPU side as 2 or 3 ranks.
usage: python tools/fuzz_cli.py [n_cases] [seed] [min_genome] [max_genome]     (>= ~110 kbp at 40x reaches the super-k-mer pass)"""
import filecmp
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GPU = os.environ.get("PB_GPU_BIN") or os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")     # (PB_GPU_BIN: e.g. an AddressSanitizer build of the host)
ORC = os.path.join(ROOT, "oracle", "pb_oracle")
REF = os.path.join(ROOT, "oracle", "_ref", "platanus_b")


def keep_case(td, case, what="fuzz_cli"):
    """A failed case leaves its inputs, command lines and outputs under gpurun_out/ (merged back from the GPU box)."""
    import shutil
    dst = os.path.join(ROOT, "gpurun_out", "failed_cases", f"{what}_{os.getpid()}_{case}")
    shutil.copytree(td, dst, dirs_exist_ok=True)
    print(f"  case kept in {dst}", flush=True)


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    g_lo = int(sys.argv[3]) if len(sys.argv) > 3 else 3_000
    g_hi = int(sys.argv[4]) if len(sys.argv) > 4 else 40_000
    asc = np.frombuffer(b"ACGTN", dtype=np.uint8)
    bad = 0
    checker = REF if os.path.exists(REF) else ORC
    print(f"checker: {'reference' if checker == REF else 'oracle'} ({checker})", flush=True)
    for case in range(n_cases):
        G = int(rng.integers(g_lo, g_hi))
        g = rng.integers(0, 4, G).astype(np.uint8)
        for _ in range(int(rng.integers(0, 6))):               # repeats: copy, inverted copy, tandem
            L = int(rng.integers(40, 600))
            a, b = int(rng.integers(0, G - L)), int(rng.integers(0, G - L))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                g[b:b + L] = g[a:a + L]
            elif kind == 1:
                g[b:b + L] = (3 - g[a:a + L])[::-1]
            else:
                unit = g[a:a + int(rng.integers(2, 40))]
                g[b:b + L] = np.resize(unit, L)
        cov = float(rng.choice([15, 30, 60]))
        err = float(rng.choice([0.0, 0.003, 0.01]))
        RL = int(rng.choice([100, 125, 150]))
        n_pairs = int(G * cov) // (2 * RL)
        with tempfile.TemporaryDirectory() as td:
            for mate in (1, 2):
                with open(os.path.join(td, f"in_{mate}.fq"), "wb") as f:
                    for i in range(n_pairs):
                        fl = int(np.clip(rng.normal(350, 30), RL, G))
                        s = int(rng.integers(0, G - fl + 1))
                        frag = g[s:s + fl]
                        r = frag[:RL].copy() if mate == 1 else (3 - frag[-RL:])[::-1].copy()
                        e = rng.random(RL) < err
                        r[e] = (r[e] + 1 + rng.integers(0, 3, int(e.sum()))) & 3
                        if rng.random() < 0.02:
                            r[int(rng.integers(0, RL))] = 4
                        f.write(b"@r%d/%d\n" % (i, mate) + asc[r].tobytes() + b"\n+\n" + b"I" * RL + b"\n")
            k = int(rng.choice([21, 25, 32, 33, 40, 45]))
            args = ["-k", str(k), "-s", str(int(rng.choice([0, 5, 10, 20]))), "-u", str(rng.choice(["0", "0", "1"])), "-m", "1"]
            if rng.random() < 0.3:
                args += ["-n", str(int(rng.integers(1, 4)))]
            if rng.random() < 0.25:                              # the options of the k schedule and of the cleaning steps
                args += ["-c", str(int(rng.integers(1, 5)))]
            if rng.random() < 0.25:
                args += ["-d", str(rng.choice(["0.3", "0.5", "0.8"]))]
            if rng.random() < 0.25:
                args += ["-K", str(rng.choice(["0.3", "0.5", "0.7"]))]
            if rng.random() < 0.2:
                args += ["-a", str(rng.choice(["5", "10", "20"]))]
            if rng.random() < 0.2:
                args += ["-e", str(rng.choice(["20", "45"]))]
            outs = {}
            signalled = False
            ranks = int(rng.choice([1, 1, 2, 3]))
            gpu_extra = ["-gpus", str(ranks), "-share_gpu"] if ranks > 1 else []
            # device code paths that these small inputs would not take by themselves (same results): the super-k-mer pass for k0 <= 32
            # -- with N ranks that is the records-before-counting exchange --, odd bucket counts, the other N-rank design
            path = int(rng.integers(0, 4))
            if path == 1:
                gpu_extra += ["-gpu_opt", "count_path=2"]
            elif path == 2:
                gpu_extra += ["-gpu_opt", "count_path=2", "-gpu_opt", f"sk_bucket_bits={int(rng.choice([4, 9, 12, 19]))}"]
            elif path == 3 and ranks > 1:
                gpu_extra += ["-gpu_opt", "dist_design=1"]
            for name, binary, extra in (("orc", checker, ["-t", "1"]), ("gpu", GPU, gpu_extra)):
                d = os.path.join(td, name)
                os.mkdir(d)
                cmd = [binary, "assemble", *args, *extra, "-o", "out", "-f", "../in_1.fq", "../in_2.fq"]
                p = subprocess.run(cmd, cwd=d, capture_output=True, text=True, timeout=600)
                if p.returncode < 0:                  # killed by a signal: a failed case, whichever binary it was; the evidence is kept
                    signalled = True
                    print(f"  SIGNAL {-p.returncode} from {os.path.basename(binary)} {' '.join(cmd[1:])}; its stderr ended:\n    " +
                          p.stderr[-3000:].replace("\n", "\n    "), flush=True)
                    with open(os.path.join(d, "argv_and_stderr.txt"), "w") as f:
                        f.write(" ".join(cmd) + f"\nexit: signal {-p.returncode}\n" + p.stderr)
                outs[name] = (p.returncode, d)
            same = outs["orc"][0] == outs["gpu"][0] and not signalled
            if same and outs["orc"][0] == 0:
                for fn in sorted(os.listdir(outs["orc"][1])):
                    if not fn.startswith("out_"):
                        continue
                    a, b = os.path.join(outs["orc"][1], fn), os.path.join(outs["gpu"][1], fn)
                    if not os.path.exists(b) or not filecmp.cmp(a, b, shallow=False):
                        same = False
                        print(f"  differs: {fn}")
            if not same:
                bad += 1
                keep_case(td, case)
                print(f"MISMATCH case {case}: G={G} cov={cov} err={err} RL={RL} args={' '.join(args)} rc={outs['orc'][0]}/{outs['gpu'][0]}", flush=True)
            else:
                nct = sum(1 for l in open(os.path.join(outs["gpu"][1], "out_contig.fa")) if l.startswith(">")) if outs["gpu"][0] == 0 else -1
                print(f"case {case}: ok (G={G} cov={cov} err={err} args={' '.join(args)} ranks={ranks} gpu_opts={' '.join(gpu_extra)} rc={outs['gpu'][0]} records={nct})", flush=True)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
