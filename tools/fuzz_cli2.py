"""Development aid: differential test of the `kmer_divide` and `merge` commands (GPU build vs the REAL reference binary
oracle/_ref/platanus_b where it is present, else the oracle's CLI) on the contigs and the k-mer occurrence table of random
heterozygous read sets, with random options.
usage: python tools/fuzz_cli2.py [n_cases] [seed]"""
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


SIGNALLED = []


def run(binary, args, cwd):
    r = subprocess.run([binary, *args], cwd=cwd, capture_output=True, text=True, timeout=600)
    if r.returncode < 0:              # killed by a signal: the case fails, whichever binary it was; the crash handler's frames are in stderr
        SIGNALLED.append((binary, args))
        print(f"  SIGNAL {-r.returncode} from {os.path.basename(binary)} {' '.join(args)}; its stderr ended:\n    " + r.stderr[-3000:].replace("\n", "\n    "), flush=True)
        with open(os.path.join(cwd, "argv_and_stderr.txt"), "a") as f:
            f.write(" ".join([binary, *args]) + f"\nexit: signal {-r.returncode}\n" + r.stderr + "\n")
    return r


def main():
    from platanus_b_amd import synth
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    checker = REF if os.path.exists(REF) else ORC
    print(f"checker: {'reference' if checker == REF else 'oracle'} ({checker})", flush=True)
    for case in range(n_cases):
        G = int(rng.integers(8_000, 60_000))
        with tempfile.TemporaryDirectory() as td:
            synth.write_pair_files(os.path.join(td, "in"), G, float(rng.choice([30, 60])), int(rng.integers(1, 1 << 30)), err=float(rng.choice([0.002, 0.01])),
                                   n_rate=0.0, het=float(rng.choice([0.0, 0.004, 0.01])))
            assert run(GPU, ["assemble", "-k", "32", "-o", "asm", "-f", "in_1.fq", "in_2.fq"], td).returncode == 0
            assert run(GPU, ["assemble", "-k", "32", "-n", "1", "-kmer_occ_only", "-o", "occ", "-f", "in_1.fq", "in_2.fq"], td).returncode == 0
            kd = [[], ["-mask"], ["-r", str(round(float(rng.uniform(0.05, 0.9)), 2))], ["-mask", "-R", str(round(float(rng.uniform(1.1, 3.0)), 2))]][int(rng.integers(0, 4))]
            mg = ["-k", str(round(float(rng.uniform(0.2, 1.0)), 2)), "-l", str(round(float(rng.uniform(0.5, 2.0)), 2)), "-u", str(rng.choice(["0", "1"])), "-m", "1"]
            ok = True
            n_sig = len(SIGNALLED)
            for cmd, args, inputs in (("kmer_divide", kd, ["-k", "../occ_kmer_occ.bin", "-f", "../asm_contig.fa"]), ("merge", mg, ["-f", "../asm_contig.fa"])):
                dirs = {}
                for name, binary in (("orc", checker), ("gpu", GPU)):
                    d = os.path.join(td, f"{cmd}_{name}")
                    os.mkdir(d)
                    r = run(binary, [cmd, *args, "-o", "out", *inputs], d)
                    dirs[name] = (r.returncode, d, r.stderr[-1500:])
                rep = int(os.environ.get("FUZZ_REPEAT", "0"))        # (development: the same GPU command again and again, exit codes tallied)
                if rep:
                    tally = {}
                    for _ in range(rep):
                        d2 = tempfile.mkdtemp(dir=td)
                        r = run(GPU, [cmd, *args, "-o", "out", *inputs], d2)
                        tally[r.returncode] = tally.get(r.returncode, 0) + 1
                        if r.returncode != 0:
                            print("   rc", r.returncode, "stderr:", r.stderr[-800:].replace("\n", " | "), flush=True)
                    print(f"  {cmd} x{rep}: exit codes {tally}", flush=True)
                if dirs["orc"][0] != dirs["gpu"][0]:
                    ok = False
                    print(f"  {cmd}: rc {dirs['orc'][0]} vs {dirs['gpu'][0]}")
                    print("  stderr of the GPU build:", dirs["gpu"][2].replace("\n", "\n    "))
                    continue
                for fn in sorted(os.listdir(dirs["orc"][1])):
                    b = os.path.join(dirs["gpu"][1], fn)
                    if not os.path.exists(b) or not filecmp.cmp(os.path.join(dirs["orc"][1], fn), b, shallow=False):
                        ok = False
                        print(f"  {cmd}: {fn} differs")
            if len(SIGNALLED) != n_sig:
                ok = False
            if not ok:
                bad += 1
                from fuzz_cli import keep_case
                keep_case(td, case, "fuzz_cli2")
                print(f"MISMATCH case {case}: G={G} kmer_divide {' '.join(kd)} merge {' '.join(mg)}", flush=True)
            else:
                print(f"case {case}: ok (G={G} kmer_divide {' '.join(kd)}; merge {' '.join(mg)})", flush=True)
    print(f"{n_cases} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
