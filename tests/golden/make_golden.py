"""Regenerate the golden fixtures from the REAL reference binary (oracle/_ref/platanus_b, built by
oracle/Makefile from /root/reference).  Runs only where /root/reference exists; the fixtures it writes
are plain data (inputs are regenerated from the seeded generator, outputs are the reference's files).

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from platanus_b_amd import synth  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "platanus_b")

# name -> generator parameters + the reference command lines whose outputs are kept
CASES = {
    "g5k": dict(genome=5000, cov=30, seed=3, err=0.01, n_rate=0.002,
                runs={"multik": ["-k", "32", "-t", "1", "-m", "1"],
                      "occ_n2": ["-k", "32", "-t", "1", "-m", "1", "-n", "2", "-kmer_occ_only"],
                      "k21_s0": ["-k", "21", "-s", "0", "-t", "1", "-m", "1"]}),
    "c1": dict(genome=50_000, cov=50, seed=1, err=0.005, n_rate=0.0,
               runs={"multik": ["-k", "32", "-t", "1", "-m", "1"],
                     "occ": ["-k", "32", "-t", "1", "-m", "1", "-kmer_occ_only"],
                     "s0": ["-k", "32", "-s", "0", "-t", "1", "-m", "1"],
                     "u1": ["-k", "32", "-t", "1", "-m", "1", "-u", "1"]}),
    "c2": dict(genome=200_000, cov=40, seed=7, err=0.01, n_rate=0.001,
               runs={"multik_n2": ["-k", "32", "-t", "1", "-m", "1", "-n", "2"]}),
    "k45": dict(genome=20_000, cov=40, seed=11, err=0.005, n_rate=0.0,
                runs={"multik_k45": ["-k", "45", "-t", "1", "-m", "1"]}),
    # two haplotypes differing at 0.4 % of the positions: bubbles, crushed with -u 1 (only -u >= 1 reaches crushBubble, Q1)
    "het": dict(genome=30_000, cov=60, seed=13, err=0.003, n_rate=0.0, het=0.004,
                runs={"u1": ["-k", "32", "-t", "1", "-m", "1", "-u", "1"], "u0": ["-k", "32", "-t", "1", "-m", "1"],
                      "occ_n1": ["-k", "32", "-t", "1", "-m", "1", "-n", "1", "-kmer_occ_only"]},
                # `kmer_divide` on the u0 contigs with the occ_n1 table
                kmer_divide={"kd_default": [], "kd_mask": ["-mask"], "kd_r05": ["-r", "0.5"]},
                # `merge` of the u0 contigs: k = 150 (5 words), 75 (3 words), 30 (1 word)
                merge={"mg_k150": ["-m", "1"], "mg_k75": ["-m", "1", "-k", "0.5", "-l", "1.0"], "mg_k30_u1": ["-m", "1", "-k", "0.2", "-l", "0.5", "-u", "1"]}),
}
KEEP_WHOLE_BELOW = 400_000      # bytes; larger outputs are pinned by md5 only

# "cfg3" in manifest.json (BASELINE configs[2]) was produced once by hand with the same generator and binary:
#   python -m platanus_b_amd.synth /tmp/cfg2 --genome 5000000 --cov 100 --seed 1
#   oracle/_ref/platanus_b assemble -k 32 -t 8 -m 16 -o ref -f /tmp/cfg2_1.fq /tmp/cfg2_2.fq      (1098 s, 8 cores)            -> runs.multik
#   oracle/_ref/platanus_b assemble -k 32 -t 1 -m 16 -kmer_occ_only -o occ -f /tmp/cfg2_1.fq /tmp/cfg2_2.fq   (234 s, VmPeak 8.0 GB) -> runs.occ_t1


def md5(path):
    h = hashlib.md5()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/platanus_b missing: run `make -C oracle ref` where /root/reference exists")
    manifest = {}
    mpath = os.path.join(HERE, "manifest.json")
    if os.path.exists(mpath):            # keep entries this script does not regenerate (cfg3: an 18-minute reference run)
        manifest = {k: v for k, v in json.load(open(mpath)).items() if k not in CASES}
    for name, c in CASES.items():
        tmp = tempfile.mkdtemp(prefix="golden_")
        try:
            synth.write_pair_files(os.path.join(tmp, "in"), c["genome"], c["cov"], c["seed"], err=c["err"], n_rate=c["n_rate"], het=c.get("het", 0.0))
            manifest[name] = {k: c[k] for k in ("genome", "cov", "seed", "err", "n_rate")}
            manifest[name]["het"] = c.get("het", 0.0)
            manifest[name]["runs"] = {}
            for run, args in c["runs"].items():
                d = os.path.join(tmp, run)
                os.makedirs(d)
                subprocess.run([REF, "assemble", *args, "-o", "out", "-f", "../in_1.fq", "../in_2.fq"], cwd=d, check=True,
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                files = {}
                for fn in sorted(os.listdir(d)):
                    p = os.path.join(d, fn)
                    files[fn] = {"md5": md5(p), "bytes": os.path.getsize(p)}
                    if os.path.getsize(p) <= KEEP_WHOLE_BELOW:
                        shutil.copy(p, os.path.join(HERE, f"{name}.{run}.{fn}"))
                manifest[name]["runs"][run] = {"args": args, "files": files}
                print(name, run, {k: v["bytes"] for k, v in files.items()})
            for run, args in c.get("kmer_divide", {}).items():
                d = os.path.join(tmp, run)
                os.makedirs(d)
                subprocess.run([REF, "kmer_divide", "-k", "../occ_n1/out_kmer_occ.bin", "-f", "../u0/out_contig.fa", "-o", "out", *args], cwd=d,
                               check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                files = {}
                for fn in sorted(os.listdir(d)):
                    p = os.path.join(d, fn)
                    files[fn] = {"md5": md5(p), "bytes": os.path.getsize(p)}
                    if os.path.getsize(p) <= KEEP_WHOLE_BELOW:
                        shutil.copy(p, os.path.join(HERE, f"{name}.{run}.{fn}"))
                manifest[name].setdefault("kmer_divide", {})[run] = {"args": args, "files": files}
                print(name, run, {k: v["bytes"] for k, v in files.items()})
            for run, args in c.get("merge", {}).items():
                d = os.path.join(tmp, run)
                os.makedirs(d)
                subprocess.run([REF, "merge", "-f", "../u0/out_contig.fa", "-o", "out", *args], cwd=d, check=True, stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL)
                files = {}
                for fn in sorted(os.listdir(d)):
                    p = os.path.join(d, fn)
                    files[fn] = {"md5": md5(p), "bytes": os.path.getsize(p)}
                    if os.path.getsize(p) <= KEEP_WHOLE_BELOW:
                        shutil.copy(p, os.path.join(HERE, f"{name}.{run}.{fn}"))
                manifest[name].setdefault("merge", {})[run] = {"args": args, "files": files}
                print(name, run, {k: v["bytes"] for k, v in files.items()})
        finally:
            shutil.rmtree(tmp)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
