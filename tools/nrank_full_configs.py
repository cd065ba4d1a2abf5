"""BASELINE configs[3] / configs[4] as configs: `platanus_b assemble -gpus N` (N = 4 and 6: this pool admits at most 6 processes on a box's GPU, so 8 ranks cannot share it) on the FULL read set (5 Mbp x 1000x, or 100 Mbp x 100x with
-m 256) against `assemble -gpus 1` on the same FASTQ files -- every output file compared byte for byte -- on ONE GPU that the N ranks share
(-share_gpu: records and tables through shared host memory; the driver's 8-GPU node runs the RCCL path).  Both designs of the k0
counting step (dist_design 0: run records exchanged before counting; 1: count, exchange (key,count), owner-side merge).

usage (GPU box): python tools/nrank_full_configs.py --config 3 [--ranks 8] [--workers 16] [--out gpurun_out/r03_configs3_nrank.json]
The reads come from platanus_b_amd/synth.py (seeded), written as 2 x `parts` FASTQ files by `workers` processes."""
import argparse
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CLI = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")
CONFIGS = {3: dict(genome=5_000_000, cov=1000.0, seed=1, extra=[], name="configs[3]: 5 Mbp x 1000x PE150, full multi-k assemble"),
           4: dict(genome=100_000_000, cov=100.0, seed=5, extra=["-m", "256"], name="configs[4]: 100 Mbp x 100x PE150, -m 256, full multi-k assemble")}


def write_part(args):
    td, genome_len, cov, seed, part, lo, hi = args
    from platanus_b_amd import synth
    genome = synth.make_genome(genome_len, seed)
    first = True
    for a in range(lo, hi, 200_000):
        r1, r2 = synth.make_reads(genome_len, cov, seed, genome=genome, pair_lo=a, pair_hi=min(a + 200_000, hi))
        synth.write_fastq(os.path.join(td, f"in{part:02d}_1.fq"), r1, 1, a, append=not first)
        synth.write_fastq(os.path.join(td, f"in{part:02d}_2.fq"), r2, 2, a, append=not first)
        first = False
    return part


def run(tag, args, files, td):
    t0 = time.perf_counter()
    p = subprocess.run([CLI, "assemble", "-k", "32", *args, "-o", os.path.join(td, tag), "-f", *files], capture_output=True, text=True,
                       env=dict(os.environ, PB_TIMING="1"))
    wall = time.perf_counter() - t0
    stages = {}
    m = re.search(r"#### STAGE WALL TIMES \(s\) ####\n(.*?)\n\(device context", p.stderr, re.S)
    if m:
        for ln in m.group(1).splitlines():
            if "\t" in ln:
                k, v = ln.rsplit("\t", 1)
                try:
                    stages[k.strip()] = float(v)
                except ValueError:
                    pass
    outs = {}
    for fn in sorted(os.listdir(td)):
        if fn.startswith(tag + "_"):
            h = hashlib.md5()
            with open(os.path.join(td, fn), "rb") as f:
                for blk in iter(lambda: f.read(1 << 24), b""):
                    h.update(blk)
            outs[fn[len(tag) + 1:]] = {"md5": h.hexdigest(), "bytes": os.path.getsize(os.path.join(td, fn))}
    return {"argv": " ".join(["platanus_b assemble -k 32", *args]), "exit_status": p.returncode, "wall_s": wall, "stage_wall_s": stages, "outputs": outs,
            "stderr_tail": p.stderr[-1500:] if p.returncode else None, "pbgpu_lines": [ln for ln in p.stderr.splitlines() if ln.startswith("[pbgpu]")][:60]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, choices=[3, 4], required=True)
    ap.add_argument("--ranks", type=int, default=4, help="ranks of the two N-rank runs (a one-GPU box of this pool admits at most 6 processes on its GPU)")
    ap.add_argument("--ranks2", type=int, default=6, help="ranks of a third run (not a power of two: counted per rank, (key,count) tables merged); 0 = none")
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--scale", type=float, default=1.0, help="fraction of the coverage (development: small runs of the same script)")
    ap.add_argument("--out", default=None)
    ap.add_argument("--only-g1", action="store_true", help="development: the one-GPU run only, stderr kept (with --extra \"-gpu_opt verbose=1\")")
    ap.add_argument("--extra", default="", help="extra CLI arguments for every run")
    a = ap.parse_args()
    c = CONFIGS[a.config]
    cov = c["cov"] * a.scale
    n_pairs = int(c["genome"] * cov) // 300
    res = {"config": c["name"] + ("" if a.scale == 1.0 else f" (coverage x {a.scale})"), "genome_bp": c["genome"], "coverage": cov, "read_pairs": n_pairs, "seed": c["seed"],
           "ranks": a.ranks, "ranks_third_run": a.ranks2, "placement": "all ranks on one MI355X (-share_gpu), exchange through shared host memory"}
    with tempfile.TemporaryDirectory(prefix="pb_cfg_", dir=os.environ.get("PB_BIG_TMP")) as td:
        parts = a.workers
        t0 = time.perf_counter()
        jobs = [(td, c["genome"], cov, c["seed"], p, n_pairs * p // parts, n_pairs * (p + 1) // parts) for p in range(parts)]
        with ProcessPoolExecutor(a.workers) as ex:
            list(ex.map(write_part, jobs))
        files = sorted(os.path.join(td, f) for f in os.listdir(td) if f.endswith(".fq"))
        res["fastq"] = {"files": len(files), "bytes": sum(os.path.getsize(f) for f in files), "written_in_s": time.perf_counter() - t0}
        print("reads written:", res["fastq"], flush=True)
        res["runs"] = {}
        plan = [("g1", []), ("n_records", ["-gpus", str(a.ranks), "-share_gpu"]), ("n_counts", ["-gpus", str(a.ranks), "-share_gpu", "-gpu_opt", "dist_design=1"])]
        if a.ranks2:
            plan.append(("n2_counts", ["-gpus", str(a.ranks2), "-share_gpu"]))
        if a.only_g1:
            plan = plan[:1]
        for tag, args in plan:
            res["runs"][tag] = run(tag, [*c["extra"], *a.extra.split(), *args], files, td)
            print(tag, "exit", res["runs"][tag]["exit_status"], "wall", round(res["runs"][tag]["wall_s"], 2), flush=True)
        ref = res["runs"]["g1"]["outputs"]
        nr = [t for t, _ in plan if t != "g1"]
        for tag in nr:
            r = res["runs"][tag]
            r["outputs_identical_to_the_1_gpu_run"] = r["exit_status"] == 0 and bool(ref) and r["outputs"] == ref
        res["all_identical"] = all(res["runs"][t].get("outputs_identical_to_the_1_gpu_run") for t in nr) and res["runs"]["g1"]["exit_status"] == 0
    s = json.dumps(res, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(s + "\n")
    print(s)
    sys.exit(0 if res["all_identical"] else 1)


if __name__ == "__main__":
    main()
