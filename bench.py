#!/usr/bin/env python3
"""bench.py -- k-mer counting throughput of the HIP hot path (BASELINE.json metric: k-mers/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one pass of Counter::makeKmerReadDistributionMT's device equivalent (occurrence-table clear,
canonical k-mer extraction + hashing of every read window, occurrenceDistribution histogram) over this
rank's resident reads, k = 32.  Workload at N=1 = BASELINE.json configs[1]: 5 Mbp synthetic genome,
100x PE150, 0.5 % substitutions, occurrence table only.  For N>1 every rank holds a full configs[1]-sized
range of read pairs of a 100*N x set (weak scaling; at N=8 that is the shape of configs[3]).  N > 1 step
(--exchange records, the default for power-of-two N): every rank cuts its reads into run records grouped by
the owner of their minimizer bucket, ONE all_to_all over RCCL moves them, every rank counts what it received,
all-reduce of the histogram.  --exchange counts: count locally, exchange (key,count) records by stripe owner,
owner-side merge.

Reads are packed on the device before the timed region (inputs resident in HBM).  The JSON line carries
`roofline` (HBM, algorithmic bytes 12.25 B per k-mer instance at k=32: SURVEY.md 8(d)), `cpu_baseline`
(the real reference binary oracle/_ref/platanus_b on the bench workload itself, counting stage only, timed
from its own progress lines; falls back to the repo's CPU port when the binary did not travel),
`end_to_end` (the drop-in CLI on the same FASTQ files, outputs checked against the reference's md5s) and
`shard_shapes` (the counting pass on what one of the 8 GPUs of configs[3] and configs[4] holds, and the bench
workload at k = 45 and k = 75: the two-word and three-word variants of the pass beside the direct pass; the batched
gap-closing graphs of SURVEY 8(f) row 4 on synthetic gaps, checked against their CPU restatement).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # this pool's driver only supports dmabuf IPC (RCCL across processes needs it)

GENOME = 5_000_000
COVERAGE = 100.0
READ_LEN = 150
K = 32
SEED = 1
BYTES_PER_INSTANCE = 0.25 + 8 * ((K + 31) // 32) + 4        # 12.25 B at k = 32
HBM_PEAK_GBS = 8000.0                                        # MI355X_MICROARCH.md: 8 TB/s spec
TRAFFIC_PROFILE = "r03_hbm_traffic_superk.json"              # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/pmc_traffic.py)


def _sha1(path):
    import hashlib
    return hashlib.sha1(open(path, "rb").read()).hexdigest()


def _write_workload(tmp):
    """The bench workload as the two FASTQ files a user would have (same generator, same seed as the resident reads)."""
    from platanus_b_amd import synth
    synth.write_pair_files(os.path.join(tmp, "in"), GENOME, COVERAGE, SEED)
    return os.path.join(tmp, "in_1.fq"), os.path.join(tmp, "in_2.fq")


def _reference_counting_rate(tmp, f1, f2, threads, instances, tag):
    """k-mers/s of the real reference's counting stage (makeKmerReadDistributionMT, between its own progress lines assemble.cpp:306 and :334)
    on the given FASTQ pair with -t `threads`, reference default -m 16; the run is stopped behind the stage.  None if it did not get there."""
    from oracle import pyoracle
    cmd = [pyoracle.REF_BIN, "assemble", "-k", str(K), "-t", str(threads), "-m", "16", "-kmer_occ_only", "-tmp", tmp, "-o", os.path.join(tmp, "ref" + tag), "-f", f1, f2]
    p = subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
    t0 = t1 = None
    for line in p.stderr:
        if line.startswith("K = ") and "saving kmers from reads" in line:
            t0 = time.perf_counter()
        elif line.startswith("AVE_READ_LEN="):
            t1 = time.perf_counter()
            break
    p.kill()
    p.wait()
    return None if t0 is None or t1 is None else (instances / (t1 - t0), t1 - t0)


def cpu_baseline(tmp, f1, f2):
    """Counting-stage rate of the CPU reference on the bench workload itself (rank 0, N=1 only): the real reference binary
    (oracle/_ref/platanus_b, built from /root/reference by oracle/Makefile) with its defaults -m 16 and -t min(8, cores) (the reference's
    counting does not scale beyond that).  Beside it: -t 1 on a fifth of the coverage (the whole
    set would take a minute and a half) and -t min(32, cores) on the whole set."""
    from oracle import pyoracle
    ncpu = os.cpu_count() or 1
    threads = min(8, ncpu)
    n_pairs = int(GENOME * COVERAGE) // (2 * READ_LEN)
    instances = 2 * n_pairs * (READ_LEN - K + 1)
    sample = f"the bench workload: {GENOME / 1e6:g} Mbp x {COVERAGE:.0f}x PE{READ_LEN} seed {SEED} ({2 * n_pairs} reads, {instances} {K}-mers), counting stage only"
    if pyoracle.have_reference():
        r = _reference_counting_rate(tmp, f1, f2, threads, instances, "")
        if r is not None:
            out = {"value": r[0], "unit": "k-mers/s", "cores": ncpu, "threads": threads, "kind": "reference",
                   "sample": sample + f", -t {threads} -m 16 (reference defaults), {r[1]:.1f} s"}
            others = []
            try:
                from platanus_b_amd import synth
                synth.write_pair_files(os.path.join(tmp, "fifth"), GENOME, COVERAGE / 5, SEED)
                np5 = int(GENOME * COVERAGE / 5) // (2 * READ_LEN)
                r1 = _reference_counting_rate(tmp, os.path.join(tmp, "fifth_1.fq"), os.path.join(tmp, "fifth_2.fq"), 1, 2 * np5 * (READ_LEN - K + 1), "t1")
                if r1 is not None:
                    others.append({"threads": 1, "value": r1[0], "sample": f"{GENOME / 1e6:g} Mbp x {COVERAGE / 5:.0f}x (a fifth of the coverage), {r1[1]:.1f} s"})
                big = min(32, ncpu)
                if big > threads:
                    rb = _reference_counting_rate(tmp, f1, f2, big, instances, "tb")
                    if rb is not None:
                        others.append({"threads": big, "value": rb[0], "sample": f"the bench workload, {rb[1]:.1f} s"})
            except Exception as e:
                others.append({"error": str(e)})
            out["other_thread_counts"] = others
            return out
    # CPU port (oracle restatement, single thread) on a fifth of the coverage: the binary did not travel
    from platanus_b_amd import synth
    r1, r2 = synth.make_reads(GENOME, COVERAGE / 20, SEED)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    oc = pyoracle.OracleCounter()
    t0 = time.perf_counter()
    oc.count(codes.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L), K, 4 * 10 ** 9)
    dt = time.perf_counter() - t0
    return {"value": n * (L - K + 1) / dt, "unit": "k-mers/s", "cores": ncpu, "threads": 1, "kind": "port",
            "sample": f"{GENOME / 1e6:g} Mbp x {COVERAGE / 20:.0f}x PE{READ_LEN} seed {SEED}, oracle/ restatement -t 1, {dt:.1f} s"}


def end_to_end(tmp, f1, f2):
    """BASELINE.json's second metric: wall time of the drop-in CLI on the bench workload (FASTQ files in, files out, process start to
    exit): `assemble -kmer_occ_only` (configs[1]) and the full multi-k `assemble` (configs[2]); the outputs are compared with the md5
    of the real reference's files for this input (tests/golden/manifest.json, cfg3: contigs, histogram and PREFIX_kmer_occ.bin)."""
    import hashlib
    cli = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")
    if not os.path.exists(cli):
        return None
    out = {"input": f"{GENOME / 1e6:g} Mbp x {COVERAGE:.0f}x PE{READ_LEN} seed {SEED}, two FASTQ files in the page cache", "runs": 2}
    for name, extra in (("assemble_kmer_occ_only_wall_s", ["-kmer_occ_only"]), ("assemble_full_multik_wall_s", [])):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            r = subprocess.run([cli, "assemble", "-k", str(K), *extra, "-o", os.path.join(tmp, "o"), "-f", f1, f2], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": f"{name}: exit status {r.returncode}"}
            best = dt if best is None else min(best, dt)
        out[name] = best
    runs = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["cfg3"]["runs"]
    man = runs["multik"]["files"]
    md5 = {fn: hashlib.md5(open(os.path.join(tmp, fn.replace("out_", "o_", 1)), "rb").read()).hexdigest() for fn in man}
    out["outputs_identical_to_reference_md5"] = all(md5[fn] == man[fn]["md5"] for fn in man)
    # configs[1]'s own artefact: PREFIX_kmer_occ.bin of the -kmer_occ_only run above (defaults = -t 1 -m 16) against the reference's md5
    occ = runs["occ_t1"]["files"]["out_kmer_occ.bin"]
    out["kmer_occ_bin_identical_to_reference_md5"] = hashlib.md5(open(os.path.join(tmp, "o_kmer_occ.bin"), "rb").read()).hexdigest() == occ["md5"]
    out["reference_wall_s"] = {"assemble_full_multik": 1098, "assemble_kmer_occ_only_t1": 234,
                               "where": "build container, 8 cores (tests/golden/manifest.json)"}
    return out


def end_to_end_n_ranks(tmp, world):
    """BASELINE.json's second metric at N > 1: process wall time of `platanus_b assemble -gpus N` (N fresh processes forked by the CLI, rank r
    on device r, tables merged by key owner before every graph stage) on the FASTQ files of the (100 N)x set -- the files the ranks
    of this bench wrote -- beside `-gpus 1` on the same files; every output file of the two runs is compared (no reference md5 exists
    at this size: the reference needs hours for it).  On a world of one (tests) the N-rank run is two ranks sharing the device."""
    import glob
    import hashlib
    cli = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")
    if not os.path.exists(cli):
        return None
    files = sorted(glob.glob(os.path.join(tmp, "in*_[12].fq")))
    n_cli, extra = (world, []) if world > 1 else (2, ["-share_gpu"])
    out = {"input": f"{len(files)} FASTQ files, {sum(os.path.getsize(f) for f in files) / 1e9:.2f} GB, written by the ranks (page cache)",
           "command": f"platanus_b assemble -k {K} -gpus {n_cli}" + (" -share_gpu" if extra else "") + " (full multi-k)"}
    md5 = {}
    for name, tag, args in ((f"assemble_gpus{n_cli}_wall_s", "n", ["-gpus", str(n_cli), *extra]), ("assemble_gpus1_wall_s", "s", [])):
        t0 = time.perf_counter()
        try:
            r = subprocess.run([cli, "assemble", "-k", str(K), *args, "-o", os.path.join(tmp, tag), "-f", *files], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                               timeout=900)
        except subprocess.TimeoutExpired:
            out["error"] = f"{name}: no result within 900 s (stopped)"      # (the bench line must not wait for a run that does not end)
            return out
        out[name] = time.perf_counter() - t0
        if r.returncode != 0:
            out["error"] = f"{name}: exit status {r.returncode}: {r.stderr[-400:]}"
            return out
        md5[tag] = {os.path.basename(f)[len(tag) + 1:]: hashlib.md5(open(f, "rb").read()).hexdigest() for f in sorted(glob.glob(os.path.join(tmp, tag + "_*")))}
    out["output_md5"] = md5["n"]
    out["outputs_identical_to_the_single_gpu_run"] = bool(md5["n"]) and md5["n"] == md5["s"]
    return out


def shard_shapes():
    """BASELINE configs[3] and [4] at the shape ONE of their 8 GPUs sees (counting pass only, this GPU): the 1/8 read range of
    5 Mbp x 1000x and of 100 Mbp x 100x (12.5x of a 100 Mbp genome: the mostly-distinct regime before the merge).  Checked through
    size-independent properties: sum(hist) = distinct, sum(i * hist) = windows."""
    from platanus_b_amd import synth
    from platanus_b_amd.binding import GpuCounter
    out = {}
    for name, genome_bp, cov, seed in (("configs[3] shard: 5 Mbp x 1000x / 8", 5_000_000, 1000.0, SEED), ("configs[4] shard: 100 Mbp x 100x / 8", 100_000_000, 100.0, 5)):
        g = GpuCounter(0)
        genome = synth.make_genome(genome_bp, seed)
        n_pairs = int(genome_bp * cov) // (2 * READ_LEN) // 8
        for lo in range(0, n_pairs, 250_000):
            r1, r2 = synth.make_reads(genome_bp, cov, seed, genome=genome, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs))
            g.reads_append_matrix(r1)
            g.reads_append_matrix(r2)
        ms = []
        for it in range(4):
            h = g.make_kmer_read_distribution(K)
            if it:
                ms.append(g.kernel_ms("count"))
        idx = np.arange(len(h), dtype=np.uint64)
        ok = int(h.sum()) == g.n_distinct and int((idx * h).sum()) == g.n_instances and g.n_instances == 2 * n_pairs * (READ_LEN - K + 1)
        k_ms = float(np.mean(ms))
        out[name] = {"reads": 2 * n_pairs, "kmer_instances": int(g.n_instances), "distinct_kmers": int(g.n_distinct), "kernel_ms": k_ms,
                     "kmers_per_s": g.n_instances / (k_ms * 1e-3), "roofline_frac": g.n_instances * BYTES_PER_INSTANCE / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "super_kmer_pass": g.kernel_ms("part1") > 0, "bucket_bits": int(g.kernel_ms("sk_bucket_bits")),
                     "flagged_buckets": int(g.kernel_ms("sk_flagged_buckets")), "spilled_records": int(g.kernel_ms("sk_spilled_records")),
                     "identities_hold": bool(ok)}
        g.close()
        del g, genome
    # the bench workload at k = 45: two-word keys (kernels_superk2.inc) beside the direct pass, which must count the same histogram
    g = GpuCounter(0)
    genome = synth.make_genome(GENOME, SEED)
    n_pairs = int(GENOME * COVERAGE) // (2 * READ_LEN)
    for lo in range(0, n_pairs, 250_000):
        r1, r2 = synth.make_reads(GENOME, COVERAGE, SEED, genome=genome, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs))
        g.reads_append_matrix(r1)
        g.reads_append_matrix(r2)
    # ... and at k = 75 (the last k of the reference's default schedule on 150-base reads): three-word keys (kernels_superk3.inc)
    for kk, name in ((45, "configs[1] reads at k = 45 (two-word keys)"), (75, "configs[1] reads at k = 75 (three-word keys)")):
        res = {}
        for path, opt in (("super_kmer_pass", 2), ("direct_pass", 1)):
            g.set_option("count_path", opt)
            ms = []
            for it in range(3):
                h = g.make_kmer_read_distribution(kk)
                if it:
                    ms.append(g.kernel_ms("count"))
            res[path] = (np.asarray(h).copy(), float(np.mean(ms)), g.kernel_ms("part1") > 0)
        W = (kk + 31) // 32
        k_ms = res["super_kmer_pass"][1]
        out[name] = {
            "kmer_instances": int(g.n_instances), "distinct_kmers": int(g.n_distinct), "kernel_ms": k_ms, "direct_pass_kernel_ms": res["direct_pass"][1],
            "kmers_per_s": g.n_instances / (k_ms * 1e-3), "bytes_per_instance": 0.25 + 8 * W + 4,
            "roofline_frac": g.n_instances * (0.25 + 8 * W + 4) / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "super_kmer_pass": bool(res["super_kmer_pass"][2]),
            "histogram_identical_to_direct_pass": bool(np.array_equal(res["super_kmer_pass"][0], res["direct_pass"][0]))}
    g.close()
    return out


def gap_graphs_leg(n_gaps=4000):
    """SURVEY 8(f) row 4 beside the headline (N = 1, in the child process of the shard shapes): the batched gap-closing graphs on synthetic
    gaps (200-1500 bp, 20x of 150-base reads, 1 % errors), k = 32 and gap_close's own k = 20 / 80 pair's high k; the CPU restatement of the
    same two functions (oracle, one thread) is this leg's CPU baseline and its checker."""
    import time
    from oracle import pyoracle
    from platanus_b_amd.binding import GpuCounter
    rng = np.random.default_rng(9)
    seqs, first = [], [0]
    for _ in range(n_gaps):
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
    out = {"gaps": n_gaps, "sequences": len(seqs), "bases": int(off[-1]), "min_coverage": 2}
    for k in (32, 80):
        g.gap_graphs(k, 2, first[:3], off[:int(first[2]) + 1], codes)              # (module load, first allocations)
        t0 = time.perf_counter()
        got = g.gap_graphs(k, 2, first, off, codes)
        t_dev = time.perf_counter() - t0
        k_ms = g.kernel_ms("gap_graphs")
        t0 = time.perf_counter()
        want, _ = pyoracle.gap_graphs(k, 2, first, off, codes)
        t_cpu = time.perf_counter() - t0
        out[f"k{k}"] = {"nodes": int(len(want["kind"])), "device_call_s": t_dev, "device_kernels_ms": k_ms,
                        "cpu_baseline": {"kind": "port", "cores": 1, "seconds": t_cpu, "sample": "the same batch, oracle/src/gapclose.hpp"},
                        "identical_to_cpu_baseline": bool(all(np.array_equal(got[f], want[f]) for f in want))}
    g.close()
    return out


def self_launch(n):
    """Start the ranks of an N-GPU run: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py <the same arguments>` as a child process; returns its exit status."""
    import socket
    with socket.socket() as s:                   # a port nobody listens on (the rendezvous of a second bench on the same host must not collide)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    args = [x for x in sys.argv[1:] if x != "--self-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *args]
    return subprocess.run(cmd).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome", type=int, default=GENOME, help="override the workload size (development only)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline, the end-to-end CLI runs and the shard shapes (profiling runs)")
    ap.add_argument("--exchange", choices=["records", "counts"], default="records",
                    help="N > 1 step: 'records' = run records by owner, ONE all_to_all, counted on arrival (power-of-two N); 'counts' = count, "
                         "export (key,count) by owner, all_to_all, owner-side merge")
    ap.add_argument("--exchange-with-one-rank", action="store_true", help=argparse.SUPPRESS)      # tests: the N > 1 step on a world of one (RCCL to itself)
    ap.add_argument("--shard-shapes-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-graph", action="store_true", help="every pass kernel by kernel (pb_set_option graph=0): counter-collection runs of the profiler")
    ap.add_argument("--self-launch", action="store_true", help=argparse.SUPPRESS)                 # tests: take the N > 1 launch path with --gpus 1
    a = ap.parse_args()
    if a.shard_shapes_only:
        res = shard_shapes()
        try:
            res["gap_graphs (SURVEY 8(f) row 4)"] = gap_graphs_leg()
        except Exception as e:
            res["gap_graphs (SURVEY 8(f) row 4)"] = {"error": str(e)}
        print(json.dumps(res))
        return
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.self_launch):
        # `python bench.py --gpus N` as the driver runs it: this process starts the N ranks itself, as a CHILD process (never an exec)
        # and before torch is imported or anything here has touched the GPU; rank 0's JSON line passes through the inherited stdout
        # and the exit status is the child's.
        sys.exit(self_launch(a.gpus))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` (it launches its own ranks) or under "
                 f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N`")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or a.exchange_with_one_rank:
        import torch.distributed as dist
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from platanus_b_amd import synth
    from platanus_b_amd.binding import GpuCounter
    from platanus_b_amd.multigpu import count_exchanging_records, exchange_and_merge
    exchanging = world > 1 or a.exchange_with_one_rank
    by_records = exchanging and a.exchange == "records" and world & (world - 1) == 0

    # this rank's range of read pairs: rank r of a (COVERAGE*world)x set holds pairs [r*P, (r+1)*P)
    pairs_per_rank = int(a.genome * COVERAGE) // (2 * READ_LEN)
    # N > 1: the end-to-end leg (`platanus_b assemble -gpus N` on FASTQ files, after the ranks have left) needs the (100 N)x set as files;
    # every rank writes the pairs it generates anyway, in a directory rank 0 names
    e2e_dir = None
    if exchanging and not a.no_cpu_baseline:
        box = [tempfile.mkdtemp(prefix="pb_bench_") if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        e2e_dir = box[0]
    g = GpuCounter(local_rank)
    if a.no_graph:
        g.set_option("graph", 0)
    genome = synth.make_genome(a.genome, SEED)
    chunk = 250_000
    lo0 = rank * pairs_per_rank
    for lo in range(lo0, lo0 + pairs_per_rank, chunk):
        hi = min(lo + chunk, lo0 + pairs_per_rank)
        r1, r2 = synth.make_reads(a.genome, COVERAGE * world, SEED, genome=genome, pair_lo=lo, pair_hi=hi)
        g.reads_append_matrix(r1)
        g.reads_append_matrix(r2)
        if e2e_dir:
            synth.write_fastq(os.path.join(e2e_dir, f"in{rank}_1.fq"), r1, 1, lo, append=lo > lo0)
            synth.write_fastq(os.path.join(e2e_dir, f"in{rank}_2.fq"), r2, 2, lo, append=lo > lo0)
    n_reads, n_bases = g.reads_info()

    def step():
        if by_records:
            count_exchanging_records(g, dist, world, rank, K)
        else:
            g.make_kmer_read_distribution(K)
            if exchanging:
                exchange_and_merge(g, dist, world, rank)

    for _ in range(a.warmup):
        step()
    count_ms, init_ms, hist_ms = [], [], []
    stage_ms = {"part1": [], "part2": [], "bucket": [], "seg_compact": [], "compact": []}
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
        if by_records:                       # this rank's kernels: the run records of its reads + scatter / bucket kernel on what it received
            count_ms.append(g.kernel_ms("part1") + g.kernel_ms("count_records"))
            init_ms.append(0.0)
            hist_ms.append(0.0)
            for st in stage_ms:
                stage_ms[st].append(g.kernel_ms(st) if st == "part1" else 0.0)
            continue
        count_ms.append(g.kernel_ms("count"))
        init_ms.append(g.kernel_ms("table_init"))
        hist_ms.append(g.kernel_ms("hist"))
        for st in stage_ms:
            stage_ms[st].append(g.kernel_ms(st))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    replayed = (not exchanging) and g.kernel_ms("graph_replay") > 0      # the timed passes were replays of the captured pass (pb_set_option "graph")
    timed_hist, timed_max = (None, 0) if exchanging else (g.occ_hist.copy(), g.max_occ)      # the LAST TIMED pass's result (checked against the reference's md5 below)
    if replayed:
        # per-kernel durations (HIP event pairs) of three kernel-by-kernel passes AFTER the timed region: what rocprofv3's per-kernel averages
        # are to be compared with; the replay's own duration (roofline.kernel_ms) also holds the memsets, the kernel boundaries and the copy-out
        g.set_option("graph", 0)
        for st in stage_ms:
            stage_ms[st] = []
        for _ in range(3):
            g.make_kmer_read_distribution(K)
            for st in stage_ms:
                stage_ms[st].append(g.kernel_ms(st))
        g.set_option("graph", 1)
    instances = g.n_instances if world == 1 else n_reads * (READ_LEN - K + 1)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([instances], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        total_instances = int(tot.item())
    else:
        total_instances = instances

    copy_gbs = None
    if rank == 0 and world == 1:
        # measured ceiling beside the spec peak: device-to-device copy of 1 GiB, bytes read + written
        a_ = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
        b_ = torch.empty_like(a_)
        b_.copy_(a_)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b_.copy_(a_)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a_, b_
    if rank == 0:
        k_ms = float(np.mean(count_ms))        # all kernels of the counting pass (HIP events on the library's stream)
        achieved = instances * BYTES_PER_INSTANCE / (k_ms * 1e-3) / 1e9
        parts = {st: float(np.mean(v)) for st, v in stage_ms.items()}
        partitioned = parts["part1"] > 0
        kernel = ("sk_part1_kernel + sk_part2_raw16_kernel + sk_bucket2_kernel (+ sk_flagged_kernel when buckets overflow: none here) -- the super-k-mer counting pass"
                  if partitioned else "count_kernel<1> (direct global-atomic counting pass)")
        # HBM bytes per counting pass from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of this same
        # command, corrected as MI355X_MICROARCH.md prescribes).  NOT measured by this run: read from the committed profile, and only
        # quoted when the run is the profiled workload and path and the profile is not older than the kernels it describes.
        traffic = traffic_source = None
        tj = os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)
        ksrc = os.path.join(ROOT, "platanus_b_amd", "csrc", "kernels_superk.inc")
        if partitioned and a.genome == GENOME and world == 1 and os.path.exists(tj):
            prof = json.load(open(tj))
            if prof.get("kernels_sha1") == _sha1(ksrc):
                traffic = prof.get("hbm_bytes_per_counting_pass")
                traffic_source = {"file": "profiles/" + TRAFFIC_PROFILE, "kernels_sha1": prof.get("kernels_sha1"), "measured_by_this_run": False}
        # the timed loop's own result: the occurrence distribution of the last step as the reference's TSV, against the reference's md5
        hist_ok = None
        if a.genome == GENOME and world == 1 and timed_hist is not None:
            import hashlib
            tsv = "".join(f"{i}\t{int(timed_hist[i])}\n" for i in range(1, timed_max + 1))
            man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["cfg3"]["runs"]["multik"]["files"]["out_32merFrq.tsv"]
            hist_ok = hashlib.md5(tsv.encode()).hexdigest() == man["md5"]
        out = {
            "metric": "k-mers/sec", "value": total_instances * a.steps / dt, "unit": "k-mers/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{a.genome / 1e6:g} Mbp synthetic genome, {COVERAGE * world:.0f}x PE{READ_LEN}, k={K}, "
                                   f"occurrence table only (BASELINE configs[1] per GPU)",
                       "reads_per_gpu": int(n_reads), "kmer_instances_per_gpu": int(instances), "distinct_kmers": int(g.n_distinct),
                       "parallelism": f"reads range-partitioned over {world} GPU(s)" + ("" if world == 1 else ", run records exchanged by bucket owner over RCCL "
                                      "(one all_to_all), counted on arrival" if by_records else ", (key,count) records exchanged by stripe owner over RCCL, owner-side merge"),
                       "torch_distributed": ({"backend": dist.get_backend(), "world_size": dist.get_world_size()} if dist is not None else None),
                       "histogram_identical_to_reference_md5": hist_ok,
                       "inherited_PB_environment": {k_: v for k_, v in os.environ.items() if k_.startswith("PB_")}},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": k_ms, "measured_d2d_copy_GBs": copy_gbs,
                         "algorithmic_bytes_per_launch": instances * BYTES_PER_INSTANCE,
                         "launch": ("one hipGraph replay per pass (3 memsets, sk_part1, sk_part2, sk_bucket, scalars + histogram head copied out): kernel_ms = HIP events "
                                    "around the replay on the library's stream; kernels_ms = the three kernels' event pairs, mean of three kernel-by-kernel passes after the timed region"
                                    if replayed else "kernel by kernel: kernel_ms = sum of the kernels' HIP event pairs on the library's stream"),
                         "sum_of_kernels_ms": parts["part1"] + parts["part2"] + parts["bucket"] + parts["seg_compact"],
                         "kernels_ms": {"part1": parts["part1"], "part2": parts["part2"], "bucket_count": parts["bucket"],
                                        "flagged_buckets": parts["seg_compact"], "table_init": float(np.mean(init_ms)),
                                        "table_to_arrays": parts["compact"], "hist": float(np.mean(hist_ms))}},
        }
        if world == 1 and not exchanging and not a.no_cpu_baseline:
            g.close()                        # the legs below bring their own contexts (and the reference needs the host's memory)
            tmp = tempfile.mkdtemp(prefix="pb_bench_")
            try:
                f1, f2 = _write_workload(tmp)
                try:
                    out["cpu_baseline"] = cpu_baseline(tmp, f1, f2)
                except Exception as e:          # the baseline is a reported number, never a reason to lose the bench line
                    out["cpu_baseline"] = {"value": None, "unit": "k-mers/s", "cores": os.cpu_count() or 0, "threads": 0, "kind": "port", "sample": f"failed: {e}"}
                try:
                    out["end_to_end"] = end_to_end(tmp, f1, f2)
                except Exception as e:
                    out["end_to_end"] = {"error": str(e)}
            finally:
                subprocess.run(["rm", "-rf", tmp])
            try:
                # in a process of its own: the kernel statistics of THIS process then hold the timed workload only (rocprofv3 writes one
                # summary per process, and profiles/ quotes the average launch duration of the counting kernels from it)
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--shard-shapes-only"], capture_output=True, text=True, timeout=900)
                out["shard_shapes"] = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": r.stderr[-500:]}
            except Exception as e:
                out["shard_shapes"] = {"error": str(e)}
    if dist is not None:
        # the ranks give the devices back and leave the process group BEFORE the end-to-end leg starts its own processes
        g.close()
        torch.cuda.synchronize()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if e2e_dir:
            try:
                out["end_to_end"] = end_to_end_n_ranks(e2e_dir, world)
            except Exception as e:
                out["end_to_end"] = {"error": str(e)}
            finally:
                subprocess.run(["rm", "-rf", e2e_dir])
        print(json.dumps(out))


if __name__ == "__main__":
    main()
