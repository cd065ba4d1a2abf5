#!/usr/bin/env python3
"""bench.py -- k-mer counting throughput of the HIP hot path (BASELINE.json metric: k-mers/sec).

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one pass of Counter::makeKmerReadDistributionMT's device equivalent (occurrence-table clear,
canonical k-mer extraction + hashing of every read window, occurrenceDistribution histogram) over this
rank's resident reads, k = 32.  Workload at N=1 = BASELINE.json configs[1]: 5 Mbp synthetic genome,
100x PE150, 0.5 % substitutions, occurrence table only.  For N>1 every rank holds a full configs[1]-sized
range of read pairs of a 100*N x set (weak scaling; at N=8 that is the shape of configs[3]) and each step
ends with the owner exchange of (key,count) records over RCCL (torch.distributed all_to_all) + owner-side
merge + all-reduce of the histogram.

Reads are packed on the device before the timed region (inputs resident in HBM).  The JSON line carries
`roofline` (HBM, algorithmic bytes 12.25 B per k-mer instance at k=32: SURVEY.md 8(d)) and `cpu_baseline`
(the real reference binary oracle/_ref/platanus_b on a bounded sample, counting stage only, timed from its
own progress lines; falls back to the repo's CPU port when the binary did not travel).
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


def cpu_baseline(sample_cov=20.0, threads=None):
    """Counting-stage rate of the CPU reference on a bounded sample of the same workload (rank 0, N=1 only)."""
    from platanus_b_amd import synth
    from oracle import pyoracle
    ncpu = os.cpu_count() or 1
    threads = threads or min(8, ncpu)
    # same genome as the GPU workload (same table pressure from genomic k-mers), one fifth of its coverage
    n_pairs = int(GENOME * sample_cov) // (2 * READ_LEN)
    instances = 2 * n_pairs * (READ_LEN - K + 1)
    sample = f"{GENOME / 1e6:g} Mbp x {sample_cov:.0f}x PE{READ_LEN} seed {SEED} ({2 * n_pairs} reads, {instances} {K}-mers), counting stage only"
    if pyoracle.have_reference():
        tmp = tempfile.mkdtemp(prefix="pb_cpu_")
        try:
            synth.write_pair_files(os.path.join(tmp, "s"), GENOME, sample_cov, SEED)
            cmd = [pyoracle.REF_BIN, "assemble", "-k", str(K), "-t", str(threads), "-m", "16", "-kmer_occ_only", "-tmp", tmp,
                   "-o", os.path.join(tmp, "o"), "-f", os.path.join(tmp, "s_1.fq"), os.path.join(tmp, "s_2.fq")]
            p = subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True)
            t0 = t1 = None
            for line in p.stderr:          # makeKmerReadDistributionMT runs between these two progress lines
                if line.startswith("K = ") and "saving kmers from reads" in line:
                    t0 = time.perf_counter()      # assemble.cpp:306
                elif line.startswith("AVE_READ_LEN="):
                    t1 = time.perf_counter()      # assemble.cpp:334
                    break
            p.kill()
            p.wait()
            # the device build of the same CLI on the same sample files: end-to-end wall time of the full multi-k assemble
            # (FASTQ parse + upload + every k round + FASTA write); reported next to the baseline, not part of `value`
            gpu_cli = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")
            extra = None
            if os.path.exists(gpu_cli):
                g0 = time.perf_counter()
                r = subprocess.run([gpu_cli, "assemble", "-k", str(K), "-o", os.path.join(tmp, "g"), "-f", os.path.join(tmp, "s_1.fq"),
                                    os.path.join(tmp, "s_2.fq")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                if r.returncode == 0:
                    extra = {"wall_s": time.perf_counter() - g0, "what": "platanus_b_amd/bin/platanus_b assemble -k 32 (full multi-k, FASTQ in, "
                             "contig.fa out) on the cpu_baseline sample files"}
            if t0 is not None and t1 is not None:
                out = {"value": instances / (t1 - t0), "unit": "k-mers/s", "cores": threads, "kind": "reference",
                       "sample": sample + f", -t {threads} -m 16 (reference defaults), {t1 - t0:.1f} s"}
                if extra:
                    out["gpu_assemble_same_sample"] = extra
                return out
        finally:
            subprocess.run(["rm", "-rf", tmp])
    # CPU port (oracle restatement, single thread)
    r1, r2 = synth.make_reads(GENOME, sample_cov / 4, SEED)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    oc = pyoracle.OracleCounter()
    t0 = time.perf_counter()
    oc.count(codes.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L), K, 4 * 10 ** 9)
    dt = time.perf_counter() - t0
    return {"value": n * (L - K + 1) / dt, "unit": "k-mers/s", "cores": 1, "kind": "port",
            "sample": f"{GENOME / 1e6:g} Mbp x {sample_cov / 4:.0f}x PE{READ_LEN} seed {SEED}, oracle/ restatement -t 1, {dt:.1f} s"}


def end_to_end():
    """BASELINE.json's second metric: wall time of the drop-in CLI on the bench workload (FASTQ files in, files out, process start to
    exit): `assemble -kmer_occ_only` (configs[1]) and the full multi-k `assemble` (configs[2]); the outputs of the latter are compared
    with the md5 of the real reference's files for this input (tests/golden/manifest.json, cfg3)."""
    import hashlib
    from platanus_b_amd import synth
    cli = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")
    if not os.path.exists(cli):
        return None
    tmp = tempfile.mkdtemp(prefix="pb_e2e_")
    try:
        synth.write_pair_files(os.path.join(tmp, "in"), GENOME, COVERAGE, SEED)
        out = {"input": f"{GENOME / 1e6:g} Mbp x {COVERAGE:.0f}x PE{READ_LEN} seed {SEED}, two FASTQ files in the page cache", "runs": 2}
        for name, extra in (("assemble_kmer_occ_only_wall_s", ["-kmer_occ_only"]), ("assemble_full_multik_wall_s", [])):
            best = None
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([cli, "assemble", "-k", str(K), *extra, "-o", os.path.join(tmp, "o"), "-f", os.path.join(tmp, "in_1.fq"),
                                    os.path.join(tmp, "in_2.fq")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                dt = time.perf_counter() - t0
                if r.returncode != 0:
                    return {"error": f"{name}: exit status {r.returncode}"}
                best = dt if best is None else min(best, dt)
            out[name] = best
        man = json.load(open(os.path.join(ROOT, "tests", "golden", "manifest.json")))["cfg3"]["runs"]["multik"]["files"]
        md5 = {fn: hashlib.md5(open(os.path.join(tmp, fn.replace("out_", "o_", 1)), "rb").read()).hexdigest() for fn in man}
        out["outputs_identical_to_reference_md5"] = all(md5[fn] == man[fn]["md5"] for fn in man)
        out["reference_wall_s"] = {"assemble_full_multik": 1098, "where": "build container, 8 cores, -t 8 (tests/golden/manifest.json)"}
        return out
    finally:
        subprocess.run(["rm", "-rf", tmp])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome", type=int, default=GENOME, help="override the workload size (development only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from platanus_b_amd import synth
    from platanus_b_amd.binding import GpuCounter
    from platanus_b_amd.multigpu import exchange_and_merge

    # this rank's range of read pairs: rank r of a (COVERAGE*world)x set holds pairs [r*P, (r+1)*P)
    pairs_per_rank = int(a.genome * COVERAGE) // (2 * READ_LEN)
    g = GpuCounter(local_rank)
    genome = synth.make_genome(a.genome, SEED)
    chunk = 250_000
    lo0 = rank * pairs_per_rank
    for lo in range(lo0, lo0 + pairs_per_rank, chunk):
        hi = min(lo + chunk, lo0 + pairs_per_rank)
        r1, r2 = synth.make_reads(a.genome, COVERAGE * world, SEED, genome=genome, pair_lo=lo, pair_hi=hi)
        g.reads_append_matrix(r1)
        g.reads_append_matrix(r2)
    n_reads, n_bases = g.reads_info()

    def step():
        g.make_kmer_read_distribution(K)
        if world > 1:
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
        count_ms.append(g.kernel_ms("count"))
        init_ms.append(g.kernel_ms("table_init"))
        hist_ms.append(g.kernel_ms("hist"))
        for st in stage_ms:
            stage_ms[st].append(g.kernel_ms(st))
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
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
        kernel = ("sk_part1_kernel + sk_part2_raw12_kernel + sk_bucket_kernel + sk_flagged_kernel (super-k-mer counting pass)"
                  if partitioned else "count_kernel<1> (direct global-atomic counting pass)")
        # HBM bytes per counting pass from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes
        # of this same command, corrected as MI355X_MICROARCH.md prescribes): profiles/r01_hbm_traffic.json.  Only quoted
        # when the run is the profiled workload and path.
        traffic = None
        tj = os.path.join(ROOT, "profiles", "r01_hbm_traffic_superk.json")
        if partitioned and a.genome == GENOME and world == 1 and os.path.exists(tj):
            traffic = json.load(open(tj)).get("hbm_bytes_per_counting_pass")
        out = {
            "metric": "k-mers/sec", "value": total_instances * a.steps / dt, "unit": "k-mers/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{a.genome / 1e6:g} Mbp synthetic genome, {COVERAGE * world:.0f}x PE{READ_LEN}, k={K}, "
                                   f"occurrence table only (BASELINE configs[1] per GPU)",
                       "reads_per_gpu": int(n_reads), "kmer_instances_per_gpu": int(instances), "distinct_kmers": int(g.n_distinct),
                       "parallelism": f"reads range-partitioned over {world} GPU(s)" + (", owner exchange over RCCL" if world > 1 else "")},
            "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel_ms": k_ms, "measured_d2d_copy_GBs": copy_gbs,
                         "algorithmic_bytes_per_launch": instances * BYTES_PER_INSTANCE,
                         "kernels_ms": {"part1": parts["part1"], "part2": parts["part2"], "bucket_count": parts["bucket"],
                                        "flagged_buckets": parts["seg_compact"], "table_init": float(np.mean(init_ms)),
                                        "table_to_arrays": parts["compact"], "hist": float(np.mean(hist_ms))}},
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:          # the baseline is a reported number, never a reason to lose the bench line
                out["cpu_baseline"] = {"value": None, "unit": "k-mers/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
            try:
                out["end_to_end"] = end_to_end()
            except Exception as e:
                out["end_to_end"] = {"error": str(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
