"""Development aid: cost of the two N > 1 step designs without the network, on one GPU (world size 1 over RCCL), bench workload:
counting pass alone; counting pass + exchange of (key,count) records + owner-side merge; exchange of run records, then the counting."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from platanus_b_amd import synth
    from platanus_b_amd.binding import GpuCounter
    from platanus_b_amd.multigpu import count_exchanging_records, exchange_and_merge
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    genome_len = 5_000_000
    g = GpuCounter(0)
    genome = synth.make_genome(genome_len, 1)
    pairs = int(genome_len * 100.0) // 300
    for lo in range(0, pairs, 250_000):
        r1, r2 = synth.make_reads(genome_len, 100.0, 1, genome=genome, pair_lo=lo, pair_hi=min(lo + 250_000, pairs))
        g.reads_append_matrix(r1); g.reads_append_matrix(r2)
    for mode in ("count", "counts", "records", "count", "counts", "records"):
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "records":
                count_exchanging_records(g, dist, 1, 0, 32)
            else:
                g.make_kmer_read_distribution(32)
                if mode == "counts":
                    exchange_and_merge(g, dist, 1, 0)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        label = {"count": "count only                          ", "counts": "count + exchange of (key,count) + merge", "records": "exchange of run records, then count   "}[mode]
        extra = f"  (part1 {g.kernel_ms('part1'):.2f} + scatter / bucket {g.kernel_ms('count_records'):.2f} ms of kernels)" if mode == "records" else ""
        print(label, " ".join(f"{t:7.2f}" for t in ts), "ms" + extra, flush=True)
    laps = {}
    for _ in range(4):
        count_exchanging_records(g, dist, 1, 0, 32, laps=laps)
    print("sections of the records design (ms per step, a device synchronisation after each):", {k: round(v / 4, 3) for k, v in laps.items()}, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
