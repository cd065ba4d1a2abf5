"""Two ranks sharing GPU 0 (gloo for the collectives, staged through the host): the product's whole N > 1 step --
count this rank's range of read pairs on the device, export by owner, exchange, owner-side merge, histogram all-reduce --
for tests/test_gpu_parity.py::test_two_ranks_share_one_gpu."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platanus_b_amd import synth  # noqa: E402
from platanus_b_amd.binding import GpuCounter  # noqa: E402
from platanus_b_amd.multigpu import count_exchanging_records, exchange_and_merge, owner_of  # noqa: E402


def main():
    k, genome, cov, seed, out = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    design = sys.argv[6] if len(sys.argv) > 6 else "counts"          # "counts": count, export, exchange (key,count) records, merge; "records": exchange run records, then count
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    n_pairs = int(genome * cov) // 300
    per = (n_pairs + world - 1) // world
    r1, r2 = synth.make_reads(genome, cov, seed, err=0.02, pair_lo=rank * per, pair_hi=min((rank + 1) * per, n_pairs))
    g = GpuCounter(0)
    g.set_option("count_path", 2)       # the super-k-mer pass also on this small input
    g.reads_append_matrix(r1)
    g.reads_append_matrix(r2)
    if design == "records":
        ghist = count_exchanging_records(g, dist, world, rank, k)
        took_superk, merged_lds = g.kernel_ms("part1") > 0, g.kernel_ms("count_records") > 0
        n_local = g.retain(1)
        keys, counts = g.fetch_retained()
    else:
        g.make_kmer_read_distribution(k)
        took_superk = g.kernel_ms("part1") > 0
        ghist = exchange_and_merge(g, dist, world, rank)
        merged_lds = g.kernel_ms("merge") > 0
        n_local = g.retain(1)
        keys, counts = g.fetch_retained()
        assert (owner_of(keys.reshape(n_local, -1)[:, 0], world) == rank).all()
    np.save(out + f".keys{rank}.npy", keys)
    np.save(out + f".counts{rank}.npy", counts)
    dist.barrier()
    if rank == 0:
        json.dump({"hist_nonzero": {int(i): int(ghist[i]) for i in np.nonzero(ghist)[0]}, "n_distinct": int(g.n_distinct),
                   "superk": bool(took_superk), "merged_lds": bool(merged_lds)}, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
