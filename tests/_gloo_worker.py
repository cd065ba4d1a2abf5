"""world_size-2 worker for tests/test_multigpu_gloo.py (gloo, CPU tensors).  Local counting and the owner-side
merge are done by the oracle / numpy here (there is no GPU); the exchange itself is the product code
platanus_b_amd.multigpu.exchange_records + owner_of."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from platanus_b_amd import synth  # noqa: E402
from platanus_b_amd.multigpu import exchange_records, exchange_records_split, gather_retained, owner_of  # noqa: E402
from oracle.pyoracle import OracleCounter  # noqa: E402


def main():
    k = int(sys.argv[1])
    out = sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    W = (k + 31) // 32
    genome, cov, seed = 20_000, 30, 9
    n_pairs = int(genome * cov) // 300
    per = (n_pairs + world - 1) // world
    r1, r2 = synth.make_reads(genome, cov, seed, n_rate=0.001, pair_lo=rank * per, pair_hi=min((rank + 1) * per, n_pairs))
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    oc = OracleCounter()
    oc.count(codes.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L), k)
    keys, counts = oc.retain(1)
    own = owner_of(keys[:, 0], world)
    mode = sys.argv[3] if len(sys.argv) > 3 else "split"
    if mode == "v1":                                    # every record with a 32-bit count
        order = np.argsort(own, kind="stable")
        keys, counts, own = keys[order], counts[order], own[order]
        send = [int((own == r).sum()) for r in range(world)]
        tk = torch.from_numpy(keys.reshape(-1).view(np.int64).copy())
        tc = torch.from_numpy(counts.astype(np.int32))
        rk, rc, recv = exchange_records(dist, torch, tk, tc, send, W)
    else:                                               # count-1 records as bare keys, the others with 16-bit counts (what pb_export_split_* lays out)
        order = np.lexsort((counts > 1, own))           # by owner, singles before multis
        keys, counts, own = keys[order], counts[order], own[order]
        n_single = [int(((own == r) & (counts == 1)).sum()) for r in range(world)]
        n_multi = [int(((own == r) & (counts > 1)).sum()) for r in range(world)]
        tk = torch.from_numpy(keys.reshape(-1).view(np.int64).copy())
        c16 = np.minimum(counts[counts > 1], 65535).astype("<u2")
        tc = torch.from_numpy(c16.view(np.uint8).copy())
        rk, rc, recv = exchange_records_split(dist, torch, tk, tc, n_single, n_multi, W)
    rkeys = rk.numpy().view(np.uint64).reshape(-1, W)
    assert (owner_of(rkeys[:, 0], world) == rank).all()
    # owner-side merge: add counts of equal keys, saturate at 65534 (counter.h:468-470)
    merged = {}
    for key, c in zip(map(tuple, rkeys.tolist()), rc.tolist()):
        merged[key] = min(merged.get(key, 0) + c, 65534)
    hist = np.zeros(65535, dtype=np.int64)
    for c in merged.values():
        hist[c] += 1
    th = torch.from_numpy(hist)
    dist.all_reduce(th)
    # the retained part (count >= 2) of every owner's shard, gathered and sorted on every rank: what the graph stage is given
    mk = np.array([kk for kk, c in merged.items() if c >= 2], dtype=np.uint64).reshape(-1, W)
    mc = np.array([c for c in merged.values() if c >= 2], dtype=np.uint16)
    gk, gc = gather_retained(dist, torch, mk, mc, W)
    gathered = [None] * world
    dist.all_gather_object(gathered, merged)
    if rank == 0:
        union = {}
        for m in gathered:
            assert not (set(m) & set(union))          # owners are disjoint
            union.update(m)
        json.dump({"hist": th.tolist(), "n": len(union), "records": sorted([list(kk) + [c] for kk, c in union.items()]),
                   "retained2": [[int(w) for w in kk] + [int(c)] for kk, c in zip(gk, gc)]}, open(out, "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
