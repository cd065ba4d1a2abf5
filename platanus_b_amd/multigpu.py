"""Multi-GPU merge of per-GPU occurrence tables: owner exchange over torch.distributed.

One process per GPU.  Reads are range-partitioned over ranks; every rank counts its reads into a private
device table; every (key,count) record then travels to the rank that owns the key's stripe,
owner = ((key.word0 & 1023) * world) >> 10 (the reference's stripe function, counter.h:461), in ONE
all_to_all over xGMI (backend "nccl" is RCCL on ROCm); owners add the counts (saturating at read-out,
counter.h:468-470), so the union of the owner shards is exactly the table a single process would have built.
Histograms (and read-length histograms) are summed with all_reduce.

The collective layer only sees flat integer tensors, so the same code runs over gloo with CPU tensors --
that is how tests/test_multigpu_gloo.py covers the N>1 path without a GPU.
"""
from __future__ import annotations

import ctypes as C
import os

from typing import Optional

import numpy as np


def owner_of(key_word0: np.ndarray, world: int) -> np.ndarray:
    return (((key_word0 & np.uint64(1023)) * np.uint64(world)) >> np.uint64(10)).astype(np.int64)


def exchange_records(dist, torch, keys, counts, send_counts, W: int):
    """keys: int64 tensor [n*W] bucketed by owner, counts: int32 tensor [n], send_counts: list[int] per owner.
    Returns (recv_keys, recv_counts, recv_counts_per_rank)."""
    world = dist.get_world_size()
    dev = keys.device
    if dev.type == "cuda" and dist.get_backend() == "gloo":
        # gloo has no all_to_all on device tensors: stage through the host (tests: two ranks sharing one GPU)
        rk, rcnt, recv = exchange_records(dist, torch, keys.cpu(), counts.cpu(), send_counts, W)
        return rk.to(dev), rcnt.to(dev), recv
    sc = torch.tensor(send_counts, dtype=torch.int64, device=dev)
    rc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc)
    recv = [int(x) for x in rc.tolist()]
    rk = torch.empty(sum(recv) * W, dtype=torch.int64, device=dev)
    rcnt = torch.empty(sum(recv), dtype=torch.int32, device=dev)
    dist.all_to_all_single(rk, keys, [r * W for r in recv], [s * W for s in send_counts])
    dist.all_to_all_single(rcnt, counts, recv, list(send_counts))
    return rk, rcnt, recv


def exchange_records_split(dist, torch, keys, counts16, n_single, n_multi, W: int):
    """The same exchange with 9 instead of 12 bytes per record: inside every owner's segment of `keys` (int64 [.. * W]) the records
    with count 1 come first and have no count at all, the others have a 16-bit count in `counts16` (uint8 [2 * sum(n_multi)], the
    little-endian u16 counts of owner 0's multis, owner 1's, ...; NCCL has no 16-bit integer type, so bytes travel).
    Returns (recv_keys, recv_counts int32 with the ones filled in, records received per rank)."""
    world = dist.get_world_size()
    dev = keys.device
    if dev.type == "cuda" and dist.get_backend() == "gloo":
        rk, rcnt, recv = exchange_records_split(dist, torch, keys.cpu(), counts16.cpu(), n_single, n_multi, W)
        return rk.to(dev), rcnt.to(dev), recv
    sc = torch.tensor([v for o in range(world) for v in (int(n_single[o]), int(n_multi[o]))], dtype=torch.int64, device=dev)
    rc = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rc, sc)
    rl = [int(x) for x in rc.tolist()]
    rs, rm = rl[0::2], rl[1::2]
    recv = [a + b for a, b in zip(rs, rm)]
    rk = torch.empty(sum(recv) * W, dtype=torch.int64, device=dev)
    dist.all_to_all_single(rk, keys, [r * W for r in recv], [(int(n_single[o]) + int(n_multi[o])) * W for o in range(world)])
    rc8 = torch.empty(2 * sum(rm), dtype=torch.uint8, device=dev)
    dist.all_to_all_single(rc8, counts16, [2 * m for m in rm], [2 * int(m) for m in n_multi])
    rcnt = torch.ones(sum(recv), dtype=torch.int32, device=dev)
    c16 = rc8.view(torch.int16)
    off = moff = 0
    for i in range(world):                      # source i sent rs[i] singles, then rm[i] multis
        if rm[i]:
            rcnt[off + rs[i]:off + recv[i]] = c16[moff:moff + rm[i]].to(torch.int32) & 0xFFFF
        off += recv[i]
        moff += rm[i]
    return rk, rcnt, recv


def count_exchanging_records(g, dist, world: int, rank: int, k: int, laps: Optional[dict] = None):
    """The N > 1 counting step with the exchange BEFORE the counting (k <= 32, world a power of two): every rank cuts its reads into
    16-byte run records grouped by owner rank (owner = top bits of the minimizer bucket), ONE all_to_all moves them, every rank counts
    what it received.  No export and no owner-side merge: copies of a run from different ranks meet in the bucket kernel's record
    table like copies from different reads.  Afterwards g holds this rank's share of the global table (a partition by bucket, not by
    the reference's stripes: any partition of the keys will do for retain + gather) and g.occ_hist / g.max_occ / g.n_distinct describe
    the GLOBAL distribution."""
    import torch
    from .binding import PB_OCC_BINS
    lib, h = g.lib, g.h
    gloo = dist.get_backend() == "gloo"
    dev = torch.device("cuda", torch.cuda.current_device())
    cdev = torch.device("cpu") if gloo else dev
    import time
    t_last = [time.perf_counter()]

    def lap(name):                                      # (development: wall time per section, device work included)
        if laps is not None:
            torch.cuda.synchronize()
            t = time.perf_counter()
            laps[name] = laps.get(name, 0.0) + (t - t_last[0]) * 1e3
            t_last[0] = t

    n_reads, n_bases = g.reads_info()
    if world & (world - 1):
        raise ValueError("count_exchanging_records needs a power-of-two number of ranks")
    n_owners = max(world, 2)                            # (one rank: two owner groups, both its own -- the single-rank RCCL test)
    # every rank receives about what it holds: the bucket bits of a single GPU with the largest rank's windows (agreed on once per
    # read set and k: the collective is not repeated by later passes over the same reads)
    memo = getattr(g, "_exchange_bits", None)
    if memo is None or memo[0] != (k, n_reads, n_bases, world):
        w = torch.tensor([max(n_bases - (k - 1) * n_reads, 1)], dtype=torch.int64, device=cdev)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        memo = g._exchange_bits = ((k, n_reads, n_bases, world), min(int(lib.pb_sk_bucket_bits(int(w.item()))), 20 - (n_owners.bit_length() - 1)))
    local_bits = memo[1]
    counts = (C.c_uint64 * n_owners)()
    nw = C.c_uint64()
    lap("bits")
    g._chk(lib.pb_sk_owner_records(h, k, n_owners, local_bits, counts, C.byref(nw)))
    lap("owner_records")
    send = [int(x) for x in counts] if world > 1 else [sum(int(x) for x in counts)]
    sc = torch.tensor(send, dtype=torch.int64, device=cdev)
    rc = torch.empty(world, dtype=torch.int64, device=cdev)
    dist.all_to_all_single(rc, sc)
    recv = [int(x) for x in rc.tolist()]
    lap("sizes")
    sbuf = torch.empty(max(sum(send), 1) * 16, dtype=torch.uint8, device=dev)
    g._chk(lib.pb_sk_owner_records_copy(h, C.c_void_p(sbuf.data_ptr())))        # (an owner's records lie in one region per XCD: packed here)
    lap("pack")
    if gloo:                                            # gloo has no all_to_all on device tensors: staged through the host (tests)
        rb = torch.empty(max(sum(recv), 1) * 16, dtype=torch.uint8)
        dist.all_to_all_single(rb[:sum(recv) * 16], sbuf[:sum(send) * 16].cpu(), [r * 16 for r in recv], [s * 16 for s in send])
        rbuf = rb.to(dev)
    else:
        rbuf = torch.empty(max(sum(recv), 1) * 16, dtype=torch.uint8, device=dev)
        dist.all_to_all_single(rbuf[:sum(recv) * 16], sbuf[:sum(send) * 16], [r * 16 for r in recv], [s * 16 for s in send])
    torch.cuda.synchronize()                 # the library runs on its own stream
    lap("exchange")
    hist = np.zeros(PB_OCC_BINS, dtype=np.uint64)
    mx, nd = C.c_uint64(), C.c_uint64()
    g._chk(lib.pb_sk_count_records(h, k, C.c_void_p(rbuf.data_ptr()), sum(recv), local_bits, hist.ctypes.data, C.byref(mx), C.byref(nd)))
    lap("count_records")
    # global histogram: the ranks first agree on the largest count anybody saw (one gather of three numbers per rank, which also sums
    # the distinct and window counts), then all-reduce only the bins up to it -- a few KB instead of the 65,535 bins
    mine = torch.tensor([mx.value, nd.value, nw.value], dtype=torch.int64, device=cdev)
    every = torch.empty(3 * world, dtype=torch.int64, device=cdev)
    dist.all_gather_into_tensor(every, mine)
    ev = every.cpu().numpy().reshape(world, 3)
    top = int(ev[:, 0].max())
    th = torch.from_numpy(hist[:top + 1].astype(np.int64)).to(cdev)
    dist.all_reduce(th)
    g.k = k
    g.occ_hist = np.zeros(PB_OCC_BINS, dtype=np.uint64)
    g.occ_hist[:top + 1] = th.cpu().numpy().astype(np.uint64)
    g.max_occ = top
    g.n_distinct_local = nd.value
    g.n_distinct = int(ev[:, 1].sum())
    g.n_instances = int(ev[:, 2].sum())
    g.records_sent, g.records_received = sum(send), sum(recv)
    lap("histogram")
    return g.occ_hist


def gather_retained(dist, torch, keys: np.ndarray, counts: np.ndarray, W: int):
    """All-gather of the owners' retained (key,count) shards: keys uint64 [n, W] (word 0 least significant),
    counts uint16 [n] of THIS rank's shard -> (keys, counts) of the union on every rank, ascending by the 2k-bit key value, i.e. the
    set the reference's sortedKeyFromKmerFile / loadKmer see (counter.h:916-951,599-640).  The C++ CLI does the same through
    pb_dist_gather_retained; this is the torch.distributed form (RCCL on GPUs, gloo in the CPU tests)."""
    world = dist.get_world_size()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n = int(len(counts))
    sizes = torch.zeros(world, dtype=torch.int64, device=dev)
    mine = torch.tensor([n], dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(sizes, mine)
    sz = [int(x) for x in sizes.tolist()]
    cap = max(max(sz), 1)
    buf = np.zeros((cap, W + 1), dtype=np.int64)              # key words + the count in one row
    if n:
        buf[:n, :W] = np.ascontiguousarray(keys).reshape(n, W).view(np.int64)
        buf[:n, W] = counts.astype(np.int64)
    send = torch.from_numpy(buf).to(dev)
    recv = torch.empty((world * cap, W + 1), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(recv, send)
    allr = recv.cpu().numpy()
    parts = [allr[r * cap:r * cap + sz[r]] for r in range(world)]
    cat = np.concatenate(parts) if parts else np.zeros((0, W + 1), dtype=np.int64)
    gk = np.ascontiguousarray(cat[:, :W]).view(np.uint64)
    gc = cat[:, W].astype(np.uint16)
    order = np.lexsort(tuple(gk[:, w] for w in range(W)))      # last key = most significant word (binstr.h:460-466)
    return gk[order], gc[order]


def exchange_and_merge(g, dist, world: int, rank: int, split: bool = True):
    """g: GpuCounter that has just counted this rank's reads.  split=False: every record travels with a 32-bit count (A/B runs).  After the call g's table is this rank's owner
    shard of the global table and g.occ_hist / g.max_occ / g.n_distinct describe the GLOBAL distribution."""
    import torch
    from .binding import PB_OCC_BINS, words_for_k
    lib, h = g.lib, g.h
    W = words_for_k(g.k)
    dev = torch.device("cuda", torch.cuda.current_device())
    if not split:                                      # every record with a 32-bit count
        part = (C.c_uint64 * world)()
        g._chk(lib.pb_export_begin(h, world, part))
        send = [int(x) for x in part]
        n = sum(send)
        keys = torch.empty(max(n, 1) * W, dtype=torch.int64, device=dev)
        counts = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        g._chk(lib.pb_export_copy(h, C.c_void_p(keys.data_ptr()), C.c_void_p(counts.data_ptr())))
        rk, rcnt, recv = exchange_records(dist, torch, keys[:n * W], counts[:n], send, W)
    else:
        ns, nm = (C.c_uint64 * world)(), (C.c_uint64 * world)()
        g._chk(lib.pb_export_split_begin(h, world, ns, nm))
        n_single, n_multi = [int(x) for x in ns], [int(x) for x in nm]
        n, n_m = sum(n_single) + sum(n_multi), sum(n_multi)
        keys = torch.empty(max(n, 1) * W, dtype=torch.int64, device=dev)
        counts16 = torch.empty(max(2 * n_m, 2), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        g._chk(lib.pb_export_split_copy(h, C.c_void_p(keys.data_ptr()), C.c_void_p(counts16.data_ptr())))
        rk, rcnt, recv = exchange_records_split(dist, torch, keys[:n * W], counts16[:2 * n_m], n_single, n_multi, W)
    torch.cuda.synchronize()                 # the library runs on its own stream
    total = sum(recv)
    hist = np.zeros(PB_OCC_BINS, dtype=np.uint64)
    mx, nd = C.c_uint64(), C.c_uint64()
    g._chk(lib.pb_merge_records(h, g.k, C.c_void_p(rk.data_ptr()), C.c_void_p(rcnt.data_ptr()), total, hist.ctypes.data, C.byref(mx), C.byref(nd)))
    red_dev = torch.device("cpu") if dist.get_backend() == "gloo" else dev
    # one all_reduce for the histogram and the distinct count (one more bin at the end)
    both = np.empty(PB_OCC_BINS + 1, dtype=np.int64)
    both[:PB_OCC_BINS] = hist.astype(np.int64)
    both[PB_OCC_BINS] = nd.value
    th = torch.from_numpy(both).to(red_dev)
    dist.all_reduce(th)
    tot = th.cpu().numpy()
    ghist = tot[:PB_OCC_BINS].astype(np.uint64)
    g.occ_hist = ghist
    nz = np.nonzero(ghist[1:65535])[0]
    g.max_occ = int(nz[-1]) + 1 if nz.size else 0
    g.n_distinct_local = nd.value
    g.n_distinct = int(tot[PB_OCC_BINS])
    return ghist
