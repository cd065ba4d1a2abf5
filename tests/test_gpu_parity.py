"""Parity tests proper: the HIP path (through the C-ABI, include/pbgpu.h) against the oracle and the golden
vectors.  Integer work: the bar is bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, case_reads

pytestmark = pytest.mark.gpu

from platanus_b_amd.binding import GpuCounter  # noqa: E402

COUNT_PATH = {"auto": 0, "direct": 1, "superk": 2}      # pb_set_option("count_path")


@pytest.fixture(scope="module")
def gpu():
    from platanus_b_amd.binding import GpuCounter
    return GpuCounter


def _oracle(codes, off, k, oracle_built):
    oc = oracle_built.OracleCounter()
    oc.count(codes.reshape(-1), off, k)
    return oc


@pytest.mark.parametrize("k", [32, 21, 1, 31, 33, 42, 64, 65, 75, 96, 97, 108, 128, 129, 150])
def test_count_matches_oracle_all_key_widths(k, gpu, oracle_built):
    from platanus_b_amd import synth
    r1, r2 = synth.make_reads(30_000, 40, seed=20 + k, n_rate=0.002)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = _oracle(codes, off, k, oracle_built)
    ohist, omax = oc.occ_hist()
    g = gpu(0)
    g.reads_append_matrix(r1)          # two appends: the second starts in the middle of a packed word
    g.reads_append_matrix(r2)
    hist = g.make_kmer_read_distribution(k)
    assert np.array_equal(hist, ohist)
    assert g.max_occ == omax
    assert g.n_distinct == int(ohist.sum())
    if omax < 65534:                    # below saturation sum(i * hist[i]) is the number of windows
        assert g.n_instances == int((np.arange(65535, dtype=np.uint64) * ohist).sum())
    assert np.array_equal(g.length_hist(200), oc.length_hist(200))
    for cutoff in (1, 2, 5):
        ok, ocn = oc.retain(cutoff)
        assert g.retain(cutoff) == len(ok)
        gk, gc = g.fetch_retained()
        assert np.array_equal(gk, ok)          # same keys in the same (ascending) order
        assert np.array_equal(gc, ocn)


@pytest.mark.parametrize("k", [159, 160])
def test_largest_k_of_the_fixed_width_kmers(k, gpu, oracle_built):
    """k = 160 is the largest k of the reference's fixed-width key types (Kmer159 = five 64-bit words, kmer.h / binstr.h); 161 and more is
    refused by the library.  PE250 reads, counts and retained keys against the oracle."""
    from platanus_b_amd import synth
    from platanus_b_amd.binding import PbError
    r1, r2 = synth.make_reads(20_000, 40, seed=300 + k, read_len=250, n_rate=0.001)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    oc = _oracle(codes, np.arange(n + 1, dtype=np.uint64) * np.uint64(L), k, oracle_built)
    ohist, omax = oc.occ_hist()
    g = gpu(0)
    g.reads_append_matrix(codes)
    hist = g.make_kmer_read_distribution(k)
    assert np.array_equal(hist, ohist) and g.max_occ == omax
    ok, ocn = oc.retain(2)
    assert g.retain(2) == len(ok)
    gk, gc = g.fetch_retained()
    assert np.array_equal(gk, ok) and np.array_equal(gc, ocn)
    with pytest.raises(PbError):
        g.make_kmer_read_distribution(161)


def test_ragged_reads_empty_reads_and_n_runs(gpu, oracle_built):
    rng = np.random.default_rng(7)
    lens = np.concatenate([[0, 1, 31, 32, 33, 0, 64, 499, 32, 0, 150, 151], rng.integers(0, 300, 2000)]).astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    codes = rng.integers(0, 4, int(off[-1])).astype(np.uint8)
    codes[rng.random(codes.size) < 0.01] = 4
    codes[off[7] + 100:off[7] + 140] = 4
    for k in (32, 45):
        oc = _oracle(codes, off, k, oracle_built)
        ohist, omax = oc.occ_hist()
        g = gpu(0)
        # ragged appends of 1, 2, 3 ... reads so that every word alignment of an append boundary occurs
        i, step = 0, 1
        while i < len(lens):
            j = min(i + step, len(lens))
            g.reads_append(codes, off[i:j + 1])
            i, step = j, step + 1
        hist = g.make_kmer_read_distribution(k)
        assert np.array_equal(hist, ohist) and g.max_occ == omax
        ok, ocn = oc.retain(1)
        g.retain(1)
        gk, gc = g.fetch_retained()
        assert np.array_equal(gk, ok) and np.array_equal(gc, ocn)
        assert np.array_equal(g.length_hist(600), oc.length_hist(600))


def test_ascii_input_uses_reference_char_table(gpu, oracle_built):
    """common.h:256: upper and lower case, N/n -> unknown."""
    reads = [b"ACGTacgtNNACGTTGCAtgcatgcaACGTACGTACGTACGTACGTnACGTTTGACCAGTCAGGCATTACGGATCAAAA", b"acgtacgtacgtacgtacgtacgtacgtacgtacgt"]
    asc = np.frombuffer(b"".join(reads), dtype=np.uint8)
    off = np.array([0, len(reads[0]), len(reads[0]) + len(reads[1])], dtype=np.uint64)
    lut = np.full(256, 4, dtype=np.uint8)
    for c, v in zip(b"ACGTacgt", [0, 1, 2, 3, 0, 1, 2, 3]):
        lut[c] = v
    codes = lut[asc]
    oc = _oracle(codes, off, 21, oracle_built)
    g = gpu(0)
    g.reads_append(asc, off, ascii=True)
    hist = g.make_kmer_read_distribution(21)
    ohist, _ = oc.occ_hist()
    assert np.array_equal(hist, ohist)
    ok, ocn = oc.retain(1)
    g.retain(1)
    gk, gc = g.fetch_retained()
    assert np.array_equal(gk, ok) and np.array_equal(gc, ocn)


def test_longest_read_the_reference_accepts(gpu, oracle_built, monkeypatch):
    """common.h:465-467 / common.cpp:28: a read of 500,000 bases or more is an error (ReadError), 499,999 is the longest.  Such a read
    spans 62 tiles of the partition kernel; three copies of it plus ordinary reads, counted by both passes, against the oracle."""
    from platanus_b_amd.binding import PbError
    rng = np.random.default_rng(12)
    long_read = rng.integers(0, 4, 499_999).astype(np.uint8)
    long_read[1000:1010] = 4
    short = rng.integers(0, 4, (2000, 150)).astype(np.uint8)
    codes = np.concatenate([long_read, long_read, short.reshape(-1), long_read])
    off = np.concatenate([[0, 499_999, 999_998], 999_998 + 150 * np.arange(1, 2001), [999_998 + 300_000 + 499_999]]).astype(np.uint64)
    oc = _oracle(codes, off, 32, oracle_built)
    ohist, omax = oc.occ_hist()
    ok, ocn = oc.retain(1)
    for path in ("direct", "superk"):
        monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH[path])
        g = gpu(0)
        g.reads_append(codes, off)
        hist = g.make_kmer_read_distribution(32)
        assert (g.kernel_ms("part1") > 0) == (path == "superk")
        assert np.array_equal(hist, ohist) and g.max_occ == omax
        g.retain(1)
        gk, gc = g.fetch_retained()
        assert np.array_equal(gk, ok) and np.array_equal(gc, ocn)
    g = gpu(0)
    with pytest.raises(PbError) as e:
        g.reads_append(np.zeros(500_000, dtype=np.uint8), np.array([0, 500_000], dtype=np.uint64))
    assert e.value.code == -4 or "500000" in str(e.value)


def test_saturation_at_65534(gpu):
    """counter.h:468-470: counts stop at UINT16_MAX-1; 70000 copies of one read."""
    read = np.array([0, 1, 2, 3] * 8 + [2, 2, 1], dtype=np.uint8)       # 35 bases -> 4 windows of 32
    n = 70000
    codes = np.tile(read, n)
    g = gpu(0)
    g.reads_append(codes, np.arange(n + 1, dtype=np.uint64) * np.uint64(len(read)))
    hist = g.make_kmer_read_distribution(32)
    assert g.max_occ == 65534
    assert int(hist[65534]) == g.n_distinct
    g.retain(1)
    _, gc = g.fetch_retained()
    assert (gc == 65534).all()


def test_histogram_matches_reference_golden_tsv(gpu, manifest, tmp_path):
    """PREFIX_32merFrq.tsv byte-identical to the file the real reference wrote (tests/golden)."""
    for name, run, k in (("c1", "occ", 32), ("g5k", "k21_s0", 21), ("k45", "multik_k45", 45), ("c2", "multik_n2", 32)):
        codes, off = case_reads(manifest[name])
        g = gpu(0)
        g.reads_append_matrix(codes)
        g.make_kmer_read_distribution(k)
        out = tmp_path / f"{name}_{k}merFrq.tsv"
        g.output_occurrence_distribution(str(out))
        gold = os.path.join(GOLDEN, f"{name}.{run}.out_{k}merFrq.tsv")
        assert open(out, "rb").read() == open(gold, "rb").read()


def test_full_size_properties(gpu):
    """BASELINE config 2 shape at a size the oracle cannot finish in seconds (1 Mbp x 100x): size-independent
    properties -- sum(hist) = distinct, sum(i*hist) = windows, retained keys strictly ascending, counting twice
    gives the same table (idempotence), and read order does not matter (counts are commutative)."""
    from platanus_b_amd import synth
    r1, r2 = synth.make_reads(1_000_000, 100, seed=5)
    g = gpu(0)
    g.reads_append_matrix(r1)
    g.reads_append_matrix(r2)
    h1 = g.make_kmer_read_distribution(32).copy()
    n_reads, n_bases = g.reads_info()
    assert g.n_instances == n_reads * (150 - 32 + 1)
    assert int(h1.sum()) == g.n_distinct
    assert int((np.arange(65535, dtype=np.uint64) * h1).sum()) == g.n_instances
    g.retain(3)
    k1, c1 = g.fetch_retained()
    assert (k1[1:, 0] > k1[:-1, 0]).all()
    assert (c1 >= 3).all()
    h2 = g.make_kmer_read_distribution(32)
    assert np.array_equal(h1, h2)
    g2 = gpu(0)
    g2.reads_append_matrix(r2[::-1])
    g2.reads_append_matrix(r1[::-1])
    h3 = g2.make_kmer_read_distribution(32)
    assert np.array_equal(h1, h3)
    g2.retain(3)
    k3, c3 = g2.fetch_retained()
    assert np.array_equal(k1, k3) and np.array_equal(c1, c3)


def test_config2_full_size_through_the_cabi(gpu, manifest, tmp_path, monkeypatch):
    """BASELINE configs[1] at its full size (5 Mbp x 100x PE150, k = 32: 396.7 M windows, 60.8 M distinct 32-mers) through
    the C-ABI.  The oracle needs minutes here, so the checks are (1) the occurrence distribution written as the reference
    writes it (counter.h:999-1007) against the md5 of the REAL reference's PREFIX_32merFrq.tsv for this input
    (tests/golden/manifest.json, cfg3), (2) sum(hist) = distinct and sum(i*hist) = windows, (3) strand invariance: the
    reverse complements of all reads give the same retained (key,count) table (keys are canonical, counter.h:429),
    (4) the direct global-table pass and the super-k-mer pass agree on that table."""
    import hashlib
    from platanus_b_amd import synth
    case = manifest["cfg3"]
    G, cov, seed = case["genome"], case["cov"], case["seed"]
    genome = synth.make_genome(G, seed)
    n_pairs = int(G * cov) // 300
    g, grc = gpu(0), gpu(0)
    for lo in range(0, n_pairs, 250_000):
        r1, r2 = synth.make_reads(G, cov, seed, genome=genome, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs), err=case["err"])
        g.reads_append_matrix(r1)
        g.reads_append_matrix(r2)
        grc.reads_append_matrix(np.ascontiguousarray(3 - r2[:, ::-1]))
        grc.reads_append_matrix(np.ascontiguousarray(3 - r1[:, ::-1]))
    h = g.make_kmer_read_distribution(32).copy()
    assert g.kernel_ms("part1") > 0                          # the super-k-mer pass took it
    n_reads, _ = g.reads_info()
    assert n_reads == 2 * n_pairs and g.n_instances == n_reads * 119
    assert int(h.sum()) == g.n_distinct
    assert int((np.arange(65535, dtype=np.uint64) * h).sum()) == g.n_instances
    out = tmp_path / "out_32merFrq.tsv"
    g.output_occurrence_distribution(str(out))
    assert hashlib.md5(open(out, "rb").read()).hexdigest() == case["runs"]["multik"]["files"]["out_32merFrq.tsv"]["md5"]
    n = g.retain(3)
    keys, counts = g.fetch_retained()
    assert n == len(counts) and (keys[1:, 0] > keys[:-1, 0]).all()
    hrc = grc.make_kmer_read_distribution(32)
    assert np.array_equal(h, hrc)
    grc.retain(3)
    krc, crc = grc.fetch_retained()
    assert np.array_equal(keys, krc) and np.array_equal(counts, crc)
    del grc
    g.set_option("count_path", COUNT_PATH["direct"])
    hd = g.make_kmer_read_distribution(32)
    assert g.kernel_ms("part1") == 0
    assert np.array_equal(h, hd)
    g.retain(3)
    kd, cd = g.fetch_retained()
    assert np.array_equal(keys, kd) and np.array_equal(counts, cd)


@pytest.mark.parametrize("name,G,cov,seed", [("configs[3]", 5_000_000, 1000.0, 1), ("configs[4]", 100_000_000, 100.0, 5)])
def test_eight_gpu_configs_at_their_per_gpu_shard_shape(name, G, cov, seed, gpu):
    """BASELINE configs[3] (5 Mbp x 1000x) and configs[4] (100 Mbp x 100x) as ONE of their 8 GPUs sees them before the merge: the
    first eighth of the read pairs (496 M / 992 M windows; for the 100 Mbp genome that is 12.5x coverage, where most k-mers of the
    shard are seen a handful of times).  Full-size property checks: the super-k-mer pass takes the input, sum(hist) = distinct,
    sum(i * hist) = windows, the reverse complements of all reads give the same retained table, and the direct global-table pass
    agrees with the super-k-mer pass on histogram and retained (key,count) set."""
    from platanus_b_amd import synth
    genome = synth.make_genome(G, seed)
    n_pairs = int(G * cov) // 300 // 8
    g, grc = gpu(0), gpu(0)
    for lo in range(0, n_pairs, 250_000):
        r1, r2 = synth.make_reads(G, cov, seed, genome=genome, pair_lo=lo, pair_hi=min(lo + 250_000, n_pairs))
        g.reads_append_matrix(r1)
        g.reads_append_matrix(r2)
        grc.reads_append_matrix(np.ascontiguousarray(3 - r2[:, ::-1]))
        grc.reads_append_matrix(np.ascontiguousarray(3 - r1[:, ::-1]))
    h = g.make_kmer_read_distribution(32).copy()
    assert g.kernel_ms("part1") > 0, "the super-k-mer pass must take a shard of this size"
    print(f"{name} shard: {g.n_instances} windows, {g.n_distinct} distinct, 2^{int(g.kernel_ms('sk_bucket_bits'))} buckets, "
          f"{int(g.kernel_ms('sk_flagged_buckets'))} flagged, {int(g.kernel_ms('sk_spilled_records'))} spilled records, kernels {g.kernel_ms('count'):.2f} ms")
    assert g.n_instances == 2 * n_pairs * 119
    assert int(h.sum()) == g.n_distinct
    assert int((np.arange(65535, dtype=np.uint64) * h).sum()) == g.n_instances
    n = g.retain(2)
    keys, counts = g.fetch_retained()
    assert n == len(counts) and (keys[1:, 0] > keys[:-1, 0]).all()
    assert np.array_equal(h, grc.make_kmer_read_distribution(32))
    grc.retain(2)
    krc, crc = grc.fetch_retained()
    assert np.array_equal(keys, krc) and np.array_equal(counts, crc)
    del grc, krc, crc
    g.set_option("count_path", COUNT_PATH["direct"])
    hd = g.make_kmer_read_distribution(32)
    assert g.kernel_ms("part1") == 0
    assert np.array_equal(h, hd)
    g.retain(2)
    kd, cd = g.fetch_retained()
    assert np.array_equal(keys, kd) and np.array_equal(counts, cd)


@pytest.mark.parametrize("k", [32, 45])
def test_export_buckets_and_merge_roundtrip(k, gpu, oracle_built):
    """X1 on one GPU: bucket the table into 8 owner parts, check every record sits in its owner's bucket, merge the
    parts back (twice: counts double, saturating) and compare with the oracle."""
    import ctypes as C
    import torch
    from platanus_b_amd import synth
    from platanus_b_amd.multigpu import owner_of
    r1, r2 = synth.make_reads(30_000, 40, seed=3, n_rate=0.001)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = _oracle(codes, off, k, oracle_built)
    ok, ocn = oc.retain(1)
    W = (k + 31) // 32
    g = gpu(0)
    g.reads_append_matrix(codes)
    g.make_kmer_read_distribution(k)
    parts = (C.c_uint64 * 8)()
    g._chk(g.lib.pb_export_begin(g.h, 8, parts))
    send = [int(x) for x in parts]
    assert sum(send) == len(ok)
    tk = torch.empty(sum(send) * W, dtype=torch.int64, device="cuda")
    tc = torch.empty(sum(send), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    g._chk(g.lib.pb_export_copy(g.h, C.c_void_p(tk.data_ptr()), C.c_void_p(tc.data_ptr())))
    keys = tk.cpu().numpy().view(np.uint64).reshape(-1, W)
    own = owner_of(keys[:, 0], 8)
    bounds = np.concatenate([[0], np.cumsum(send)])
    for r in range(8):
        assert (own[bounds[r]:bounds[r + 1]] == r).all()
    g._chk(g.lib.pb_merge_begin(g.h, k, sum(send)))
    for _ in range(2):
        g._chk(g.lib.pb_merge_add(g.h, C.c_void_p(tk.data_ptr()), C.c_void_p(tc.data_ptr()), sum(send)))
    hist = np.zeros(65535, dtype=np.uint64)
    mx, nd = C.c_uint64(), C.c_uint64()
    g._chk(g.lib.pb_merge_end(g.h, hist.ctypes.data, C.byref(mx), C.byref(nd)))
    assert nd.value == len(ok)
    g.retain(1)
    gk, gc = g.fetch_retained()
    assert np.array_equal(gk, ok)
    assert np.array_equal(gc, np.minimum(ocn.astype(np.int64) * 2, 65534).astype(np.uint16))


@pytest.mark.parametrize("k", [32, 45])
def test_split_export_layout(k, gpu, oracle_built):
    """pb_export_split_*: 5 owners; inside every owner's key segment the count-1 records come first (no count travels for them), then
    the others, whose 16-bit counts sit owner by owner in the count buffer.  Reassembled, the records are exactly the oracle's table."""
    import ctypes as C
    import torch
    from platanus_b_amd import synth
    from platanus_b_amd.multigpu import owner_of
    r1, r2 = synth.make_reads(30_000, 40, seed=4, n_rate=0.001)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    oc = _oracle(codes, np.arange(n + 1, dtype=np.uint64) * np.uint64(L), k, oracle_built)
    ok, ocn = oc.retain(1)
    W = (k + 31) // 32
    P = 5
    g = gpu(0)
    g.reads_append_matrix(codes)
    g.make_kmer_read_distribution(k)
    ns, nm = (C.c_uint64 * P)(), (C.c_uint64 * P)()
    g._chk(g.lib.pb_export_split_begin(g.h, P, ns, nm))
    ns, nm = [int(x) for x in ns], [int(x) for x in nm]
    assert sum(ns) + sum(nm) == len(ok) and sum(ns) == int((ocn == 1).sum())
    tk = torch.empty((sum(ns) + sum(nm)) * W, dtype=torch.int64, device="cuda")
    tc = torch.empty(max(2 * sum(nm), 2), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    g._chk(g.lib.pb_export_split_copy(g.h, C.c_void_p(tk.data_ptr()), C.c_void_p(tc.data_ptr())))
    keys = tk.cpu().numpy().view(np.uint64).reshape(-1, W)
    c16 = tc.cpu().numpy()[:2 * sum(nm)].view("<u2")
    own = owner_of(keys[:, 0], P)
    counts = np.ones(len(keys), dtype=np.int64)
    koff = moff = 0
    for o in range(P):
        assert (own[koff:koff + ns[o] + nm[o]] == o).all()
        counts[koff + ns[o]:koff + ns[o] + nm[o]] = c16[moff:moff + nm[o]]
        assert (c16[moff:moff + nm[o]] > 1).all()
        koff += ns[o] + nm[o]
        moff += nm[o]
    order = np.lexsort(tuple(keys[:, w] for w in range(W)))       # most significant word last = primary
    assert np.array_equal(keys[order], ok) and np.array_equal(counts[order], ocn.astype(np.int64))


def test_exchange_and_merge_over_rccl_single_rank(gpu, oracle_built):
    """The multi-GPU step (export by owner -> all_to_all over RCCL -> owner merge -> histogram all-reduce) on a real
    device with world_size 1: every record travels through torch.distributed (backend nccl = RCCL) and must come back
    unchanged; k = 32 (1 word) and k = 45 (2 words)."""
    import torch
    import torch.distributed as dist
    from platanus_b_amd import synth
    from platanus_b_amd.multigpu import exchange_and_merge
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        r1, r2 = synth.make_reads(30_000, 40, seed=8, n_rate=0.001)
        codes = np.concatenate([r1, r2])
        n, L = codes.shape
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
        for k in (32, 45):
            oc = _oracle(codes, off, k, oracle_built)
            ohist, omax = oc.occ_hist()
            g = gpu(0)
            g.reads_append_matrix(codes)
            g.make_kmer_read_distribution(k)
            ghist = exchange_and_merge(g, dist, 1, 0)
            assert np.array_equal(ghist, ohist) and g.max_occ == omax
            assert g.n_distinct == int(ohist.sum())
            ok, ocn = oc.retain(2)
            g.retain(2)
            gk, gc = g.fetch_retained()
            assert np.array_equal(gk, ok) and np.array_equal(gc, ocn)
            if k <= 32:                                   # the other design: run records through the all_to_all, counted on arrival
                from platanus_b_amd.multigpu import count_exchanging_records
                g2 = gpu(0)
                g2.reads_append_matrix(codes)
                ghist2 = count_exchanging_records(g2, dist, 1, 0, k)
                assert np.array_equal(ghist2, ohist) and g2.max_occ == omax and g2.n_distinct == int(ohist.sum())
                assert g2.records_received == g2.records_sent > 0
                g2.retain(2)
                gk2, gc2 = g2.fetch_retained()
                assert np.array_equal(gk2, ok) and np.array_equal(gc2, ocn)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("path", ["direct", "superk"])
@pytest.mark.parametrize("mem_bytes,cutoff", [(1 << 20, 1), (1 << 20, 2), (3 << 20, 1), (10 ** 9, 2)])
def test_kmer_occ_bin_byte_exact_layout(mem_bytes, cutoff, path, gpu, oracle_built, tmp_path, monkeypatch):
    """CT8: PREFIX_kmer_occ.bin byte-identical to the -t 1 reference layout.  Small -m values make the 1024 stripe tables
    tiny (64 / 128 slots), so the 32-probe limit, the spill passes and the iterateTimes increments are all exercised.
    The first occurrence of every k-mer comes from the probing kernel after a direct pass, and from the position-carrying
    super-k-mer pass (sk_*<POS>) after a super-k-mer pass."""
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH[path])
    from platanus_b_amd import synth
    r1, r2 = synth.make_reads(40_000, 40, seed=12, n_rate=0.001)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = oracle_built.OracleCounter()
    dh = oc.count(codes.reshape(-1), off, 32, mem_bytes)
    oc.write_table_binary(cutoff, str(tmp_path / "oracle.bin"))
    g = gpu(0)
    g.reads_append_matrix(codes)
    g.make_kmer_read_distribution(32)
    g.output_occurrence_table_binary(str(tmp_path / "gpu.bin"), cutoff, dh)
    a, b = open(tmp_path / "gpu.bin", "rb").read(), open(tmp_path / "oracle.bin", "rb").read()
    assert len(a) == len(b)
    assert a == b


def test_kmer_occ_bin_matches_reference_golden_file(gpu, manifest, tmp_path):
    """The file the real reference wrote (`-k 32 -t 1 -m 1 -n 2 -kmer_occ_only`, tests/golden)."""
    codes, off = case_reads(manifest["g5k"])
    g = gpu(0)
    g.reads_append_matrix(codes)
    g.make_kmer_read_distribution(32)
    g.output_occurrence_table_binary(str(tmp_path / "gpu.bin"), 2, 1 << 25)       # -m 1 -> 2^25 slots (counter.h:300-309)
    assert open(tmp_path / "gpu.bin", "rb").read() == open(os.path.join(GOLDEN, "g5k.occ_n2.out_kmer_occ.bin"), "rb").read()


def test_low_coverage_mostly_distinct_kmers(gpu, oracle_built, monkeypatch):
    """2x coverage: almost every k-mer instance is distinct (the opposite regime of the benchmark); both counting paths."""
    from platanus_b_amd import synth
    r1, r2 = synth.make_reads(300_000, 2, seed=91, err=0.02)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = _oracle(codes, off, 31, oracle_built)
    ohist, omax = oc.occ_hist()
    for path in ("superk", "direct"):
        monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH[path])
        g = gpu(0)
        g.reads_append_matrix(codes)
        hist = g.make_kmer_read_distribution(31)
        assert np.array_equal(hist, ohist) and g.max_occ == omax, path


def _superk_case(gpu, oracle_built, codes, k, cutoffs=(1, 3), must_take_path=True):
    n, L = codes.shape
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = _oracle(codes, off, k, oracle_built)
    ohist, omax = oc.occ_hist()
    g = gpu(0)
    g.reads_append_matrix(codes)
    hist = g.make_kmer_read_distribution(k)
    if must_take_path:
        assert g.kernel_ms("part1") > 0
    assert np.array_equal(hist, ohist) and g.max_occ == omax
    assert g.n_distinct == int(ohist.sum())
    for cutoff in cutoffs:
        ok, ocn = oc.retain(cutoff)
        assert g.retain(cutoff) == len(ok)
        gk, gc = g.fetch_retained()
        assert np.array_equal(gk, ok) and np.array_equal(gc, ocn)
    return g


@pytest.mark.parametrize("k", [32, 31, 25, 21, 20, 14, 13, 9, 2, 1])
def test_superkmer_count_path_matches_oracle(k, gpu, oracle_built, monkeypatch):
    """Super-k-mer records + minimizer buckets + LDS counting (kernels_superk.inc) forced on a small input with Ns:
    every window width w = k - m + 1 class (w >= 8 uses the a8 window minimum, w < 8 the direct one, k < m has w = 1).
    With k = 1 or 2 a handful of minimizers carry the whole input: the pass notices (spill list full) and hands over to the
    direct path, which must give the same answer."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(60_000, 40, seed=140 + k, n_rate=0.002)
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), k, must_take_path=k >= 9)


@pytest.mark.parametrize("k", [33, 34, 40, 45, 47, 48, 51, 52, 63, 64])
def test_superkmer_two_word_keys(k, gpu, oracle_built, monkeypatch):
    """33 <= k <= 64 through the super-k-mer pass for two-word keys (kernels_superk2.inc: minimizers of k - 19 bases hashed on both
    strands, 24-byte records, claim-through-the-count-word table of two-word keys) against the oracle: histogram, max_occ, the
    retained (key,count) sets.  k = 33 / 64 are the ends of the range, 47 / 48 / 51 / 52 sit around the 32-base minimizer and the
    record's word boundaries; reads carry N bases."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(60_000, 40, seed=300 + k, n_rate=0.002)
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), k)


@pytest.mark.parametrize("k", [65, 66, 72, 75, 77, 82, 83, 84, 95, 96])
def test_superkmer_three_word_keys(k, gpu, oracle_built, monkeypatch):
    """r3: 65 <= k <= 96 through the super-k-mer pass for three-word keys (kernels_superk3.inc: minimizers of k - 19 = 46 .. 77 bases hashed on
    both strands, 32-byte records, table of three-word keys claimed on the low word) against the oracle: histogram, max_occ, the retained
    (key,count) sets.  k = 65 / 96 are the ends of the range (2 and 64 bits in the top word), 82 / 83 / 84 sit around the 64-base
    minimizer, 75 is the last k of the reference's default schedule on 150-base reads; reads carry N bases."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(60_000, 40, seed=600 + k, n_rate=0.002)
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), k)


@pytest.mark.parametrize("bits,k", [(3, 75), (9, 65), (12, 96), (12, 83)])
def test_superkmer_three_word_keys_small_tables_and_low_complexity(bits, k, gpu, oracle_built, monkeypatch):
    """Three-word keys with 8 .. 4096 buckets: the record table is far too small (windows go straight into the table of keys), tables of
    keys overflow (flagged buckets through the global table), poly-A / poly-T reads bring keys whose low word is all ones (the empty mark:
    their buckets are flagged) and long runs in one bucket (spill list)."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    monkeypatch.setitem(GpuCounter.default_options, "sk_bucket_bits", bits)
    r1, r2 = synth.make_reads(70_000, 50, seed=700 + bits + k, err=0.01, n_rate=0.001)
    poly = np.full((40, r1.shape[1]), 3, dtype=np.uint8)
    poly[:20] = 0
    poly[::3, 70] = 1
    _superk_case(gpu, oracle_built, np.concatenate([r1, poly, r2]), k, cutoffs=(1, 3), must_take_path=False)


def test_superkmer_three_word_keys_low_coverage_and_short_reads(gpu, oracle_built, monkeypatch):
    """2x coverage with 2 % errors at k = 75 (nearly every window a distinct key: the pass repeats itself with more buckets or hands over to
    the direct pass -- same results), and reads shorter than k (nothing to count)."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(300_000, 2, seed=791, err=0.02)
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 75, cutoffs=(1,), must_take_path=False)
    short = synth.make_reads(20_000, 10, seed=778, read_len=70)[0]
    _superk_case(gpu, oracle_built, short, 75, cutoffs=(1,), must_take_path=False)


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("bits,k", [(3, 45), (9, 33), (12, 64), (12, 48)])
def test_superkmer_two_word_keys_both_forms_of_the_bucket_kernel(form, bits, k, gpu, oracle_built, monkeypatch):
    """r3: sk2_bucket2_kernel (whole bucket per round, block-wide piece list, two home slots + parked keys, rolled reverse key: the default)
    and sk2_bucket_kernel (chunks, per-wave piece lists, claim loops) against the oracle.  With 8 buckets the record table of either form is
    far too small (the second form then puts windows straight into the table of keys, on the slot sequence the pieces use) and the tables of
    keys overflow: those buckets go through the global table.  Poly-T / poly-A reads bring keys whose low word is all ones (the second
    form's empty mark: their buckets are flagged)."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    monkeypatch.setitem(GpuCounter.default_options, "sk_bucket_bits", bits)
    monkeypatch.setitem(GpuCounter.default_options, "sk_bucket_form", form)
    r1, r2 = synth.make_reads(70_000, 50, seed=500 + bits + k, err=0.01, n_rate=0.001)
    poly = np.full((40, r1.shape[1]), 3, dtype=np.uint8)
    poly[:20] = 0
    poly[::3, 70] = 1                                        # (some of them broken by one other base)
    _superk_case(gpu, oracle_built, np.concatenate([r1, poly, r2]), k, cutoffs=(1, 3), must_take_path=False)


def test_dist_rccl_calls_with_the_rank_as_its_own_peer(gpu):
    """csrc/dist.inc talks to RCCL through dlopen (no link-time dependency): symbols, the unique id passed by value, the datatype constant,
    group calls and the library's stream can be checked on one GPU with a communicator of one rank that sends to and receives from
    itself (pb_dist_rccl_selftest).  More than one rank needs more than one device: the driver's N > 1 runs."""
    import ctypes as C
    g = gpu(0)
    lib = g.lib
    lib.pb_dist_shared_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.pb_dist_attach.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.pb_dist_rccl_selftest.argtypes = [C.c_void_p, C.c_uint64]
    lib.pb_dist_detach.argtypes = [C.c_void_p]
    shared = C.c_void_p()
    assert lib.pb_dist_shared_create(1, C.byref(shared)) == 0
    g._chk(lib.pb_dist_attach(g.h, shared, 0, 1, 1))
    for n in (1, 4096, 3_000_001, 64 << 20):
        g._chk(lib.pb_dist_rccl_selftest(g.h, n))
    g._chk(lib.pb_dist_detach(g.h))


def test_superkmer_two_word_keys_low_coverage_repeats_with_more_buckets(gpu, oracle_built, monkeypatch):
    """2x coverage, 2 % errors, k = 40: nearly every window is a distinct key, several times what the tables of the first attempt hold --
    the pass reports the misfit, repeats itself with 4x the buckets and must then give the oracle's answer through its own kernels."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(300_000, 2, seed=391, err=0.02)
    g = _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 40, cutoffs=(1,))
    assert g.kernel_ms("sk_bucket_bits") >= 10


def test_superkmer_two_word_keys_flagged_buckets_through_the_global_table(gpu, oracle_built, monkeypatch):
    """A 21 kbp tandem repeat of a 7-mer inside a random genome, k = 45: the repeat's windows share a handful of minimizers, their buckets
    overflow, the records go to the spill list and sk2_flagged_kernel counts the flagged buckets through the global table -- the pass
    itself must finish (no hand-over to the direct pass) with the oracle's answer."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    genome = synth.make_genome(80_000, 5)
    genome[30_000:51_000] = np.resize(np.array([0, 1, 2, 3, 3, 2, 0], dtype=np.uint8), 21_000)
    r1, r2 = synth.make_reads(80_000, 40, seed=395, genome=genome, n_rate=0.001)
    g = _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 45, cutoffs=(2,))
    assert g.kernel_ms("sk_spilled_records") + g.kernel_ms("sk_flagged_buckets") > 0


def test_superkmer_two_word_keys_misfits_fall_back(gpu, oracle_built, monkeypatch):
    """Poly-A and dinucleotide reads put most records into a few buckets: spill list and global table take them, or -- when the spill
    list itself overflows -- the direct pass counts the input (same results either way).  Reads shorter than k contribute nothing."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(20_000, 30, seed=377)
    poly = np.zeros((3000, 150), dtype=np.uint8)
    di = np.tile(np.array([1, 2], dtype=np.uint8), (250, 75))
    _superk_case(gpu, oracle_built, np.concatenate([r1, poly, di, r2]), 45, cutoffs=(2,), must_take_path=False)
    short = synth.make_reads(20_000, 10, seed=378, read_len=40)[0]
    _superk_case(gpu, oracle_built, short, 45, cutoffs=(1,), must_take_path=False)


@pytest.mark.parametrize("bits", [2, 9, 15, 18, 19, 20])
def test_superkmer_other_bucket_counts(bits, gpu, oracle_built, monkeypatch):
    """4 .. 2^20 minimizer buckets: 4 buckets overflow their LDS tables and go through the global table, 2^18 is what a
    3 G-window shard would use, 2^20 (level 1 with 2^11 digits per XCD: the reservation loop instead of one lane per digit, 24 KB of
    digit arrays) what 13 G windows on one GPU would -- BASELINE configs[4] whole on one device."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    monkeypatch.setitem(GpuCounter.default_options, "sk_bucket_bits", bits)
    r1, r2 = synth.make_reads(50_000, 50, seed=160 + bits, n_rate=0.001)
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 32, cutoffs=(1,))


def test_superkmer_heavy_hitter_and_low_complexity(gpu, oracle_built, monkeypatch):
    """Poly-A reads (every m-mer of a window ties for the minimum, so runs are 1 to 20 windows long and all land in one bucket, which
    overflows its record capacity: spill list + flagged bucket through the global table; one 32-mer with 59500 instances) and
    dinucleotide repeats; results must not change.  With seven times the poly-A reads the spill list overflows too and the pass
    hands the input to the direct path: same results."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(20_000, 30, seed=177)
    di = np.tile(np.array([1, 2], dtype=np.uint8), (250, 75))
    for n_poly, must in ((500, True), (3600, False)):
        poly = np.zeros((n_poly, 150), dtype=np.uint8)
        _superk_case(gpu, oracle_built, np.concatenate([r1, poly, di, r2]), 32, cutoffs=(2,), must_take_path=must)


def test_superkmer_low_coverage_mostly_distinct(gpu, oracle_built, monkeypatch):
    """2x coverage, 2 % errors: nearly every instance is a distinct key, far more than the LDS tables are sized for."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(300_000, 2, seed=191, err=0.02)
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 31, cutoffs=(1,))


def test_merge_records_through_lds_tables(gpu, oracle_built, monkeypatch):
    """pb_merge_records on > 1 M records takes the partition + LDS-table path (stage "merge" of the counting machinery); the
    result must equal both the oracle and the global-table path (pb_set_option "merge_table").  Input: every distinct k-mer of a 2 %-error
    read set split into two records whose counts add up, plus a heavy hitter that arrives as 3000 records."""
    import ctypes as C
    import torch
    from platanus_b_amd import synth
    k = 31
    r1, r2 = synth.make_reads(400_000, 16, seed=23, err=0.02)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = _oracle(codes, off, k, oracle_built)
    ok, ocn = oc.retain(1)
    ohist, omax = oc.occ_hist()
    g = gpu(0)
    g.reads_append_matrix(codes)
    g.make_kmer_read_distribution(k)
    keys, counts = g.fetch_all()
    keys = keys.reshape(-1)
    assert len(keys) == len(ok) and len(keys) > 1_000_000
    rng = np.random.default_rng(5)
    first = (counts.astype(np.int64) * rng.random(len(counts))).astype(np.int64)
    rk = np.concatenate([keys, keys])
    rc = np.concatenate([first, counts.astype(np.int64) - first])          # zero counts are legal records: ignored
    hot = int(np.argmax(counts))
    rk = np.concatenate([rk, np.full(3000, keys[hot], dtype=np.uint64)])
    rc = np.concatenate([rc, np.full(3000, 20, dtype=np.int64)])
    perm = rng.permutation(len(rk))
    tk = torch.from_numpy(rk[perm].view(np.int64)).cuda()
    tc = torch.from_numpy(rc[perm].astype(np.int32)).cuda()
    expect = np.minimum(ocn.astype(np.int64) + np.where(ok.reshape(-1) == keys[hot], 60000, 0), 65534).astype(np.uint16)
    for table_path in (False, True):
        if table_path:
            g.set_option("merge_table", 1)
        hist = np.zeros(65535, dtype=np.uint64)
        mx, nd = C.c_uint64(), C.c_uint64()
        torch.cuda.synchronize()
        g._chk(g.lib.pb_merge_records(g.h, k, C.c_void_p(tk.data_ptr()), C.c_void_p(tc.data_ptr()), len(rk), hist.ctypes.data, C.byref(mx), C.byref(nd)))
        assert nd.value == len(ok), table_path
        g.retain(1)
        gk, gc = g.fetch_retained()
        assert np.array_equal(gk.reshape(-1), ok.reshape(-1)) and np.array_equal(gc, expect), table_path


@pytest.mark.parametrize("world,port,design", [(2, 29551, "counts"), (4, 29552, "counts"), (2, 29553, "records"), (4, 29554, "records")])
def test_two_ranks_share_one_gpu(world, port, design, oracle_built, tmp_path):
    """The N > 1 step with real kernels: two (four) processes on GPU 0 (gloo collectives staged through the host), each holding half
    (a quarter) of the read pairs.  Design "counts": super-k-mer count, export by owner, exchange of (key,count) records, LDS-table
    merge.  Design "records": run records by owner, exchange of the records, then the counting (no export, no merge).  The all-reduced
    histogram and the union of the ranks' shards must equal the single-process oracle on all reads."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    from platanus_b_amd import synth
    k, genome, cov, seed = 31, 300_000, 40.0, 33
    out = tmp_path / "res.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_gpu2_worker.py"), str(k), str(genome), str(cov), str(seed), str(out), design]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    res = json.load(open(out))
    assert res["superk"] and res["merged_lds"]
    r1, r2 = synth.make_reads(genome, cov, seed, err=0.02)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    oc = _oracle(codes, np.arange(n + 1, dtype=np.uint64) * np.uint64(L), k, oracle_built)
    ohist, _ = oc.occ_hist()
    assert res["hist_nonzero"] == {str(int(i)): int(ohist[i]) for i in np.nonzero(ohist)[0]}
    ok, ocn = oc.retain(1)
    assert res["n_distinct"] == len(ok)
    keys = np.concatenate([np.load(str(out) + f".keys{r}.npy").reshape(-1) for r in range(world)])
    counts = np.concatenate([np.load(str(out) + f".counts{r}.npy") for r in range(world)])
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(keys[order], ok.reshape(-1)) and np.array_equal(counts[order], ocn)


def test_superkmer_ragged_reads(gpu, oracle_built, monkeypatch):
    """Reads of very different lengths in one stream: empty, shorter than k, exactly k, 150 bp, 20 kbp (longer than a kernel
    tile, so its runs cross tile and lane boundaries), with N runs -- the super-k-mer path against the oracle."""
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    rng = np.random.default_rng(77)
    genome = rng.integers(0, 4, 60_000).astype(np.uint8)
    reads = []
    for i in range(4000):
        kind = i % 8
        if kind == 0:
            L = 0
        elif kind == 1:
            L = int(rng.integers(1, 31))
        elif kind == 2:
            L = 31
        elif kind == 3:
            L = 20_000 if i % 64 == 3 else 500
        else:
            L = 150
        s = int(rng.integers(0, len(genome) - max(L, 1)))
        r = genome[s:s + L].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        if L >= 40 and kind == 4:
            a = int(rng.integers(0, L - 8))
            r[a:a + int(rng.integers(1, 8))] = 4                 # a short run of Ns
        err = rng.random(L) < 0.01
        r[err & (r < 4)] = (r[err & (r < 4)] + 1 + rng.integers(0, 3, int((err & (r < 4)).sum()))) & 3
        reads.append(r)
    lens = np.array([len(r) for r in reads], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    codes = np.concatenate(reads)
    for k in (31, 20):
        oc = _oracle(codes, off, k, oracle_built)
        ohist, omax = oc.occ_hist()
        g = gpu(0)
        g.reads_append(codes, off)
        hist = g.make_kmer_read_distribution(k)
        assert g.kernel_ms("part1") > 0
        assert np.array_equal(hist, ohist) and g.max_occ == omax, k
        ok, ocn = oc.retain(1)
        assert g.retain(1) == len(ok)
        gk, gc = g.fetch_retained()
        assert np.array_equal(gk, ok) and np.array_equal(gc, ocn), k


def test_superkmer_random_differential():
    """tools/fuzz_count.py: random genomes (with low-complexity inserts), coverages, error / N rates, mixed read lengths, k in
    1..96 (one-, two- and three-word keys) and forced bucket counts -- the super-k-mer passes and the direct pass must agree on histogram, keys and counts."""
    import subprocess
    import sys
    from conftest import ROOT
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_count.py"), "30", "11"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "30 cases, 0 mismatches" in p.stdout


@pytest.mark.parametrize("form", [1, 2])
@pytest.mark.parametrize("bits", [3, 11, 15])
def test_superkmer_both_forms_of_the_bucket_kernel(form, bits, gpu, oracle_built, monkeypatch):
    """r3: sk_bucket2_kernel (the whole bucket in one round, block-wide piece list: the default) and sk_bucket_kernel (chunks, per-wave piece
    lists) against the oracle.  With 8 buckets a bucket holds thousands of distinct records: the second form's record table (1024 slots)
    fills up and records put their windows straight into the table of keys -- on the same slot sequence the pieces use (a first version
    used another one and counted four keys of 60.8 M twice); then the table of keys overflows and the bucket goes through the global table."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    monkeypatch.setitem(GpuCounter.default_options, "sk_bucket_bits", bits)
    monkeypatch.setitem(GpuCounter.default_options, "sk_bucket_form", form)
    r1, r2 = synth.make_reads(80_000, 60, seed=300 + bits, err=0.01, n_rate=0.001)
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 32, cutoffs=(1, 3))
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 27, cutoffs=(2,))


@pytest.mark.parametrize("div", [3, 1000000])
def test_superkmer_output_arrays_sized_by_estimate(div, gpu, oracle_built, monkeypatch):
    """r3: for large inputs the (key, count) arrays of the k <= 32 pass are sized for windows / 5 keys instead of one per window (device memory
    costs ~30 ms per GB); a pass that finds more distinct keys than that is repeated once with the number it counted.  `sk_out_div` applies
    the rule to a small input: windows / 3 does not hold a 20x set with 1 % errors, windows / 10^6 holds nothing -- same results as ever."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    monkeypatch.setitem(GpuCounter.default_options, "sk_out_div", div)
    r1, r2 = synth.make_reads(120_000, 20, seed=41, err=0.01, n_rate=0.001)
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 32, cutoffs=(1, 2))
    _superk_case(gpu, oracle_built, np.concatenate([r1, r2]), 21, cutoffs=(2,))


def test_counting_pass_replayed_as_a_graph(gpu, oracle_built, monkeypatch):
    """r3: the k <= 32 pass is captured on its first run over a set of buffers and replayed as one graph launch afterwards; a replay must
    give what the kernel-by-kernel pass gives (histogram head copied out with the scalars, tail fetched only when counts reach beyond it --
    here a 3000-fold repeat does), and anything that moves a buffer (more reads, another k) must end in a new capture, not a stale replay."""
    from platanus_b_amd import synth
    monkeypatch.setitem(GpuCounter.default_options, "count_path", COUNT_PATH["superk"])
    r1, r2 = synth.make_reads(60_000, 50, seed=77, n_rate=0.001)
    rep = np.tile(r1[:1], (3000, 1))                       # one read 3000 times: counts beyond the 1024 bins of the head
    codes = np.concatenate([r1, r2, rep])
    n, L = codes.shape
    off = np.arange(n + 1, dtype=np.uint64) * np.uint64(L)
    oc = _oracle(codes, off, 32, oracle_built)
    ohist, omax = oc.occ_hist()
    assert omax > 2000
    g = gpu(0)
    g.reads_append_matrix(codes)
    seen = []
    for it in range(4):
        hist = g.make_kmer_read_distribution(32)
        seen.append(g.kernel_ms("graph_replay"))
        assert np.array_equal(hist, ohist) and g.max_occ == omax, it
    assert seen[0] == 0 and seen[1] == 1 and seen[3] == 1           # captured by the first pass, replayed by the others
    g.set_option("graph", 0)
    assert np.array_equal(g.make_kmer_read_distribution(32), ohist) and g.kernel_ms("graph_replay") == 0
    g.set_option("graph", 1)
    o25 = _oracle(codes, off, 25, oracle_built)
    assert np.array_equal(g.make_kmer_read_distribution(25), o25.occ_hist()[0])                 # another k: other kernels, other buffers
    assert np.array_equal(g.make_kmer_read_distribution(32), ohist)
    g.reads_append_matrix(r2[:500])                        # more reads: the stream moves
    o2 = _oracle(np.concatenate([codes, r2[:500]]), np.arange(n + 501, dtype=np.uint64) * np.uint64(L), 32, oracle_built)
    for it in range(2):
        assert np.array_equal(g.make_kmer_read_distribution(32), o2.occ_hist()[0]), it
