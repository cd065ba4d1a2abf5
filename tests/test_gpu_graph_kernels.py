"""Direct C-ABI parity tests of the graph-stage and multi-k kernels (K6 pb_graph_adjacency, K7 pb_graph_chains / pb_graph_nodes,
M3 pb_reads_filter, M5 pb_contig_table_build, M6 pb_count_considering_contigs, pb_occ_lookup) for 1- to 5-word keys.

The checker is a plain-Python restatement on arbitrary-precision integers of what the reference computes at these points
(graph.h:352-380 neighbour probes, graph.h:406-542 unitig walk, counter.h:869-910 read filter, graph.h:1738-1767 contig k-mers,
counter.h:827-861 missing k-mers, kmer_divide.cpp:150-197 look-ups); inputs are small genomes with copied and inverted segments
(cycles, hairpins), N runs and errors, so every structural case of the kernels occurs.  Integer work: the bar is bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from platanus_b_amd.binding import GpuCounter, words_for_k  # noqa: E402

JUNCTION, CYCLE, NONE = 0xFFFFFFFF, 0xFFFFFFFE, 0xFFFFFFFF


def _rc(x, k):
    r = 0
    for _ in range(k):
        r = (r << 2) | (3 - (x & 3))
        x >>= 2
    return r


def _canon(x, k):
    """(canonical key, 1 if x is the reverse complement of its canonical key)"""
    r = _rc(x, k)
    return (r, 1) if r < x else (x, 0)


def _key_ints(keys):
    return [sum(int(w) << (64 * i) for i, w in enumerate(row)) for row in keys]


def _make_reads(k, seed, G=4000, cov=30, err=0.004, n_rate=0.002, read_len=160):
    """Reads (uint8 codes, offsets) of a genome with a copied and an inverted segment and a short tandem repeat."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, G).astype(np.uint8)
    g[1500:1800] = g[300:600]                                   # repeat: junctions
    g[2500:2800] = (3 - g[900:1200])[::-1]                      # inverted repeat: both strands of a k-mer in one walk
    g[3300:3380] = np.resize(rng.integers(0, 4, 7).astype(np.uint8), 80)   # tandem repeat: cycles for small k
    L = max(read_len, k + 10)
    reads = []
    for _ in range(int(G * cov / L)):
        s = int(rng.integers(0, G - L + 1))
        r = g[s:s + L].copy()
        if rng.random() < 0.5:
            r = (3 - r)[::-1]
        e = rng.random(L) < err
        r[e] = (r[e] + 1 + rng.integers(0, 3, int(e.sum()))) & 3
        r[rng.random(L) < n_rate] = 4
        reads.append(r)
    codes = np.concatenate(reads)
    off = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(L)
    return g, codes, off


def _windows(read, k):
    """(position, forward k-mer as int) of every N-free window of one read"""
    x, run, mask = 0, 0, (1 << (2 * k)) - 1
    for i, c in enumerate(read):
        c = int(c)
        if c > 3:
            run, x = 0, 0
            continue
        x = ((x << 2) | c) & mask
        run += 1
        if run >= k:
            yield i - k + 1, x


@pytest.mark.parametrize("k", [21, 32, 45, 75, 110, 150])
def test_adjacency_and_chains_against_plain_loops(k):
    _, codes, off = _make_reads(k, 1000 + k)
    g = GpuCounter(0)
    g.reads_append(codes, off)
    g.make_kmer_read_distribution(k)
    n = g.retain(2)
    keys, counts = g.fetch_retained()
    K = _key_ints(keys)
    index = {x: i for i, x in enumerate(K)}
    assert len(index) == n and n > 500
    mask = (1 << (2 * k)) - 1
    adj, nbr = g.graph_adjacency()

    def neighbours(x, right):
        """[(base, index, flip)] of the retained successors (right) / predecessors of the k-mer x (not nec. canonical)"""
        out = []
        for b in range(4):
            y = ((x << 2) | b) & mask if right else (x >> 2) | (b << (2 * (k - 1)))
            cy, flip = _canon(y, k)
            if cy in index:
                out.append((b, index[cy], flip))
        return out

    # ---- K6: the eight presence probes per retained k-mer (graph.h:352-380) and the unique neighbour of either side
    for i, x in enumerate(K):
        R, Lf = neighbours(x, True), neighbours(x, False)
        want = (sum(1 << b for b, _, _ in Lf) << 4) | sum(1 << b for b, _, _ in R)
        assert adj[i] == want, (i, hex(x))
        assert nbr[i, 0] == (((R[0][1] << 1) | R[0][2]) if len(R) == 1 else NONE), (i, "right")
        assert nbr[i, 1] == (((Lf[0][1] << 1) | Lf[0][2]) if len(Lf) == 1 else NONE), (i, "left")

    # ---- K7: chains (the unitig walk of graph.h:406-542 as list ranking)
    ch = g.graph_chains()
    chain_of, orient = ch["chain_of"], ch["orient"]
    junction = np.array([bin(a >> 4).count("1") > 1 or bin(a & 15).count("1") > 1 for a in adj])
    assert ((chain_of == JUNCTION) == junction).all()
    members = np.bincount(chain_of[chain_of < CYCLE].astype(np.int64), minlength=ch["n_chains"])
    assert ch["n_chains"] > 0 and (members == ch["len"]).all()
    for c in range(ch["n_chains"]):
        ln = int(ch["len"][c])
        L = ln + k - 1
        w0 = int(ch["seq_woff"][c])
        # Straight::seq layout (graph.h:67-71): position p counts from the B (right) end, 32 positions per word
        S = [(int(ch["seq"][w0 + p // 32]) >> (2 * (p % 32))) & 3 for p in range(L - 1, -1, -1)]
        x = 0
        for b in S[:k]:
            x = (x << 2) | b
        total, seen = 0, []
        for j in range(ln):
            if j:
                x = ((x << 2) | S[k - 1 + j]) & mask
            cx, flip = _canon(x, k)
            assert cx in index, (c, j)
            i = index[cx]
            assert chain_of[i] == c and orient[i] == flip, (c, j)
            seen.append((i, flip, x))
            total += int(counts[i])
        assert len({i for i, _, _ in seen}) == ln
        assert total == int(ch["sum"][c])
        assert int(ch["end_a"][c]) == (seen[0][0] << 1 | seen[0][1]) and int(ch["end_b"][c]) == (seen[-1][0] << 1 | seen[-1][1])
        # maximal: what lies beyond either end is nothing, several k-mers, a junction or a k-mer on a cycle / hairpin
        for x_end, right in ((seen[-1][2], True), (seen[0][2], False)):
            nb = neighbours(x_end, right)
            if len(nb) == 1:
                assert chain_of[nb[0][1]] in (JUNCTION, CYCLE), (c, right)
    # ---- compact node export: junctions, cycle k-mers and chain ends, with everything the host replay reads
    nd = g.graph_nodes()
    node_at = {int(i): s for s, i in enumerate(nd["idx"])}
    ends = set()
    for c in range(ch["n_chains"]):
        ends.add(int(ch["end_a"][c]) >> 1)
        ends.add(int(ch["end_b"][c]) >> 1)
    must = {i for i in range(n) if chain_of[i] in (JUNCTION, CYCLE)} | ends
    assert must <= set(node_at)
    for i, s in node_at.items():
        meta = int(nd["meta"][s])
        assert meta & 0xFF == adj[i] and (meta >> 8) & 1 == (orient[i] if chain_of[i] < CYCLE else (meta >> 8) & 1) and meta >> 16 == counts[i]
        assert nd["chain"][s] == chain_of[i]
        assert (nd["nbr"][s] == nbr[i]).all()
        assert _key_ints(nd["keys"][s:s + 1])[0] == K[i]
        if chain_of[i] == JUNCTION:
            R, Lf = {b: (j << 1 | f) for b, j, f in neighbours(K[i], True)}, {b: (j << 1 | f) for b, j, f in neighbours(K[i], False)}
            for b in range(4):
                assert nd["nb8"][s, b] == R.get(b, NONE) and nd["nb8"][s, 4 + b] == Lf.get(b, NONE)


@pytest.mark.parametrize("k,k2", [(21, 29), (32, 42), (42, 52), (75, 96), (96, 106), (128, 140)])
def test_read_filter_contig_table_and_missing_kmers(k, k2, oracle_built):
    """One multi-k round through the C-ABI: pickupReadMatchedEdgeKmer with graph k-mers of length k, the contig-derived table and
    the count of read k-mers absent from it at the next k (k2)."""
    genome, codes, off = _make_reads(k2, 2000 + k)
    nreads = len(off) - 1
    reads = [codes[int(off[i]):int(off[i + 1])] for i in range(nreads)]
    g = GpuCounter(0)
    g.reads_append(codes, off)
    g.make_kmer_read_distribution(k)
    g.retain(3)
    keys, _ = g.fetch_retained()
    rng = np.random.default_rng(k)
    # ---- M3: edge k-mers = a thin sample of the retained keys plus keys that occur nowhere
    pick = keys[rng.random(len(keys)) < 0.004]
    fake = rng.integers(0, 1 << 40, (5, keys.shape[1]), dtype=np.uint64)
    edge = np.concatenate([pick, fake])
    eset = set(_key_ints(edge))
    want_alive = np.array([any(_canon(x, k)[0] in eset for _, x in _windows(r, k)) for r in reads])
    assert 0 < want_alive.sum() < nreads
    assert g.reads_filter(k, edge) == int(want_alive.sum())
    # cumulative (readFP is replaced by the filtered file, counter.h:908-909): a second filter only sees the survivors
    edge2 = keys[rng.random(len(keys)) < 0.01]
    eset2 = set(_key_ints(edge2))
    want2 = want_alive & np.array([any(_canon(x, k)[0] in eset2 for _, x in _windows(r, k)) for r in reads])
    assert g.reads_filter(k, edge2) == int(want2.sum())
    # ---- M5: contig strings with one value each; value 0 writes nothing; a k-mer in two contigs keeps the larger value
    contigs, values = [], []
    for s, ln, v in ((100, 700, 17), (500, 400, 9), (1500, 300, 40), (2600, k2 + 3, 0), (3000, k2 - 1, 5), (3100, k2, 65535)):
        c = genome[s:s + ln]
        contigs.append(c if len(contigs) % 2 == 0 else (3 - c)[::-1])
        values.append(v)
    cb = np.concatenate(contigs)
    co = np.concatenate([[0], np.cumsum([len(c) for c in contigs])]).astype(np.uint64)
    g.contig_table_build(k2, cb, co, np.array(values, dtype=np.uint16))
    table = {}
    for c, v in zip(contigs, values):
        if v:
            for _, x in _windows(c, k2):
                cx = _canon(x, k2)[0]
                table[cx] = max(table.get(cx, 0), v)
    # ---- M6: k-mers of the surviving reads that the contigs do not have, counted; plus the table's own entries
    hist = g.count_considering_contigs(k2)
    expect = dict(table)
    novel = {}
    for r, alive in zip(reads, want2):
        if alive:
            for _, x in _windows(r, k2):
                cx = _canon(x, k2)[0]
                if cx not in table:
                    novel[cx] = novel.get(cx, 0) + 1
    assert novel and not (set(novel) & set(table))
    expect.update(novel)
    gk, gc = g.fetch_all()
    got = dict(zip(_key_ints(gk), (int(c) for c in gc)))
    assert len(got) == len(gk)
    assert got == expect
    want_hist = np.zeros(65535, dtype=np.uint64)
    for c in expect.values():
        want_hist[min(c, 65534)] += 1
    assert np.array_equal(hist, want_hist)
    # the result feeds retain(): cutoff 2 keeps the table entries >= 2 and the read k-mers seen twice
    n2 = g.retain(2)
    rk, rc = g.fetch_retained()
    want_ret = sorted((x, min(c, 65534)) for x, c in expect.items() if c >= 2)
    assert n2 == len(want_ret) and list(zip(_key_ints(rk), (int(c) for c in rc))) == want_ret
    # pb_reads_restore = allReadFP (assemble.cpp:164-171): every read counts again
    g.reads_restore()
    g.make_kmer_read_distribution(k2)
    oc = oracle_built.OracleCounter()
    oc.count(codes, off, k2)
    assert np.array_equal(g.occ_hist, oc.occ_hist()[0])


@pytest.mark.parametrize("k", [25, 32, 60, 150])
def test_occurrence_lookup(k):
    """pb_occ_table_load + pb_occ_lookup (ContigDivider::getOccurrenceArray, kmer_divide.cpp:150-197): occurrence of the canonical
    k-mer at every position; 0 for absent k-mers, windows over an N and positions past len - k."""
    genome, codes, off = _make_reads(k, 3000 + k)
    g = GpuCounter(0)
    g.reads_append(codes, off)
    g.make_kmer_read_distribution(k)
    g.retain(2)
    keys, counts = g.fetch_retained()
    table = dict(zip(_key_ints(keys), (int(c) for c in counts)))
    g.occ_table_load(k, keys, counts)
    rng = np.random.default_rng(k)
    seqs = [genome[200:900].copy(), (3 - genome[1000:1400])[::-1].copy(), genome[10:10 + k].copy(), genome[50:50 + k - 1].copy(), rng.integers(0, 4, 300).astype(np.uint8)]
    seqs[0][[100, 101, 400]] = 4
    sb = np.concatenate(seqs)
    so = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    got = g.occ_lookup(sb, so)
    want = np.zeros(len(sb), dtype=np.uint16)
    for s, base in zip(seqs, so[:-1]):
        for p, x in _windows(s, k):
            want[int(base) + p] = table.get(_canon(x, k)[0], 0)
    assert want.any()
    assert np.array_equal(got, want)


def test_edit_distances_against_plain_dp():
    """pb_edit_distances = crushBubble's pairwiseAlignment (graph.h:712-746): unit-cost edit distance of pairs of base strings."""
    rng = np.random.default_rng(5)
    seqs = []
    for _ in range(150):
        a = rng.integers(0, 4, int(rng.integers(1, 400))).astype(np.uint8)
        b = a.copy()
        for _ in range(int(rng.integers(0, 12))):                     # substitutions, insertions, deletions
            kind, p = int(rng.integers(0, 3)), int(rng.integers(0, max(len(b), 1)))
            if kind == 0 and len(b):
                b[p] = (b[p] + 1 + rng.integers(0, 3)) & 3
            elif kind == 1:
                b = np.insert(b, p, rng.integers(0, 4))
            elif len(b) > 1:
                b = np.delete(b, p)
        if rng.random() < 0.2:
            b = rng.integers(0, 4, int(rng.integers(1, 300))).astype(np.uint8)
        seqs += [a, b.astype(np.uint8)]
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    g = GpuCounter(0)
    got = g.edit_distances(np.concatenate(seqs), off)
    for p in range(len(seqs) // 2):
        a, b = seqs[2 * p], seqs[2 * p + 1]
        prev = list(range(len(b) + 1))
        for m in range(len(a)):
            cur = [m + 1] + [0] * len(b)
            for n in range(len(b)):
                cur[n + 1] = prev[n] if a[m] == b[n] else min(prev[n], prev[n + 1], cur[n]) + 1
            prev = cur
        assert int(got[p]) == prev[len(b)], p


def test_edit_distances_long_branches():
    """Beyond one strip and beyond the wave kernel's LDS: a 100-base branch against a 20,000-base one (313 strips of 64 columns), and two
    branches of 9,000 bases (both sides over ED_MAXROWS = 8192: the lane-per-pair kernel with its rows in global memory); distances known by
    construction (substitutions far apart in random sequence; a prefix against the whole)."""
    rng = np.random.default_rng(17)
    a = rng.integers(0, 4, 100).astype(np.uint8)
    b = np.concatenate([a, rng.integers(0, 4, 19_900).astype(np.uint8)])
    c = rng.integers(0, 4, 9_000).astype(np.uint8)
    d = c.copy()
    for pos in (100, 2_000, 4_500, 7_000, 8_900):
        d[pos] = (d[pos] + 1) & 3
    e = rng.integers(0, 4, 8_192).astype(np.uint8)                    # exactly ED_MAXROWS rows: still the wave kernel
    f = np.delete(e, [10, 5_000])
    seqs = [a, b, c, d, b, a, e, f]
    off = np.concatenate([[0], np.cumsum([len(x) for x in seqs])]).astype(np.uint64)
    g = GpuCounter(0)
    got = g.edit_distances(np.concatenate(seqs), off)
    assert [int(x) for x in got] == [19_900, 5, 19_900, 2]


@pytest.mark.parametrize("k", [21, 32, 45, 100])
def test_node_cleaning_decisions_against_plain_loops(k):
    """pb_nodes_clean_decide on random node sets: junctions with tips, real branches and junction-junction links on both sides; the
    checker restates cutBranch (graph.h:829-926) and deleteErroneousStraightNode's predicate (graph.h:2048-2049) with dictionaries."""
    rng = np.random.default_rng(k)
    W = words_for_k(k)
    mask = (1 << (2 * k)) - 1
    rk = lambda: int.from_bytes(rng.bytes(8 * W), "little") & mask
    succ = lambda x, b: ((x << 2) | b) & mask
    pred = lambda x, b: (x >> 2) | (b << (2 * (k - 1)))
    junctions, straights = {}, []                                     # key -> [cov, out]; dict(head, tail, length, cov, out)
    for _ in range(400):
        jk = rk()
        jout = 0
        for side in (0, 1):                                           # 0: right links (heads of straights leaving it), 1: left links (tails)
            for b in rng.permutation(4)[:int(rng.integers(0, 4))]:
                b = int(b)
                nk = succ(jk, b) if side == 0 else pred(jk, b)
                kind = rng.random()
                jout |= 1 << (b if side == 0 else b + 4)
                if kind < 0.15:
                    junctions.setdefault(nk, [int(rng.integers(1, 200)), 0])     # a junction as neighbour
                    continue
                length = int(rng.integers(1, 3 * k)) if kind < 0.7 else int(rng.integers(1, k + 1))
                other = 0 if rng.random() < 0.6 else 1 << int(rng.integers(0, 4))   # a tip has nothing on its far side
                base = 1 << (jk >> (2 * (k - 1)) if side == 0 else jk & 3)          # flag of the base that leads back to the junction
                if side == 0:
                    straights.append(dict(head=nk, tail=rk(), length=length, cov=int(rng.integers(1, 120)), out=(base << 4) | other))
                else:
                    straights.append(dict(head=rk(), tail=nk, length=length, cov=int(rng.integers(1, 120)), out=(other << 4) | base))
        junctions[jk] = [int(rng.integers(1, 200)), jout]
    head_of = {s["head"]: i for i, s in enumerate(straights)}
    tail_tab = {s["tail"]: i for i, s in enumerate(straights)}
    assert len(head_of) == len(straights) and len(tail_tab) == len(straights)
    jlist = list(junctions.items())
    th, len_cut, cov_cut = 0.5, 2 * k, 30

    def words(x):
        return [(x >> (64 * w)) & 0xFFFFFFFFFFFFFFFF for w in range(W)]

    want = []
    for i, s in enumerate(straights):
        cut = False
        lf, rf = s["out"] >> 4, s["out"] & 15
        if s["length"] <= k and (lf != 0) != (rf != 0):
            left = lf != 0
            b0 = {1: 0, 2: 1, 4: 2, 8: 3}[lf if left else rf]
            jk = pred(s["head"], b0) if left else succ(s["tail"], b0)
            if jk in junctions:
                mx = 0
                for b in range(4):
                    if not junctions[jk][1] & (1 << (b if left else b + 4)):
                        continue
                    nk = succ(jk, b) if left else pred(jk, b)
                    t = (head_of if left else tail_tab).get(nk)
                    if t is not None and t != i and straights[t]["cov"] > mx:
                        mx = straights[t]["cov"]
                        continue
                    if nk in junctions and junctions[nk][0] > mx:
                        mx = junctions[nk][0]
                cut = not (s["cov"] > mx * th)
        err = (not cut) and s["cov"] < cov_cut and s["length"] < len_cut
        want.append((1 if cut else 0) | (2 if err else 0))
    g = GpuCounter(0)
    got = g.nodes_clean_decide(k, [w for s in straights for w in words(s["head"])], [w for s in straights for w in words(s["tail"])],
                               [s["length"] for s in straights], [s["cov"] for s in straights], [s["out"] for s in straights],
                               [w for s in straights for w in words(s["tail"])], list(range(len(straights))),
                               [w for jk, _ in jlist for w in words(jk)], [v[0] for _, v in jlist], [v[1] for _, v in jlist], th, len_cut, cov_cut)
    assert sum(1 for w_ in want if w_ & 1) > 20 and sum(1 for w_ in want if w_ & 2) > 20
    assert list(got) == want
