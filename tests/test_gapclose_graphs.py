"""the local de Bruijn graphs of gap closing (counter.h:1015-1056 countKmerForGapClose, graph.h:2116-2345
makeBruijnGraphForGapClose), for a batch of gaps.

Pinning: tests/golden/gapclose/*.txt are dumps of the REFERENCE's own tables after its own two functions ran on the inputs of
tests/gapclose_cases.py (oracle/ref_harness/gapclose_ref.cpp compiled against the reference's sources, tests/golden/make_gapclose_golden.py).
CPU: the oracle's restatement (oracle/src/gapclose.hpp) must leave the same tables in the same iteration order.  GPU: pb_gap_graphs_build /
_fetch must return the oracle's nodes, field for field, in the oracle's (= the reference's insertion) order."""
import os

import numpy as np
import pytest

import gapclose_cases
from conftest import ROOT

GOLD = os.path.join(ROOT, "tests", "golden", "gapclose")


@pytest.fixture(scope="module")
def gpu():
    from platanus_b_amd.binding import GpuCounter
    return GpuCounter


def _golden(name):
    from oracle import pyoracle
    return pyoracle.split_gap_dump(open(os.path.join(GOLD, name + ".txt")).read())


@pytest.mark.parametrize("name", sorted(gapclose_cases.CASES))
def test_oracle_gap_graphs_against_the_reference_tables(name, oracle_built):
    from oracle import pyoracle
    k, mc, first, off, codes = gapclose_cases.make_gaps(name)
    nodes, dumps = pyoracle.gap_graphs(k, mc, first, off, codes)
    gold = _golden(name)
    assert len(dumps) == len(gold) == len(first) - 1
    for g, (a, b) in enumerate(zip(dumps, gold)):
        assert a == b, f"{name}: tables of gap {g} differ from the reference's"
    assert int(nodes["node_first"][-1]) == len(nodes["kind"]) == sum(d.count("\n") for d in gold)
    assert dumps[2] == ""                                    # (the gap without sequences)
    if pyoracle.have_reference() and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "gapclose_ref")):
        assert pyoracle.gap_graphs_reference(k, mc, first, off, codes) == gold          # the committed dumps are what the reference prints here and now


def test_oracle_gap_graphs_saturation_and_threshold(oracle_built):
    """One 24-mer 70,000 times (count saturates at 65534, counter.h:1051), and a threshold above every count (no node at all)."""
    from oracle import pyoracle
    rng = np.random.default_rng(5)
    r = rng.integers(0, 4, 24).astype(np.uint8)
    codes = np.tile(r, 70_000)
    off = np.arange(70_001, dtype=np.uint64) * 24
    nodes, dumps = pyoracle.gap_graphs(24, 2, np.array([0, 70_000], dtype=np.uint64), off, codes)
    assert list(nodes["kind"]) == [1] and int(nodes["cov"][0]) == 65534 and int(nodes["length"][0]) == 1
    nodes, dumps = pyoracle.gap_graphs(24, 65535, np.array([0, 70_000], dtype=np.uint64), off, codes)
    assert len(nodes["kind"]) == 0 and dumps == [""]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(gapclose_cases.CASES))
def test_device_gap_graphs_against_the_oracle(name, gpu, oracle_built):
    from oracle import pyoracle
    k, mc, first, off, codes = gapclose_cases.make_gaps(name)
    want, _ = pyoracle.gap_graphs(k, mc, first, off, codes)
    g = gpu(0)
    got = g.gap_graphs(k, mc, first, off, codes)
    for field in ("node_first", "kind", "cov", "out", "length", "word_off", "words"):
        assert np.array_equal(got[field], want[field]), f"{name}: {field} differs from the oracle"
    assert g.kernel_ms("gap_graphs") > 0


@pytest.mark.gpu
def test_device_gap_graphs_edges(gpu, oracle_built):
    """No gap at all; gaps without a countable window; the saturating count; k = 32 with a T^32 window (the forward key that is the one-word
    table's empty mark); k = 64 / 96 / 128 / 160 with poly-T reads (keys of all ones in every word)."""
    from oracle import pyoracle
    g = gpu(0)
    empty = g.gap_graphs(21, 2, np.zeros(1, dtype=np.uint64), np.zeros(1, dtype=np.uint64), np.zeros(0, dtype=np.uint8))
    assert len(empty["kind"]) == 0 and list(empty["node_first"]) == [0]
    rng = np.random.default_rng(7)
    r = rng.integers(0, 4, 24).astype(np.uint8)
    codes = np.tile(r, 70_000)
    off = np.arange(70_001, dtype=np.uint64) * 24
    first = np.array([0, 70_000], dtype=np.uint64)
    for mc in (2, 65535):
        want, _ = pyoracle.gap_graphs(24, mc, first, off, codes)
        got = g.gap_graphs(24, mc, first, off, codes)
        for field in want:
            assert np.array_equal(got[field], want[field]), field
    for k in (32, 64, 96, 128, 160):
        genome = rng.integers(0, 4, 700).astype(np.uint8)
        genome[200:200 + k + 40] = 3                          # poly-T longer than k (and a few A after it)
        genome[200 + k + 40:200 + k + 60] = 0
        seqs = []
        for _ in range(60):
            s = int(rng.integers(0, 700 - 300))
            seqs.append(genome[s:s + 300].copy())
        seqs.append(np.full(k, 4, dtype=np.uint8))
        first = np.array([0, 0, len(seqs), len(seqs)], dtype=np.uint64)           # an empty gap on either side
        off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
        codes = np.concatenate(seqs)
        want, _ = pyoracle.gap_graphs(k, 2, first, off, codes)
        got = g.gap_graphs(k, 2, first, off, codes)
        assert len(want["kind"]) > 0
        for field in want:
            assert np.array_equal(got[field], want[field]), f"k={k}: {field}"


@pytest.mark.gpu
def test_device_gap_graphs_many_gaps(gpu, oracle_built):
    """3000 gaps in one call (the batch is the point: the reference takes them one by one)."""
    from oracle import pyoracle
    rng = np.random.default_rng(11)
    seqs, first = [], [0]
    for _ in range(3000):
        G = int(rng.integers(60, 400))
        genome = rng.integers(0, 4, G).astype(np.uint8)
        for _ in range(int(rng.integers(0, 30))):
            L = int(min(100, G))
            s = int(rng.integers(0, G - L + 1))
            r = genome[s:s + L].copy()
            e = rng.random(L) < 0.01
            r[e] = (r[e] + 1) & 3
            seqs.append(r)
        first.append(len(seqs))
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    codes = np.concatenate(seqs)
    first = np.array(first, dtype=np.uint64)
    want, _ = pyoracle.gap_graphs(25, 2, first, off, codes)
    got = gpu(0).gap_graphs(25, 2, first, off, codes)
    for field in want:
        assert np.array_equal(got[field], want[field]), field


def test_host_mirror_of_the_gap_graphs_builds(tmp_path):
    """platanus_b_amd/host/gap_graphs.hpp (GapBatch -> pb_gap_graphs_build / _fetch -> the reference's tables per gap) compiles against the
    library and its table step runs without a GPU."""
    import subprocess
    import __graft_entry__
    __graft_entry__.build()
    lib_dir = os.path.join(ROOT, "platanus_b_amd")
    exe = str(tmp_path / "gap_graphs_host")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-sign-compare", "-o", exe, os.path.join(ROOT, "tests", "native", "gap_graphs_host.cpp"), "-L" + lib_dir,
                    "-lpbgpu", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    assert subprocess.run([exe], timeout=60).returncode == 0


@pytest.mark.parametrize("name", sorted(gapclose_cases.ASSEMBLE_CASES))
def test_oracle_gap_assembly_against_the_reference_tables(name, oracle_built):
    """The graph part of GapCloseDBG<M, N>::gapAssemble (gapCloseDBG.h:416-436: low k and high k counted, built and branch-cut, mergeGraph =
    saveLargeKmerForGapClose of both graphs into the high table, the high graph built again, branch cut, bubble crush): the restatement's
    low and high tables must be the reference's own, node for node in iteration order (dumps of GapCloseDBG's tables, incl. gap_close's
    own pair k = 20 / 80)."""
    from oracle import pyoracle
    kl, kh, mc, case = gapclose_cases.ASSEMBLE_CASES[name]
    _, _, first, off, codes = gapclose_cases.make_gaps(case)
    dumps = pyoracle.gap_assemble(kl, kh, mc, first, off, codes)
    gold = _golden(name)
    assert len(dumps) == len(gold)
    for g, (a, b) in enumerate(zip(dumps, gold)):
        assert a == b, f"{name}: graphs of gap {g} differ from the reference's"
    assert sum(d.count("\nS ") for d in gold) > 0


@pytest.fixture(scope="module")
def gap_assemble_host(tmp_path_factory):
    import subprocess
    import __graft_entry__
    __graft_entry__.build()
    lib_dir = os.path.join(ROOT, "platanus_b_amd")
    exe = str(tmp_path_factory.mktemp("native") / "gap_assemble_host")
    subprocess.run(["g++", "-std=c++17", "-O2", "-Wall", "-Wno-sign-compare", "-o", exe, os.path.join(ROOT, "tests", "native", "gap_assemble_host.cpp"), "-L" + lib_dir,
                    "-lpbgpu", "-lpthread", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return exe


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(gapclose_cases.ASSEMBLE_CASES))
def test_device_gap_assembly_against_the_reference_tables(name, gap_assemble_host, tmp_path):
    """The same through the product: host/gap_graphs.hpp gap_assemble_batch -- pb_gap_graphs_build at both k for all gaps at once, the
    node-level cleaning on the host, pb_gap_graphs_build_from_nodes for the merged table -- must leave the reference's tables."""
    import struct
    import subprocess
    from oracle import pyoracle
    kl, kh, mc, case = gapclose_cases.ASSEMBLE_CASES[name]
    _, _, first, off, codes = gapclose_cases.make_gaps(case)
    inp = tmp_path / "gaps.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<IqQ", kl, mc, len(first) - 1))
        f.write(first.tobytes())
        f.write(struct.pack("<Q", len(off) - 1))
        f.write(off.tobytes())
        f.write(codes[:int(off[-1])].tobytes())
    p = subprocess.run([gap_assemble_host, str(inp), str(kh)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = pyoracle.split_gap_dump(p.stdout)
    gold = _golden(name)
    assert len(got) == len(gold)
    for g, (a, b) in enumerate(zip(got, gold)):
        assert a == b, f"{name}: graphs of gap {g} differ from the reference's"


@pytest.mark.parametrize("name", sorted(gapclose_cases.CLOSE_CASES))
def test_oracle_gap_closing_decisions_against_the_reference(name, oracle_built):
    """What the reference decides about a gap from its final graph and the gap's flanks -- closedGap, closedGapPartial, extendEdgeEntity
    (gapCloseDBG.h:479-741; the order of gapClose.cpp:716-760 without the OLC fallback; -vd 32, -ed 0.01) -- restated in
    platanus_b_amd/host/gap_graphs.hpp (GapCloser): closed / partly closed / extended / nothing, the closing sequence base for base."""
    from oracle import pyoracle
    kl, kh, first, off, codes, flanks = gapclose_cases.make_gaps_with_flanks(name)
    dumps = pyoracle.gap_assemble(kl, kh, 2, first, off, codes, flanks)
    gold = _golden(name)
    assert len(dumps) == len(gold)
    for g, (a, b) in enumerate(zip(dumps, gold)):
        assert a == b, f"{name}: gap {g} differs from the reference's"
    text = "".join(gold)
    for tag in ("\nC 1", "\nC 0", "\nP 1", "\nE 1", "X\n"):
        assert tag in text, f"{name}: no {tag!r} case"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(gapclose_cases.CLOSE_CASES))
def test_device_gap_closing_decisions_against_the_reference(name, gap_assemble_host, tmp_path):
    """The same through the product: the graphs of all gaps by gap_assemble_batch (device + host cleaning), the decisions by GapCloser."""
    import struct
    import subprocess
    from oracle import pyoracle
    kl, kh, first, off, codes, flanks = gapclose_cases.make_gaps_with_flanks(name)
    inp, fl = tmp_path / "gaps.bin", tmp_path / "flanks.bin"
    with open(inp, "wb") as f:
        f.write(struct.pack("<IqQ", kl, 2, len(first) - 1))
        f.write(first.tobytes())
        f.write(struct.pack("<Q", len(off) - 1))
        f.write(off.tobytes())
        f.write(codes[:int(off[-1])].tobytes())
    fl.write_bytes(flanks)
    p = subprocess.run([gap_assemble_host, str(inp), str(kh), str(fl)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = pyoracle.split_gap_dump(p.stdout)
    gold = _golden(name)
    assert len(got) == len(gold)
    for g, (a, b) in enumerate(zip(got, gold)):
        assert a == b, f"{name}: gap {g} differs from the reference's"
