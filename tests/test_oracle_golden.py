"""The CPU restatement (oracle/) against the golden vectors captured from the real reference binary
(tests/golden/make_golden.py).  The reference ships no tests or fixtures of its own,
so these files ARE the pin: byte-identical PREFIX_<k>merFrq.tsv, PREFIX_contig.fa and (k <= 32, -t 1)
PREFIX_kmer_occ.bin."""
import filecmp
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, case_reads


def _md5(path):
    with open(path, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def _args_to_kw(args):
    kw = dict(k0=32, step=10, n_cutoff=0, memory_gb=1, kmer_occ_only=False)
    it = iter(args)
    for a in it:
        if a == "-k": kw["k0"] = int(next(it))
        elif a == "-s": kw["step"] = int(next(it))
        elif a == "-n": kw["n_cutoff"] = int(next(it))
        elif a == "-m": kw["memory_gb"] = int(next(it))
        elif a == "-t": next(it)
        elif a == "-kmer_occ_only": kw["kmer_occ_only"] = True
        elif a == "-u": kw["bubble"] = next(it)
        else: raise AssertionError(a)
    return kw


CASE_RUNS = [("g5k", "multik"), ("g5k", "occ_n2"), ("g5k", "k21_s0"), ("c1", "multik"), ("c1", "occ"), ("c1", "s0"),
             ("c2", "multik_n2"), ("k45", "multik_k45"), ("c1", "u1"), ("het", "u1"), ("het", "u0")]


@pytest.mark.parametrize("name,run", CASE_RUNS)
def test_oracle_assemble_matches_reference_files(name, run, manifest, oracle_built, tmp_path):
    case = manifest[name]
    info = case["runs"][run]
    codes, off = case_reads(case)
    rc = oracle_built.assemble(codes.reshape(-1), off, str(tmp_path / "out"), **_args_to_kw(info["args"]))
    assert rc == 0
    for fn, meta in info["files"].items():
        got = tmp_path / fn
        assert got.exists(), fn
        assert os.path.getsize(got) == meta["bytes"], fn
        assert _md5(got) == meta["md5"], fn                       # every reference output, large ones by md5
        gold = os.path.join(GOLDEN, f"{name}.{run}.{fn}")
        if os.path.exists(gold):
            assert filecmp.cmp(got, gold, shallow=False), fn      # small ones byte for byte


def test_known_answer_vectors(oracle_built):
    """vectors computed against the reference headers."""
    from oracle.pyoracle import OracleCounter
    seq = "ACGTACGTTTGACCAGTCAGGCATTACGGATC"
    codes = np.array(["ACGT".index(c) for c in seq], dtype=np.uint8)
    oc = OracleCounter()
    oc.count(codes, np.array([0, 32], dtype=np.uint64), 32)
    keys, counts = oc.retain(1)
    assert keys.shape == (1, 1) and int(keys[0, 0]) == 0x1b1bf852d293c68d and int(counts[0]) == 1
    oc = OracleCounter()
    oc.count(codes[:21], np.array([0, 21], dtype=np.uint64), 21)
    keys, _ = oc.retain(1)
    assert int(keys[0, 0]) == min(0x0000006c6fe14b4a, 0x000001787ad01b1b)


def test_counter_invariants_and_edge_cases(oracle_built):
    """sum(hist) = distinct, sum(i*hist) = N-free windows; reads shorter than k, empty reads, N runs."""
    from oracle.pyoracle import OracleCounter
    rng = np.random.default_rng(5)
    lens = np.array([0, 1, 31, 32, 33, 150, 64, 500, 32, 0], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    codes = rng.integers(0, 4, int(off[-1])).astype(np.uint8)
    codes[off[5] + 40] = 4
    codes[off[7] + 100:off[7] + 103] = 4
    oc = OracleCounter()
    oc.count(codes, off, 32)
    hist, mx = oc.occ_hist()
    # brute force window count
    windows = 0
    for r in range(len(lens)):
        b = codes[off[r]:off[r + 1]]
        for i in range(0, max(len(b) - 32 + 1, 0)):
            windows += int(not (b[i:i + 32] == 4).any())
    assert int((np.arange(65535, dtype=np.uint64) * hist).sum()) == windows
    keys, counts = oc.file_order()
    assert int(hist.sum()) == len(keys)
    lh = oc.length_hist(501)
    assert int(lh[0]) == 2 and int(lh[32]) == 2 and int(lh[500]) == 1 and int(lh.sum()) == len(lens)


def test_kmer_schedule(oracle_built):
    """extendKmer / decreaseCoverageCutoff / calcLogProbabilityJoin (assemble.cpp:657-736) against schedules the
    survey logged from the reference: 50x run -> [32,42,52,62,72,75] / [3,2,1,1,1,1]."""
    import math
    ks, cs = oracle_built.extend_kmer(math.log(1.0 - 10.0 ** -10.0), 33.718 * 150 / (150 - 32 + 1.0) * (150 - 32 + 1.0) / 150 * 150 / (150 - 32 + 1.0),
                                      150.0, 1, 0.5, 32, 3, 10)
    assert ks[0] == 32 and ks[-1] == 75 and all(b > a for a, b in zip(ks, ks[1:]))
    assert cs[0] == 3 and cs[-1] == 1 and all(b <= a for a, b in zip(cs, cs[1:]))


def test_repeat_mode_has_no_reference_output(tmp_path, oracle_built):
    """`-repeat` is listed among the CLI options (assemble.cpp:55-70) but the reference binary built from v1.3.2 dies with
    SIGSEGV in that mode on every input tried (here the 5 kbp case): the final stage's saveGraph runs
    pickupReadMatchedEdgeKmer on a table of size 0.  The oracle restates the same control flow (three cutoff passes +
    mergeContig) and dies at the same place.  So there is no reference output the device path could be identical to; it
    rejects the option with an error instead."""
    import signal
    import subprocess
    from conftest import ROOT
    from platanus_b_amd import synth
    synth.write_pair_files(str(tmp_path / "in"), 5000, 30, 3, err=0.01, n_rate=0.002)
    bins = [os.path.join(ROOT, "oracle", "pb_oracle")]
    ref = os.path.join(ROOT, "oracle", "_ref", "platanus_b")
    if os.path.exists(ref):
        bins.append(ref)
    for b in bins:
        p = subprocess.run([b, "assemble", "-k", "32", "-t", "1", "-m", "1", "-repeat", "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path,
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == -signal.SIGSEGV, (b, p.returncode)
        assert "extracting reads (containing kmer used in contig assemble)..." in p.stderr.splitlines()[-1]
        assert not os.path.exists(tmp_path / "out_contig.fa")


def test_oracle_cli_equals_real_reference_on_random_options(oracle_built):
    """tools/fuzz_oracle_vs_ref.py, a slice: where the real reference binary was built (oracle/_ref, build container only) the oracle
    CLI must give the same exit status and files for random read sets and random -k -s -u -n -c -d -K -a -e."""
    import subprocess
    import sys
    ref = os.path.join(ROOT, "oracle", "_ref", "platanus_b")
    if not os.path.exists(ref):
        pytest.skip("oracle/_ref/platanus_b not built (no /root/reference here)")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_oracle_vs_ref.py"), "4", "77"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "4 cases, 0 mismatches" in p.stdout
