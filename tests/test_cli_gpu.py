"""End-to-end drop-in check: the MI355X build of `platanus_b assemble` (platanus_b_amd/bin/platanus_b = C++ host over
libpbgpu.so) must write the same PREFIX_contig.fa / PREFIX_<k>merFrq.tsv as the real reference did for the golden
cases (tests/golden, captured by make_golden.py from oracle/_ref/platanus_b)."""
import filecmp
import hashlib
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")


def _md5(p):
    with open(p, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def _decode_occ(path, W):
    """(k, table mask, sorted list of (key words..., count)) of a PREFIX_kmer_occ.bin (doubleHash.h:266-278)."""
    data = open(path, "rb").read()
    k, mask = struct.unpack_from("<QQ", data, 0)
    rec = 8 + (24 if W > 1 else 0) + 8 * W + 2
    n = (len(data) - 16) // rec
    out = []
    for i in range(n):
        o = 16 + i * rec + 8 + (24 if W > 1 else 0)
        words = struct.unpack_from("<%dQ" % W, data, o)
        (c,) = struct.unpack_from("<H", data, o + 8 * W)
        out.append(words + (c,))
    return k, mask, sorted(out)


def _run_case(manifest, name, run, tmp_path):
    from platanus_b_amd import synth
    case = manifest[name]
    info = case["runs"][run]
    synth.write_pair_files(str(tmp_path / "in"), case["genome"], case["cov"], case["seed"], err=case["err"], n_rate=case["n_rate"])
    p = subprocess.run([BIN, "assemble", *info["args"], "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return info, p.stderr


@pytest.mark.parametrize("name,run", [("g5k", "multik"), ("g5k", "k21_s0"), ("c1", "s0"), ("c1", "multik"), ("k45", "multik_k45"), ("c2", "multik_n2")])
def test_cli_contigs_and_histogram_identical_to_reference(name, run, manifest, tmp_path):
    assert os.path.exists(BIN), "platanus_b_amd/bin/platanus_b missing: run __graft_entry__.build()"
    info, _ = _run_case(manifest, name, run, tmp_path)
    for fn, meta in info["files"].items():
        got = tmp_path / fn
        assert got.exists(), fn
        assert _md5(got) == meta["md5"], fn
        gold = os.path.join(GOLDEN, f"{name}.{run}.{fn}")
        if os.path.exists(gold):
            assert filecmp.cmp(got, gold, shallow=False), fn


def test_cli_kmer_occ_only(manifest, tmp_path):
    """-kmer_occ_only: TSV byte-identical; PREFIX_kmer_occ.bin holds the same k, table size and (key,count) set as the
    reference's file (slot layout: see DESIGN.md CT8 -- byte-exact layout is an oracle-only result so far)."""
    info, _ = _run_case(manifest, "g5k", "occ_n2", tmp_path)
    assert filecmp.cmp(tmp_path / "out_32merFrq.tsv", os.path.join(GOLDEN, "g5k.occ_n2.out_32merFrq.tsv"), shallow=False)
    got = _decode_occ(tmp_path / "out_kmer_occ.bin", 1)
    ref = _decode_occ(os.path.join(GOLDEN, "g5k.occ_n2.out_kmer_occ.bin"), 1)
    assert got == ref
    assert os.path.getsize(tmp_path / "out_kmer_occ.bin") == info["files"]["out_kmer_occ.bin"]["bytes"]


def test_cli_usage_and_errors(tmp_path):
    p = subprocess.run([BIN, "assemble", "-h"], capture_output=True, text=True)
    assert p.returncode == 1 and "Usage platanus_b assemble" in p.stderr          # main.cpp:114-117
    p = subprocess.run([BIN, "assemble", "-bogus", "1"], capture_output=True, text=True)
    assert p.returncode == 1
