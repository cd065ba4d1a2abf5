"""`kmer_divide`, the consumer of PREFIX_kmer_occ.bin (kmer_divide.cpp:26-388).  Inputs are rebuilt
from the seeded generator (contigs by `assemble`, table by `assemble -kmer_occ_only -n 1`, both already pinned
byte-for-byte), outputs are compared with the files the real reference's `kmer_divide` wrote (tests/golden).
CPU test: the oracle CLI (CPU look-ups).  GPU test: platanus_b_amd/bin/platanus_b (look-ups on the device)."""
import filecmp
import hashlib
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT

ORACLE_CLI = os.path.join(ROOT, "oracle", "pb_oracle")
GPU_CLI = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")


def _md5(p):
    with open(p, "rb") as f:
        return hashlib.md5(f.read()).hexdigest()


def _run_all(exe, manifest, tmp_path):
    from platanus_b_amd import synth
    case = manifest["het"]
    synth.write_pair_files(str(tmp_path / "in"), case["genome"], case["cov"], case["seed"], err=case["err"], n_rate=case["n_rate"], het=case["het"])
    run = lambda *a: subprocess.run([exe, *a], cwd=tmp_path, capture_output=True, text=True, timeout=900)
    p = run("assemble", *case["runs"]["u0"]["args"], "-o", "a", "-f", "in_1.fq", "in_2.fq")
    assert p.returncode == 0, p.stderr[-2000:]
    p = run("assemble", *case["runs"]["occ_n1"]["args"], "-o", "t", "-f", "in_1.fq", "in_2.fq")
    assert p.returncode == 0, p.stderr[-2000:]
    assert _md5(tmp_path / "a_contig.fa") == case["runs"]["u0"]["files"]["out_contig.fa"]["md5"]
    assert _md5(tmp_path / "t_kmer_occ.bin") == case["runs"]["occ_n1"]["files"]["out_kmer_occ.bin"]["md5"]
    for name, info in case["kmer_divide"].items():
        p = run("kmer_divide", "-k", "t_kmer_occ.bin", "-f", "a_contig.fa", "-o", name, *info["args"])
        assert p.returncode == 0, p.stderr[-2000:]
        for fn, meta in info["files"].items():
            got = tmp_path / fn.replace("out_", name + "_", 1)
            assert got.exists(), got
            assert _md5(got) == meta["md5"], (name, fn)
            assert filecmp.cmp(got, os.path.join(GOLDEN, f"het.{name}.{fn}"), shallow=False)


def test_kmer_divide_oracle_cli_matches_reference(manifest, oracle_built, tmp_path):
    assert os.path.exists(ORACLE_CLI)
    _run_all(ORACLE_CLI, manifest, tmp_path)


@pytest.mark.gpu
def test_kmer_divide_gpu_cli_matches_reference(manifest, tmp_path):
    assert os.path.exists(GPU_CLI), "run __graft_entry__.build()"
    _run_all(GPU_CLI, manifest, tmp_path)
