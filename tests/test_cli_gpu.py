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
    synth.write_pair_files(str(tmp_path / "in"), case["genome"], case["cov"], case["seed"], err=case["err"], n_rate=case["n_rate"], het=case.get("het", 0.0))
    p = subprocess.run([BIN, "assemble", *info["args"], "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    return info, p.stderr


@pytest.mark.parametrize("name,run", [("g5k", "multik"), ("g5k", "k21_s0"), ("c1", "s0"), ("c1", "multik"), ("k45", "multik_k45"), ("c2", "multik_n2"), ("c1", "u1"), ("het", "u1"), ("het", "u0")])
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


@pytest.mark.parametrize("world,name,run", [(2, "c2", "multik_n2"), (4, "c2", "multik_n2"), (3, "c1", "multik"), (2, "k45", "multik_k45"), (2, "het", "u1")])
def test_cli_n_ranks_write_the_reference_files(world, name, run, manifest, tmp_path):
    """`assemble -gpus N`: N processes, reads range-partitioned over them, owner exchange + merge of the occurrence tables, gather
    of the retained shards before every graph stage, read filter and missing-k-mer count sharded by reads.
    Here the ranks share the one GPU of the box and exchange through shared host memory (-share_gpu; RCCL refuses two ranks on a
    device); the files must be the real reference's, byte for byte: 1-, 2- and 3-word keys, N bases, -u 1."""
    case = manifest[name]
    info = case["runs"][run]
    from platanus_b_amd import synth
    synth.write_pair_files(str(tmp_path / "in"), case["genome"], case["cov"], case["seed"], err=case["err"], n_rate=case["n_rate"], het=case.get("het", 0.0))
    args = [a for a in info["args"]]
    p = subprocess.run([BIN, "assemble", *args, "-gpus", str(world), "-share_gpu", "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    for fn, meta in info["files"].items():
        assert (tmp_path / fn).exists(), fn
        assert _md5(tmp_path / fn) == meta["md5"], fn
        gold = os.path.join(GOLDEN, f"{name}.{run}.{fn}")
        if os.path.exists(gold):
            assert filecmp.cmp(tmp_path / fn, gold, shallow=False), fn


@pytest.mark.parametrize("ranks", [1, 2])
def test_cli_k0_above_32_through_the_two_word_super_kmer_pass(ranks, manifest, tmp_path):
    """`assemble -k 45`: the first counting pass through kernels_superk2.inc (forced with -gpu_opt count_path=2: the golden input is below
    the size at which it is the default), one rank and two (count per rank + owner merge of two-word keys): the reference's files."""
    case = manifest["k45"]
    info = case["runs"]["multik_k45"]
    from platanus_b_amd import synth
    synth.write_pair_files(str(tmp_path / "in"), case["genome"], case["cov"], case["seed"], err=case["err"], n_rate=case["n_rate"], het=case.get("het", 0.0))
    extra = ["-gpus", str(ranks), "-share_gpu"] if ranks > 1 else []
    p = subprocess.run([BIN, "assemble", *info["args"], *extra, "-gpu_opt", "count_path=2", "-gpu_opt", "verbose=1", "-o", "out", "-f", "in_1.fq", "in_2.fq"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "super-k-mer pass (two-word keys): k 45" in p.stderr and "overflow 0" in p.stderr, p.stderr[-3000:]
    for fn, meta in info["files"].items():
        assert (tmp_path / fn).exists(), fn
        assert _md5(tmp_path / fn) == meta["md5"], fn


@pytest.mark.parametrize("world,name,run", [(2, "c2", "multik_n2"), (4, "c1", "multik"), (2, "het", "u1")])
def test_cli_n_ranks_exchanging_run_records(world, name, run, manifest, tmp_path):
    """The same with the k0 counting done the other way round (pb_dist_count: run records grouped by the owner of their minimizer
    bucket, exchanged, counted on arrival) -- forced here with -gpu_opt count_path=2 because these inputs are below the size at which
    it is the default; the stage list (PB_TIMING) says which design the ranks agreed on."""
    case = manifest[name]
    info = case["runs"][run]
    from platanus_b_amd import synth
    synth.write_pair_files(str(tmp_path / "in"), case["genome"], case["cov"], case["seed"], err=case["err"], n_rate=case["n_rate"], het=case.get("het", 0.0))
    p = subprocess.run([BIN, "assemble", *info["args"], "-gpus", str(world), "-share_gpu", "-gpu_opt", "count_path=2", "-o", "out", "-f", "in_1.fq", "in_2.fq"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=900, env=dict(os.environ, PB_TIMING="1"))
    assert p.returncode == 0, p.stderr[-3000:]
    assert "run records exchanged before the counting" in p.stderr, p.stderr[-3000:]
    for fn, meta in info["files"].items():
        assert (tmp_path / fn).exists(), fn
        assert _md5(tmp_path / fn) == meta["md5"], fn


def test_cli_gpu_opt_is_checked(tmp_path):
    """-gpu_opt without NAME=INT is a usage error; an option the library does not know ends the run with its message, not a crash."""
    from platanus_b_amd import synth
    synth.write_pair_files(str(tmp_path / "in"), 5000, 20, 3)
    p = subprocess.run([BIN, "assemble", "-gpu_opt", "count_path", "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 1 and "Usage" in p.stderr
    p = subprocess.run([BIN, "assemble", "-gpu_opt", "no_such_option=1", "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode not in (0, 1) and p.returncode > 0 and "unknown option" in p.stderr, p.stderr[-500:]


def test_cli_n_ranks_refuses_kmer_occ_only(tmp_path):
    from platanus_b_amd import synth
    synth.write_pair_files(str(tmp_path / "in"), 5000, 20, 3)
    p = subprocess.run([BIN, "assemble", "-kmer_occ_only", "-gpus", "2", "-share_gpu", "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode == 2 and "one GPU" in p.stderr and sum(ln.startswith("Error") for ln in p.stderr.splitlines()) == 1


def test_cli_n_ranks_error_every_rank_hits_is_said_once_with_the_reference_status(tmp_path):
    """A missing input file is an error of every rank: ONE message, and the exit status of the one-rank run (round 2: the ranks killed
    each other and the status depended on timing)."""
    from platanus_b_amd import synth
    synth.write_pair_files(str(tmp_path / "in"), 5000, 20, 3)
    one = subprocess.run([BIN, "assemble", "-o", "out", "-f", "in_1.fq", "nowhere.fq"], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert one.returncode not in (0, -9)
    for _ in range(3):
        p = subprocess.run([BIN, "assemble", "-gpus", "3", "-share_gpu", "-o", "out", "-f", "in_1.fq", "nowhere.fq"], cwd=tmp_path, capture_output=True, text=True,
                           timeout=300)
        assert p.returncode == one.returncode, (p.returncode, p.stderr[-600:])
        errs = [ln for ln in p.stderr.splitlines() if ln.startswith("Error")]
        assert errs == [ln for ln in one.stderr.splitlines() if ln.startswith("Error")], p.stderr[-600:]


def test_cli_kmer_occ_only_byte_identical(manifest, tmp_path):
    """-kmer_occ_only (BASELINE configs[1]'s mode): TSV and PREFIX_kmer_occ.bin byte-identical to the reference's files
    (k = 32, -t 1 layout; md5 for the 1 MB file of the 50 kbp case)."""
    info, _ = _run_case(manifest, "g5k", "occ_n2", tmp_path)
    assert filecmp.cmp(tmp_path / "out_32merFrq.tsv", os.path.join(GOLDEN, "g5k.occ_n2.out_32merFrq.tsv"), shallow=False)
    assert filecmp.cmp(tmp_path / "out_kmer_occ.bin", os.path.join(GOLDEN, "g5k.occ_n2.out_kmer_occ.bin"), shallow=False)
    sub = tmp_path / "c1"
    sub.mkdir()
    info, _ = _run_case(manifest, "c1", "occ", sub)
    for fn, meta in info["files"].items():
        assert _md5(sub / fn) == meta["md5"], fn


def test_cli_usage_and_errors(tmp_path):
    p = subprocess.run([BIN, "assemble", "-h"], capture_output=True, text=True)
    assert p.returncode == 1 and "Usage platanus_b assemble" in p.stderr          # main.cpp:114-117
    p = subprocess.run([BIN, "assemble", "-bogus", "1"], capture_output=True, text=True)
    assert p.returncode == 1


def test_cli_degenerate_inputs_and_exit_status():
    """tools/edge_cli.py: inputs without a countable k-mer, empty / unreadable / mis-formatted files, gzip data under a plain name,
    homopolymers and palindromes: exit status (the reference's error ids, main.cpp:118-127) and files equal to the oracle CLI's and,
    where oracle/_ref was built, the real reference's."""
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "edge_cli.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "0 mismatches" in p.stdout


def _write_fasta(path, codes, width=60, lower_every=7):
    with open(path, "w") as f:
        for i, r in enumerate(codes):
            s = "".join("ACGTN"[c] for c in r)
            if i % lower_every == 3:          # not record 0: the second line of a file may only hold "ACGTN" (baseCommand.cpp:39-44)
                s = s.lower()
            f.write(f">read{i}\n")
            for j in range(0, len(s), width):
                f.write(s[j:j + width] + "\n")


def test_cli_fasta_multiline_lowercase_and_gzip_match_oracle_cli(tmp_path):
    """Reader rules of assemble.cpp:790-986: multi-line FASTA records are joined, lower case is accepted after the format check
    (Char2Bin, common.h:256), gzip input goes through `gzip -cd`.  Compared with the oracle's CLI on the same files."""
    from platanus_b_amd import synth
    oracle_cli = os.path.join(ROOT, "oracle", "pb_oracle")
    assert os.path.exists(oracle_cli)
    r1, r2 = synth.make_reads(8000, 30, seed=21, n_rate=0.002)
    _write_fasta(tmp_path / "a.fa", r1)
    _write_fasta(tmp_path / "b.fa", r2)
    for tag, exe in (("gpu", BIN), ("cpu", oracle_cli)):
        p = subprocess.run([exe, "assemble", "-k", "25", "-o", tag, "-f", "a.fa", "b.fa"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
    assert filecmp.cmp(tmp_path / "gpu_contig.fa", tmp_path / "cpu_contig.fa", shallow=False)
    assert filecmp.cmp(tmp_path / "gpu_25merFrq.tsv", tmp_path / "cpu_25merFrq.tsv", shallow=False)
    # the same reads as gzip-compressed FASTQ
    synth.write_fastq(str(tmp_path / "c_1.fq"), r1, 1)
    synth.write_fastq(str(tmp_path / "c_2.fq"), r2, 2)
    subprocess.run(["gzip", "-k", "c_1.fq", "c_2.fq"], cwd=tmp_path, check=True)
    p = subprocess.run([BIN, "assemble", "-k", "25", "-o", "gz", "-f", "c_1.fq.gz", "c_2.fq.gz"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert filecmp.cmp(tmp_path / "gz_contig.fa", tmp_path / "cpu_contig.fa", shallow=False)


def test_cli_config3_full_multik_5mbp_md5(manifest, tmp_path):
    """BASELINE configs[2]: 5 Mbp x 100x PE150, full multi-k assemble (k = 32 ... 109, 1- to 4-word keys).  The real
    reference needed 1098 s (-t 8) for this input; its PREFIX_contig.fa / PREFIX_32merFrq.tsv are pinned by md5."""
    import time
    info, err = None, None
    t0 = time.time()
    info, err = _run_case(manifest, "cfg3", "multik", tmp_path)
    for fn, meta in info["files"].items():
        assert os.path.getsize(tmp_path / fn) == meta["bytes"], fn
        assert _md5(tmp_path / fn) == meta["md5"], fn
    print(f"config 3 incl. FASTQ generation: {time.time() - t0:.1f} s")
    # BASELINE configs[1] on the same files: `-kmer_occ_only` at the -t 1 schedule.  PREFIX_kmer_occ.bin (5,107,483 records of the
    # 60.8 M distinct 32-mers, slot layout of the reference's three sequential insertions) is pinned by the md5 of the real
    # reference's file (tests/golden/manifest.json cfg3.runs.occ_t1: 234 s of reference run).
    occ = manifest["cfg3"]["runs"]["occ_t1"]
    p = subprocess.run([BIN, "assemble", *occ["args"], "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    for fn, meta in occ["files"].items():
        assert os.path.getsize(tmp_path / fn) == meta["bytes"], fn
        assert _md5(tmp_path / fn) == meta["md5"], fn


def test_cli_piecewise_parser_equals_sequential_line_rules(manifest, tmp_path):
    """The host reader cuts a file at lines that start with the record mark and parses the pieces in parallel; by the
    reference's line rules (assemble.cpp:790-986) every such line is a restart point, so the result must not depend on the
    cuts.  Inputs that stress the rules: quality lines that start with '@' or '+', sequences split over several lines,
    empty lines, a header without sequence, no newline at the end of the file; FASTA with multi-line records.  The checker is
    the real reference binary (oracle/_ref, when built) or the oracle's CLI with its own sequential reader; pieces of 4 KiB
    force ~50 cuts per file."""
    from platanus_b_amd import synth
    oracle_bin = os.path.join(ROOT, "oracle", "_ref", "platanus_b")          # the real reference when it was built
    extra = ["-t", "1"]
    if not os.path.exists(oracle_bin):
        oracle_bin, extra = os.path.join(ROOT, "oracle", "pb_oracle"), []
    assert os.path.exists(oracle_bin)
    r1, r2 = synth.make_reads(20_000, 30, seed=5, n_rate=0.001)
    code2ascii = np.frombuffer(b"ACGTN", dtype=np.uint8)
    rng = np.random.default_rng(11)

    def fastq(path, reads):
        with open(path, "wb") as f:
            for i, r in enumerate(reads):
                seq = code2ascii[r].tobytes()
                f.write(b"@r%d\n" % i)
                if i % 7 == 3:                                   # sequence over three lines, with an empty line between
                    f.write(seq[:50] + b"\n\n" + seq[50:90] + b"\n" + seq[90:] + b"\n")
                else:
                    f.write(seq + b"\n")
                f.write(b"+\n")
                q = bytearray(b"I" * len(seq))
                if i % 5 == 1:
                    q[0] = ord("@")                              # quality line that looks like a header
                if i % 11 == 2:
                    q[0] = ord("+")
                f.write(bytes(q) + b"\n")
                if i % 13 == 4:
                    f.write(b"@orphan header without sequence\n+\n\n")
            f.write(b"@last\n" + code2ascii[reads[0]].tobytes())   # no newline at the end of the file

    def fasta(path, reads):
        with open(path, "wb") as f:
            for i, r in enumerate(reads):
                seq = code2ascii[r].tobytes()
                cut = int(rng.integers(10, 140))
                f.write(b">s%d\n" % i + seq[:cut] + b"\n" + seq[cut:] + b"\n")

    fastq(tmp_path / "a.fq", r1)
    fasta(tmp_path / "b.fa", r2)
    ref = tmp_path / "ref"
    ref.mkdir()
    p = subprocess.run([oracle_bin, "assemble", "-k", "32", *extra, "-o", "out", "-f", "../a.fq", "../b.fa"], cwd=ref, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    for piece in ("4096", "1000000000"):
        got = tmp_path / ("gpu" + piece)
        got.mkdir()
        env = dict(os.environ, PB_PARSE_PIECE_BYTES=piece)
        p = subprocess.run([BIN, "assemble", "-k", "32", "-o", "out", "-f", "../a.fq", "../b.fa"], cwd=got, capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        for fn in ("out_contig.fa", "out_32merFrq.tsv"):
            assert filecmp.cmp(got / fn, ref / fn, shallow=False), (piece, fn)


def test_cli_random_differential_against_oracle_cli():
    """tools/fuzz_cli.py: random small genomes with copied, inverted and tandem repeats (junctions, hairpins and cycles in the
    graph), random k / step / -u / -n / -c / -d / -K / -a / -e, one to three ranks: every output file of the GPU CLI must be
    byte-identical to the checker's.  The checker is the REAL reference binary (oracle/_ref/platanus_b travels to the GPU box);
    only where it is absent the oracle CLI, which shares the node-level host headers with the product."""
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_cli.py"), "12", "21"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "12 cases, 0 mismatches" in p.stdout and "SIGNAL" not in p.stdout          # (a run that dies of a signal is a failed case)
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "platanus_b")):
        assert "checker: reference" in p.stdout


def test_cli_kmer_divide_and_merge_random_options_against_oracle_cli():
    """tools/fuzz_cli2.py: contigs and k-mer occurrence table of random heterozygous read sets through `kmer_divide` (default / -mask /
    -r / -R) and `merge` (random -k / -l / -u): every output file of the GPU CLI byte-identical to the real reference binary's
    (the oracle CLI's where the binary is absent)."""
    import sys
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_cli2.py"), "4", "11"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert "4 cases, 0 mismatches" in p.stdout and "SIGNAL" not in p.stdout
