"""pb_reads_append_raw parses FASTA / FASTQ bytes on the device (newline positions, line classes, record segments and
read offsets by map / prefix-scan kernels).  Checked against a plain-Python restatement of the reference's line rules
(assemble.cpp:790-986 -- the same rules as host/read_files.hpp::parse_piece, which the CPU suite pins against the oracle's sequential
reader and the GPU CLI against the real reference binary) on files made to be awkward, whole and cut into pieces at record marks; the read
set that comes out must be the one pb_reads_append builds from the parsed reads: same read starts, same length histogram, same k-mers."""
import numpy as np
import pytest

from test_host_reader import _multiline_fasta, _tricky_fastq

pytestmark = pytest.mark.gpu

_TAB = {ord("A"): 0, ord("a"): 0, ord("C"): 1, ord("c"): 1, ord("G"): 2, ord("g"): 2, ord("T"): 3, ord("t"): 3}


def _parse(data: bytes, fastq: bool):
    """The reference's sequential reader on a whole file: list of reads (bytes)."""
    lines = data.split(b"\n")
    if data.endswith(b"\n"):
        lines.pop()                                   # nothing behind the final newline
    reads, cur, flag = [], bytearray(), True
    for ln in lines[1:]:                              # the format line has been consumed
        if fastq:
            if not ln:
                continue
            if ln[:1] != b"@":
                if flag and ln[:1] != b"+":
                    cur += ln
                else:
                    flag = False
            else:
                if cur:
                    reads.append(bytes(cur))
                    cur = bytearray()
                flag = True
        else:
            if not ln or ln[:1] != b">":
                cur += ln
            elif cur:
                reads.append(bytes(cur))
                cur = bytearray()
    reads.append(bytes(cur))                          # the trailing flush is unconditional
    return reads


def _cuts(data: bytes, mark: bytes, n_pieces: int):
    """Piece boundaries at lines that start with the record mark (as read_file cuts)."""
    cuts = [0]
    for i in range(1, n_pieces):
        pos = len(data) * i // n_pieces
        while True:
            nl = data.find(b"\n", pos)
            if nl < 0:
                pos = len(data)
                break
            pos = nl + 1
            if pos >= len(data) or data[pos:pos + 1] == mark:
                break
        if cuts[-1] < pos < len(data):
            cuts.append(pos)
    cuts.append(len(data))
    return cuts


def _host_set(g, reads):
    codes = np.frombuffer(b"".join(reads), dtype=np.uint8)
    off = np.concatenate([[0], np.cumsum([len(r) for r in reads])]).astype(np.uint64)
    g.reads_append(codes, off, ascii=True)


@pytest.mark.parametrize("n_pieces", [1, 2, 7, 40])
@pytest.mark.parametrize("kind", ["fastq", "fasta"])
def test_device_parse_equals_the_line_rules(kind, n_pieces, tmp_path):
    from platanus_b_amd.binding import GpuCounter
    rng = np.random.default_rng(11)
    path = str(tmp_path / ("in." + kind))
    (_tricky_fastq if kind == "fastq" else _multiline_fasta)(path, rng)
    data = open(path, "rb").read()
    fastq = kind == "fastq"
    reads = _parse(data, fastq)
    assert len(reads) > 2900
    a, b = GpuCounter(0), GpuCounter(0)
    _host_set(a, reads)
    cuts = _cuts(data, b"@" if fastq else b">", n_pieces)
    assert n_pieces == 1 or len(cuts) > 2
    got_reads = got_bases = 0
    for i in range(len(cuts) - 1):
        nr, nb = b.reads_append_raw(data[cuts[i]:cuts[i + 1]], fastq, first_piece=i == 0, last_piece=i + 2 == len(cuts))
        got_reads += nr
        got_bases += nb
    assert (got_reads, got_bases) == a.reads_info() == b.reads_info() == (len(reads), sum(len(r) for r in reads))
    assert np.array_equal(a.reads_starts(), b.reads_starts())
    assert np.array_equal(a.length_hist(500), b.length_hist(500))
    for k in (15, 33):
        ha, hb = a.make_kmer_read_distribution(k), b.make_kmer_read_distribution(k)
        assert np.array_equal(ha, hb) and a.n_instances == b.n_instances > 0
        a.retain(1)
        b.retain(1)
        ka, ca = a.fetch_retained()
        kb, cb = b.fetch_retained()
        assert np.array_equal(ka, kb) and np.array_equal(ca, cb)


def test_device_parse_edge_cases():
    from platanus_b_amd.binding import GpuCounter, PbError
    cases = [(b"@r\nACGT\n+\nIIII\n", True), (b"@r\nACGT", True), (b"@only a header\n", True), (b"@h\n+\n\n@h2\nAC\nGT\n+\nII\nII", True),
             (b">s\nACGT\nNNAC\n>empty\n>t\nAC", False), (b">s\n", False), (b"@a\nAC\n+\n@@\n@b\nGG\n+\n+I\n", True)]
    for data, fastq in cases:
        reads = _parse(data, fastq)
        a, b = GpuCounter(0), GpuCounter(0)
        _host_set(a, reads)
        assert b.reads_append_raw(data, fastq) == (len(reads), sum(len(r) for r in reads)), data
        assert a.reads_info() == b.reads_info() and np.array_equal(a.reads_starts(), b.reads_starts()), data
    g = GpuCounter(0)
    with pytest.raises(PbError):                       # a read of 500,000 bases: ReadError in the reference (common.cpp:28); nothing is appended
        g.reads_append_raw(b">s\n" + b"A" * 500_000 + b"\n", False)
    assert g.reads_info() == (0, 0)
    assert g.reads_append_raw(b">s\n" + b"A" * 499_999 + b"\n", False) == (1, 499_999)


@pytest.mark.parametrize("name,run,extra", [("c2", "multik_n2", []), ("c1", "multik", ["-gpus", "3", "-share_gpu"]), ("het", "u1", [])])
def test_cli_device_parse_writes_the_reference_files(name, run, extra, manifest, tmp_path):
    """`assemble -device_parse`: the read files' bytes go to the device as they are (this rank's byte range, cut at record marks) and
    the line rules run there; the output files must still be the real reference's (N bases, -n 2, 3 ranks, -u 1)."""
    import hashlib
    import os
    import subprocess
    from conftest import ROOT
    from platanus_b_amd import synth
    case = manifest[name]
    info = case["runs"][run]
    synth.write_pair_files(str(tmp_path / "in"), case["genome"], case["cov"], case["seed"], err=case["err"], n_rate=case["n_rate"], het=case.get("het", 0.0))
    cli = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")
    p = subprocess.run([cli, "assemble", *info["args"], *extra, "-device_parse", "-o", "out", "-f", "in_1.fq", "in_2.fq"], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    for fn, meta in info["files"].items():
        assert hashlib.md5(open(tmp_path / fn, "rb").read()).hexdigest() == meta["md5"], fn


def test_cli_device_parse_on_awkward_files(tmp_path):
    """The awkward FASTQ / FASTA files of the host-reader tests (multi-line sequences, quality lines that start with '@' or '+', headers
    without a sequence, no final newline, gzip) through the CLI: -device_parse and the host parser must write the same files."""
    import filecmp
    import os
    import subprocess
    from conftest import ROOT
    rng = np.random.default_rng(3)
    _tricky_fastq(str(tmp_path / "a.fq"), rng)
    _multiline_fasta(str(tmp_path / "b.fa"), rng)
    subprocess.run(["gzip", "-k", str(tmp_path / "a.fq")], check=True)
    cli = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")
    for files in (["a.fq", "b.fa"], ["a.fq.gz", "b.fa"]):
        outs = []
        for tag, extra in (("h", []), ("d", ["-device_parse"]), ("d2", ["-device_parse", "-gpus", "2", "-share_gpu"])):
            p = subprocess.run([cli, "assemble", "-k", "21", "-n", "1", *extra, "-o", tag, "-f", *files], cwd=tmp_path, capture_output=True, text=True, timeout=600)
            assert p.returncode == 0, p.stderr[-2000:]
            outs.append(tag)
        for tag in outs[1:]:
            for fn in ("_contig.fa", "_21merFrq.tsv"):
                assert filecmp.cmp(tmp_path / ("h" + fn), tmp_path / (tag + fn), shallow=False), (files, tag, fn)
