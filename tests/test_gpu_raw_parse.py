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
