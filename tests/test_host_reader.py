"""Host logic without a GPU: the product's piece-wise parallel FASTA/FASTQ reader (platanus_b_amd/host/read_files.hpp) must give
the reads of the sequential line rules (assemble.cpp:790-986, restated in oracle/src/fastx.hpp) whatever the cut points are.
The same inputs go through the real reference binary and the GPU CLI in tests/test_cli_gpu.py."""
import numpy as np
import pytest


def _tricky_fastq(path, rng, n=3000):
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for i in range(n):
            L = int(rng.integers(1, 200))
            seq = bases[rng.choice(5, L, p=[0.245, 0.245, 0.245, 0.245, 0.02])].tobytes()
            f.write(b"@r%d some description\n" % i)
            if i % 7 == 3 and L > 20:
                f.write(seq[:10] + b"\n\n" + seq[10:15] + b"\n" + seq[15:] + b"\n")     # multi-line sequence, empty line inside
            else:
                f.write(seq + b"\n")
            f.write(b"+\n")
            q = bytearray(b"I" * L)
            if i % 5 == 1:
                q[0] = ord("@")                                                         # quality line that starts like a header
            if i % 11 == 2:
                q[0] = ord("+")
            f.write(bytes(q) + b"\n")
            if i % 13 == 4:
                f.write(b"@header without a sequence\n+\n\n")
        f.write(b"@last\nACGTACGT")                                                     # no newline at the end


def _multiline_fasta(path, rng, n=3000):
    bases = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as f:
        for i in range(n):
            L = int(rng.integers(1, 400))
            seq = bases[rng.choice(5, L, p=[0.245, 0.245, 0.245, 0.245, 0.02])].tobytes()
            f.write(b">s%d\n" % i)
            for a in range(0, L, 60):
                f.write(seq[a:a + 60] + b"\n")
            if i % 17 == 5:
                f.write(b">empty record\n")


@pytest.mark.parametrize("piece", ["64", "777", "4096", "100000000"])
@pytest.mark.parametrize("kind", ["fastq", "fasta"])
def test_piecewise_reader_equals_sequential(kind, piece, tmp_path, oracle_built, monkeypatch):
    rng = np.random.default_rng(42)
    path = str(tmp_path / ("in." + kind))
    (_tricky_fastq if kind == "fastq" else _multiline_fasta)(path, rng)
    monkeypatch.setenv("PB_PARSE_PIECE_BYTES", piece)
    rc, n_reads, n_pieces = oracle_built.parse_file_check(path, 8)
    assert rc == 0
    assert n_reads > 2900
    if int(piece) < 100000:
        assert n_pieces > 1                    # really cut


@pytest.mark.parametrize("parts", [2, 3, 8, 64])
@pytest.mark.parametrize("kind", ["fastq", "fasta"])
def test_rank_byte_ranges_concatenate_to_the_sequential_result(kind, parts, tmp_path, oracle_built, monkeypatch):
    """`assemble -gpus N`: rank r parses byte range r of every file (cut at record marks, like the thread pieces).  The ranges of all
    ranks, in rank order, must hold exactly the sequential reader's reads -- including the trailing read without a newline, which
    belongs to the last non-empty range, and ranks whose range is empty (64 parts of a small file)."""
    rng = np.random.default_rng(7)
    path = str(tmp_path / ("in." + kind))
    (_tricky_fastq if kind == "fastq" else _multiline_fasta)(path, rng, n=400 if parts == 64 else 3000)
    monkeypatch.setenv("PB_PARSE_PIECE_BYTES", "4096")
    rc, n_reads, n_pieces = oracle_built.parse_file_check(path, 4, parts)
    assert rc == 0
    assert n_reads > (380 if parts == 64 else 2900) and n_pieces >= min(parts, 8)
