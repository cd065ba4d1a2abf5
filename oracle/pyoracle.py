"""TEST INFRASTRUCTURE ONLY -- ctypes binding of oracle/liboracle.so (the CPU restatement).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only; the product
package platanus_b_amd never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_BIN = os.path.join(_HERE, "_ref", "platanus_b")
_lib = None


def build():
    subprocess.run(["make", "-C", _HERE, "-j4", "oracle"], check=True, stdout=subprocess.DEVNULL)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        l = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        l.pbo_create.restype = vp
        l.pbo_destroy.argtypes = [vp]
        l.pbo_count.argtypes = [vp, vp, vp, C.c_uint64, C.c_uint, C.c_uint64]
        l.pbo_count.restype = C.c_uint64
        l.pbo_occ_hist.argtypes = [vp, vp, C.POINTER(C.c_uint64)]
        l.pbo_length_hist.argtypes = [vp, vp, C.c_uint64]
        l.pbo_left_local_minimal_value.argtypes = [vp, C.c_uint64]
        l.pbo_left_local_minimal_value.restype = C.c_uint64
        l.pbo_num_distinct.argtypes = [vp]
        l.pbo_num_distinct.restype = C.c_uint64
        l.pbo_fetch_file_order.argtypes = [vp, vp, vp]
        l.pbo_retain.argtypes = [vp, C.c_uint64]
        l.pbo_retain.restype = C.c_uint64
        l.pbo_fetch_retained.argtypes = [vp, vp, vp]
        l.pbo_write_table_binary.argtypes = [vp, C.c_uint64, C.c_char_p]
        l.pbo_write_occ_distribution.argtypes = [vp, C.c_char_p]
        l.pbo_assemble.argtypes = [vp, vp, C.c_uint64, C.c_char_p, C.c_uint, C.c_uint, C.c_uint, C.c_uint64, C.c_int, C.c_int, C.c_char_p]
        l.pbo_extend_kmer.argtypes = [C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_double, C.c_uint, C.c_uint64,
                                      C.c_uint64, vp, vp, C.c_uint]
        l.pbo_extend_kmer.restype = C.c_uint64
        _lib = l
    return _lib


class OracleCounter:
    """Counter<KMER> of the reference at -t 1, restated on the CPU (oracle/src/counter.hpp)."""

    def __init__(self):
        self.l = lib()
        self.h = C.c_void_p(self.l.pbo_create())
        self.k = 0

    def __del__(self):
        try:
            self.l.pbo_destroy(self.h)
        except Exception:
            pass

    def count(self, codes: np.ndarray, offsets: np.ndarray, k: int, memory_bytes: int = 10 ** 9) -> int:
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.k = k
        r = self.l.pbo_count(self.h, codes.ctypes.data, offsets.ctypes.data, offsets.size - 1, k, memory_bytes)
        if r == 0:
            raise RuntimeError("oracle count failed")
        return r

    def occ_hist(self):
        h = np.zeros(65535, dtype=np.uint64)
        mx = C.c_uint64()
        self.l.pbo_occ_hist(self.h, h.ctypes.data, C.byref(mx))
        return h, mx.value

    def length_hist(self, n: int):
        h = np.zeros(n, dtype=np.uint64)
        self.l.pbo_length_hist(self.h, h.ctypes.data, n)
        return h

    def left_local_minimal_value(self, window: int = 1) -> int:
        return self.l.pbo_left_local_minimal_value(self.h, window)

    def file_order(self):
        n = self.l.pbo_num_distinct(self.h)
        W = (self.k + 31) // 32
        keys = np.zeros((n, W), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint16)
        self.l.pbo_fetch_file_order(self.h, keys.ctypes.data, counts.ctypes.data)
        return keys, counts

    def retain(self, min_occ: int):
        n = self.l.pbo_retain(self.h, min_occ)
        W = (self.k + 31) // 32
        keys = np.zeros((n, W), dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint16)
        self.l.pbo_fetch_retained(self.h, keys.ctypes.data, counts.ctypes.data)
        return keys, counts

    def write_table_binary(self, min_occ: int, path: str):
        if self.l.pbo_write_table_binary(self.h, min_occ, path.encode()) != 0:
            raise RuntimeError("oracle write_table_binary failed")

    def write_occ_distribution(self, path: str):
        if self.l.pbo_write_occ_distribution(self.h, path.encode()) != 0:
            raise RuntimeError("oracle write_occ_distribution failed")


def assemble(codes: np.ndarray, offsets: np.ndarray, out_prefix: str, k0: int = 32, step: int = 10, n_cutoff: int = 0,
             memory_gb: int = 1, kmer_occ_only: bool = False, verbose: bool = False, bubble: str = "") -> int:
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    return lib().pbo_assemble(codes.ctypes.data, offsets.ctypes.data, offsets.size - 1, out_prefix.encode(), k0, step,
                              n_cutoff, memory_gb, int(kmer_occ_only), int(verbose), bubble.encode())


def extend_kmer(min_log_p_join, avg_cov, avg_len, min_cov, max_k_ratio, k0, cutoff0, step):
    ks = np.zeros(512, dtype=np.uint32)
    cs = np.zeros(512, dtype=np.uint64)
    n = lib().pbo_extend_kmer(min_log_p_join, avg_cov, avg_len, min_cov, max_k_ratio, k0, cutoff0, step,
                              ks.ctypes.data, cs.ctypes.data, 512)
    return [int(x) for x in ks[:n]], [int(x) for x in cs[:n]]


def have_reference() -> bool:
    return os.path.exists(REF_BIN) and os.access(REF_BIN, os.X_OK)


def replay_check(codes: np.ndarray, offsets: np.ndarray, k: int, min_occ: int) -> int:
    """0 if the product's adjacency-array DFS replay (platanus_b_amd/host/device_graph.hpp) builds exactly the node
    tables of the oracle's k-mer-level DFS, in container order; CPU only."""
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    l = lib()
    l.pbo_replay_check.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint, C.c_uint64, C.c_void_p, C.c_void_p]
    return l.pbo_replay_check(codes.ctypes.data, offsets.ctypes.data, offsets.size - 1, k, min_occ, None, None)


def parse_file_check(path: str, max_threads: int = 8, parts: int = 1):
    """Product reader (platanus_b_amd/host/read_files.hpp, pieces in parallel; parts > 1: the byte ranges of that many ranks, one after
    the other) vs the oracle's sequential reader on one file.  Returns (code, n_reads, n_pieces); code 0 = identical reads."""
    l = lib()
    l.pbo_parse_file_check_parts.argtypes = [C.c_char_p, C.c_uint, C.c_uint, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    nr, npc = C.c_uint64(), C.c_uint64()
    rc = l.pbo_parse_file_check_parts(path.encode(), max_threads, parts, C.byref(nr), C.byref(npc))
    return rc, nr.value, npc.value


def gap_graphs(k: int, min_cov: int, seq_first: np.ndarray, seq_off: np.ndarray, codes: np.ndarray):
    """gapclose.hpp: the local de Bruijn graphs of a batch of gaps (counter.h:1015-1056 + graph.h:2116-2345).  Gap g owns the sequences
    seq_first[g] .. seq_first[g + 1] - 1, sequence s = codes[seq_off[s] : seq_off[s + 1]] (0..3, 4 = N).  Returns a dict of arrays
    (nodes in the reference's insertion order: node_first, kind 1 = straight / 0 = junction, cov, out, length, word_off, words) and the
    per-gap table dumps (iteration order of a fresh Graph<W> after those insertions)."""
    L = lib()
    L.pbo_gap_graphs_run.argtypes = [C.c_uint, C.c_long, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]
    L.pbo_gap_graphs_fetch.argtypes = [C.c_void_p] * 7
    L.pbo_gap_graphs_dump.argtypes = [C.c_ulonglong]
    L.pbo_gap_graphs_dump.restype = C.c_char_p
    seq_first = np.ascontiguousarray(seq_first, dtype=np.uint64)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n_gaps = len(seq_first) - 1
    nn, nw = C.c_ulonglong(), C.c_ulonglong()
    rc = L.pbo_gap_graphs_run(k, min_cov, n_gaps, seq_first.ctypes.data, seq_off.ctypes.data, codes.ctypes.data, C.byref(nn), C.byref(nw))
    if rc != 0:
        raise RuntimeError(f"pbo_gap_graphs_run: {rc}")
    out = {"node_first": np.zeros(n_gaps + 1, dtype=np.uint64), "kind": np.zeros(nn.value, dtype=np.uint8), "cov": np.zeros(nn.value, dtype=np.uint16),
           "out": np.zeros(nn.value, dtype=np.uint8), "length": np.zeros(nn.value, dtype=np.uint32), "word_off": np.zeros(nn.value + 1, dtype=np.uint64),
           "words": np.zeros(nw.value, dtype=np.uint64)}
    L.pbo_gap_graphs_fetch(*(out[n].ctypes.data for n in ("node_first", "kind", "cov", "out", "length", "word_off", "words")))
    dumps = [L.pbo_gap_graphs_dump(g).decode() for g in range(n_gaps)]
    return out, dumps


def gap_graphs_reference(k: int, min_cov: int, seq_first: np.ndarray, seq_off: np.ndarray, codes: np.ndarray):
    """The same tables from the REFERENCE's own functions (oracle/_ref/gapclose_ref, built from ref_harness/gapclose_ref.cpp where
    /root/reference exists): per-gap dumps in the format of gap_graphs()'s second result."""
    import struct
    import subprocess
    import tempfile
    exe = os.path.join(_HERE, "_ref", "gapclose_ref")
    seq_first = np.ascontiguousarray(seq_first, dtype=np.uint64)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        f.write(struct.pack("<IqQ", k, min_cov, len(seq_first) - 1))
        f.write(seq_first.tobytes())
        f.write(struct.pack("<Q", len(seq_off) - 1))
        f.write(seq_off.tobytes())
        f.write(codes[:int(seq_off[-1])].tobytes())
        f.flush()
        p = subprocess.run([exe, f.name], capture_output=True, text=True, timeout=600)
    if p.returncode != 0:
        raise RuntimeError(f"gapclose_ref: {p.returncode} {p.stderr[-500:]}")
    return split_gap_dump(p.stdout)


def split_gap_dump(text: str):
    dumps, cur = [], None
    for ln in text.splitlines(keepends=True):
        if ln.startswith("G "):
            if cur is not None:
                dumps.append(cur)
            cur = ""
        elif cur is not None:
            cur += ln
    if cur is not None:
        dumps.append(cur)
    return dumps


def gap_assemble(k_low: int, k_high: int, min_cov: int, seq_first: np.ndarray, seq_off: np.ndarray, codes: np.ndarray, flanks: bytes = None):
    """gapclose.hpp gap_assemble(): the graph part of GapCloseDBG<M, N>::gapAssemble (gapCloseDBG.h:416-436) gap by gap; per gap the dump
    "L\\n<low graph's live nodes>H\\n<high graph's>" in the tables' iteration order."""
    L = lib()
    L.pbo_gap_assemble_run.argtypes = [C.c_uint, C.c_uint, C.c_long, C.c_ulonglong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p]
    L.pbo_gap_assemble_dump.argtypes = [C.c_ulonglong]
    L.pbo_gap_assemble_dump.restype = C.c_char_p
    seq_first = np.ascontiguousarray(seq_first, dtype=np.uint64)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    n_gaps = len(seq_first) - 1
    rc = L.pbo_gap_assemble_run(k_low, k_high, min_cov, n_gaps, seq_first.ctypes.data, seq_off.ctypes.data, codes.ctypes.data, flanks)
    if rc != 0:
        raise RuntimeError(f"pbo_gap_assemble_run: {rc}")
    return [L.pbo_gap_assemble_dump(g).decode() for g in range(n_gaps)]


def gap_assemble_reference(k_low: int, k_high: int, min_cov: int, seq_first: np.ndarray, seq_off: np.ndarray, codes: np.ndarray, flanks: bytes = None):
    """The same from the REFERENCE's own GapCloseDBG<k_low, k_high>::gapAssemble (oracle/_ref/gapclose_ref INPUT K_HIGH; the pairs the harness
    instantiates: (20, 80) -- the command's own --, (25, 60), (28, 45), (32, 100), (40, 90))."""
    import struct
    import tempfile
    exe = os.path.join(_HERE, "_ref", "gapclose_ref")
    seq_first = np.ascontiguousarray(seq_first, dtype=np.uint64)
    seq_off = np.ascontiguousarray(seq_off, dtype=np.uint64)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        f.write(struct.pack("<IqQ", k_low, min_cov, len(seq_first) - 1))
        f.write(seq_first.tobytes())
        f.write(struct.pack("<Q", len(seq_off) - 1))
        f.write(seq_off.tobytes())
        f.write(codes[:int(seq_off[-1])].tobytes())
        f.flush()
        args = [exe, f.name, str(k_high)]
        ff = None
        if flanks is not None:
            ff = tempfile.NamedTemporaryFile(suffix=".flanks")
            ff.write(flanks)
            ff.flush()
            args.append(ff.name)
        p = subprocess.run(args, capture_output=True, text=True, timeout=900)
        if ff is not None:
            ff.close()
    if p.returncode != 0:
        raise RuntimeError(f"gapclose_ref: {p.returncode} {p.stderr[-500:]}")
    return split_gap_dump(p.stdout)
