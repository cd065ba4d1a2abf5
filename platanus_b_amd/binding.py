"""ctypes binding of libpbgpu.so (include/pbgpu.h) -- the product path.

There is NO CPU fallback: if the in-tree HIP library is missing or no gfx950 device is visible the
constructors raise.  The class mirrors the call surface `Assemble` uses on the reference's
`Counter<KMER>` (counter.h:146-164) so tests read like the reference's call sites
(assemble.cpp:304-350).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpbgpu.so")

PB_OCC_BINS = 65535
PB_COUNT_MAX = 65534

_lib = None


class PbError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libpbgpu error {code}: {msg}")
        self.code = code


def load_library() -> C.CDLL:
    """Load the in-tree library (the .so next to this file or nothing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    # torch's ROCm wheel bundles its own libamdhip64.so.7 / libhsa-runtime64; whichever HIP runtime is loaded
    # first serves the whole process.  Loading libpbgpu first and torch afterwards leaves torch without a
    # device ("No HIP GPUs are available", measured), so when torch is installed it goes first and both share
    # one runtime -- torch tensors' device pointers are then valid in libpbgpu and vice versa.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    vp, u64p = C.c_void_p, C.POINTER(C.c_uint64)
    lib.pb_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.pb_destroy.argtypes = [vp]
    lib.pb_destroy.restype = None
    lib.pb_last_error.argtypes = [vp]
    lib.pb_last_error.restype = C.c_char_p
    lib.pb_set_option.argtypes = [vp, C.c_char_p, C.c_long]
    lib.pb_device_info.argtypes = [vp, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), u64p]
    lib.pb_reads_append.argtypes = [vp, vp, vp, C.c_uint64, C.c_int]
    lib.pb_reads_append_raw.argtypes = [vp, vp, C.c_uint64, C.c_int, C.c_int, C.c_int, u64p, u64p]
    lib.pb_reads_starts.argtypes = [vp, vp, C.c_uint64]
    lib.pb_reads_clear.argtypes = [vp]
    lib.pb_reads_info.argtypes = [vp, u64p, u64p]
    lib.pb_reads_length_hist.argtypes = [vp, vp, C.c_uint64]
    lib.pb_count.argtypes = [vp, C.c_uint, vp, u64p, u64p, u64p]
    lib.pb_retain.argtypes = [vp, C.c_uint, u64p]
    lib.pb_fetch_retained.argtypes = [vp, vp, vp]
    lib.pb_fetch_all.argtypes = [vp, vp, vp, C.c_uint64, u64p]
    lib.pb_last_kernel_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float)]
    lib.pb_export_begin.argtypes = [vp, C.c_int, u64p]
    lib.pb_export_copy.argtypes = [vp, vp, vp]
    lib.pb_export_split_begin.argtypes = [vp, C.c_int, u64p, u64p]
    lib.pb_export_split_copy.argtypes = [vp, vp, vp]
    lib.pb_merge_begin.argtypes = [vp, C.c_uint, C.c_uint64]
    lib.pb_merge_add.argtypes = [vp, vp, vp, C.c_uint64]
    lib.pb_merge_end.argtypes = [vp, vp, u64p, u64p]
    lib.pb_merge_records.argtypes = [vp, C.c_uint, vp, vp, C.c_uint64, vp, u64p, u64p]
    lib.pb_kmer_occ_layout.argtypes = [vp, C.c_uint64, C.c_uint, u64p, u64p]
    lib.pb_sk_bucket_bits.argtypes = [C.c_uint64]
    lib.pb_sk_bucket_bits.restype = C.c_uint
    lib.pb_sk_owner_records.argtypes = [vp, C.c_uint, C.c_uint, C.c_uint, u64p, u64p]
    lib.pb_sk_owner_records_copy.argtypes = [vp, vp]
    lib.pb_sk_count_records.argtypes = [vp, C.c_uint, vp, C.c_uint64, C.c_uint, vp, u64p, u64p]
    lib.pb_kmer_occ_fetch.argtypes = [vp, vp, vp, vp]
    lib.pb_graph_adjacency.argtypes = [vp, vp, vp]
    lib.pb_graph_chains.argtypes = [vp, u64p, u64p]
    lib.pb_graph_fetch_chains.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.pb_graph_nodes.argtypes = [vp, u64p]
    lib.pb_graph_fetch_nodes.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.pb_graph_fetch_chain_records.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.pb_reads_filter.argtypes = [vp, C.c_uint, vp, C.c_uint64, u64p]
    lib.pb_reads_restore.argtypes = [vp]
    lib.pb_contig_table_build.argtypes = [vp, C.c_uint, vp, vp, vp, C.c_uint64, u64p]
    lib.pb_count_considering_contigs.argtypes = [vp, C.c_uint, vp, u64p, u64p]
    lib.pb_occ_table_load.argtypes = [vp, C.c_uint, vp, vp, C.c_uint64]
    lib.pb_occ_lookup.argtypes = [vp, vp, vp, C.c_uint64, vp]
    lib.pb_nodes_clean_decide.argtypes = [vp, C.c_uint, C.c_uint64, vp, vp, vp, vp, vp, C.c_uint64, vp, vp, C.c_uint64, vp, vp, vp, C.c_double, C.c_uint64, C.c_uint64, vp]
    lib.pb_edit_distances.argtypes = [vp, vp, vp, C.c_uint64, vp]
    lib.pb_gap_graphs_build.argtypes = [vp, C.c_uint, C.c_long, C.c_uint64, vp, C.c_uint64, vp, vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pb_gap_graphs_fetch.argtypes = [vp] * 8
    _lib = lib
    return lib


def words_for_k(k: int) -> int:
    return (k + 31) // 32


class GpuCounter:
    """Device-side Counter<KMER> (counter.h:36-1056) for one GPU."""

    # pb_set_option calls applied to every new context (tests force a counting path on inputs the library would route elsewhere)
    default_options: dict = {}

    def __init__(self, device: int = 0):
        self.lib = load_library()
        self.h = C.c_void_p()
        rc = self.lib.pb_create(device, C.byref(self.h))
        if rc != 0:
            raise PbError(rc, self.lib.pb_last_error(None).decode())
        self.k = 0
        self.occ_hist: Optional[np.ndarray] = None
        self.max_occ = 0
        self.n_instances = 0
        self.n_distinct = 0
        self.n_retained = 0
        try:
            self.set_option("hist_zeroed", 1)          # every occ_hist array this class hands over is np.zeros
        except PbError:
            pass                                       # (a development build of the library from before the option)
        for name, value in self.default_options.items():
            self.set_option(name, value)

    def close(self):
        if self.h:
            self.lib.pb_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int):
        if rc != 0:
            raise PbError(rc, self.lib.pb_last_error(self.h).decode())

    def set_option(self, name: str, value: int):
        """pb_set_option: choice between equivalent implementations (e.g. "count_path" 0 auto / 1 direct / 2 super-k-mer pass)."""
        self._chk(self.lib.pb_set_option(self.h, name.encode(), int(value)))

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu = C.c_int()
        hbm = C.c_uint64()
        self._chk(self.lib.pb_device_info(self.h, name, 256, C.byref(ncu), C.byref(hbm)))
        return name.value.decode(), ncu.value, hbm.value

    # ---- reads -------------------------------------------------------------------------------------
    def reads_clear(self):
        self._chk(self.lib.pb_reads_clear(self.h))

    def reads_append(self, bases: np.ndarray, offsets: np.ndarray, ascii: bool = False):
        bases = np.ascontiguousarray(bases, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = offsets.size - 1
        self._chk(self.lib.pb_reads_append(self.h, bases.ctypes.data, offsets.ctypes.data, n, 1 if ascii else 0))

    def reads_append_raw(self, file_bytes, fastq: bool, first_piece: bool = True, last_piece: bool = True):
        """the bytes of (a piece of) a FASTA / FASTQ file, parsed on the device (assemble.cpp:790-986's line rules);
        returns (reads, bases) appended."""
        b = np.frombuffer(file_bytes, dtype=np.uint8) if not isinstance(file_bytes, np.ndarray) else np.ascontiguousarray(file_bytes, dtype=np.uint8)
        nr, nb = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.pb_reads_append_raw(self.h, b.ctypes.data if b.size else None, b.size, 1 if fastq else 0, 1 if first_piece else 0, 1 if last_piece else 0,
                                               C.byref(nr), C.byref(nb)))
        return nr.value, nb.value

    def reads_starts(self) -> np.ndarray:
        n, _ = self.reads_info()
        out = np.zeros(n + 1, dtype=np.uint64)
        self._chk(self.lib.pb_reads_starts(self.h, out.ctypes.data, n + 1))
        return out

    def reads_append_matrix(self, codes: np.ndarray):
        """codes: [n_reads, read_len] uint8 matrix (all reads the same length)."""
        n, L = codes.shape
        self.reads_append(codes.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L))

    def reads_info(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.pb_reads_info(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def length_hist(self, n: int) -> np.ndarray:
        h = np.zeros(n, dtype=np.uint64)
        self._chk(self.lib.pb_reads_length_hist(self.h, h.ctypes.data, n))
        return h

    # ---- counter.h:154 makeKmerReadDistributionMT ---------------------------------------------------
    def make_kmer_read_distribution(self, k: int):
        hist = np.zeros(PB_OCC_BINS, dtype=np.uint64)          # (untouched zero pages; with "hist_zeroed" the library writes the non-zero head only)
        mx, ni, nd = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._chk(self.lib.pb_count(self.h, k, hist.ctypes.data, C.byref(mx), C.byref(ni), C.byref(nd)))
        self.k, self.occ_hist, self.max_occ = k, hist, mx.value
        self.n_instances, self.n_distinct = ni.value, nd.value
        return hist

    # counter.h:153 / 244-267 (host arithmetic on the histogram, as in the reference)
    def get_left_local_minimal_value(self, window: int = 1) -> int:
        d, mx = self.occ_hist, self.max_occ
        if mx <= window:
            return 0
        n = mx - window + 2
        w = [0] * n
        for i in range(window):
            w[1] += int(d[1 + i])
        i = 2
        while i < n:
            w[i] = w[i - 1] - int(d[i - 1]) + int(d[i + window - 1])
            if w[i] >= w[i - 1]:
                break
            i += 1
        return (i - 1 + window // 2) if i <= mx else (1 + window // 2)

    # counter.h:149 / 220-238
    def calc_occurrence_distribution_average(self, start: int, end: int) -> float:
        d = self.occ_hist
        s = sum(i * int(d[i]) for i in range(start, end + 1))
        num = sum(int(d[i]) for i in range(start, end + 1))
        if num == 0:
            raise PbError(-1, "kmer distribution error")
        return float(s) / num

    # counter.h:163 / 999-1007
    def output_occurrence_distribution(self, path: str):
        with open(path, "w") as f:
            for i in range(1, self.max_occ + 1):
                f.write(f"{i}\t{int(self.occ_hist[i])}\n")

    # counter.h:160 + 155: sortedKeyFromKmerFile(minOccurrence) + loadKmer
    def retain(self, min_occ: int) -> int:
        n = C.c_uint64()
        self._chk(self.lib.pb_retain(self.h, min_occ, C.byref(n)))
        self.n_retained = n.value
        return n.value

    def fetch_retained(self):
        W = words_for_k(self.k)
        keys = np.zeros((self.n_retained, W), dtype=np.uint64)
        counts = np.zeros(self.n_retained, dtype=np.uint16)
        self._chk(self.lib.pb_fetch_retained(self.h, keys.ctypes.data, counts.ctypes.data))
        return keys, counts

    def fetch_all(self):
        W = words_for_k(self.k)
        cap = max(self.n_distinct, 1)
        keys = np.zeros((cap, W), dtype=np.uint64)
        counts = np.zeros(cap, dtype=np.uint32)
        n = C.c_uint64()
        self._chk(self.lib.pb_fetch_all(self.h, keys.ctypes.data, counts.ctypes.data, cap, C.byref(n)))
        return keys[:n.value], counts[:n.value]

    # counter.h:155 + 161: loadKmer + outputOccurrenceTableBinary with the reference's -t 1 slot layout (k <= 32)
    def output_occurrence_table_binary(self, path: str, min_occ: int, dh_size: int):
        ts, nr = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.pb_kmer_occ_layout(self.h, dh_size, min_occ, C.byref(ts), C.byref(nr)))
        n = nr.value
        slots = np.zeros(n, dtype=np.uint64)
        keys = np.zeros(n, dtype=np.uint64)
        counts = np.zeros(n, dtype=np.uint16)
        self._chk(self.lib.pb_kmer_occ_fetch(self.h, slots.ctypes.data, keys.ctypes.data, counts.ctypes.data))
        rec = np.zeros(n, dtype=np.dtype([("slot", "<u8"), ("key", "<u8"), ("count", "<u2")]))
        rec["slot"], rec["key"], rec["count"] = slots, keys, counts
        with open(path, "wb") as f:
            f.write(np.array([self.k, ts.value - 1], dtype="<u8").tobytes())
            f.write(rec.tobytes())
        return ts.value, n

    # ---- graph stage (graph.h:289-579, device part) on the retained set of the last retain() ----
    def graph_adjacency(self):
        """adj uint8 [n], nbr uint32 [n, 2] (include/pbgpu.h pb_graph_adjacency)."""
        n = self.n_retained
        adj = np.zeros(max(n, 1), dtype=np.uint8)
        nbr = np.zeros((max(n, 1), 2), dtype=np.uint32)
        self._chk(self.lib.pb_graph_adjacency(self.h, adj.ctypes.data, nbr.ctypes.data))
        return adj[:n], nbr[:n]

    def graph_chains(self):
        """dict of the per-k-mer and per-chain arrays of pb_graph_chains / pb_graph_fetch_chains."""
        nc, words = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.pb_graph_chains(self.h, C.byref(nc), C.byref(words)))
        n, c, w = self.n_retained, nc.value, words.value
        r = dict(chain_of=np.zeros(max(n, 1), dtype=np.uint32), orient=np.zeros(max(n, 1), dtype=np.uint8), len=np.zeros(max(c, 1), dtype=np.uint32),
                 sum=np.zeros(max(c, 1), dtype=np.uint64), end_a=np.zeros(max(c, 1), dtype=np.uint32), end_b=np.zeros(max(c, 1), dtype=np.uint32),
                 seq_woff=np.zeros(max(c, 1), dtype=np.uint64), seq=np.zeros(max(w, 1), dtype=np.uint64))
        self._chk(self.lib.pb_graph_fetch_chains(self.h, r["chain_of"].ctypes.data, r["orient"].ctypes.data, r["len"].ctypes.data, r["sum"].ctypes.data,
                                                 r["end_a"].ctypes.data, r["end_b"].ctypes.data, r["seq_woff"].ctypes.data, r["seq"].ctypes.data))
        for key in ("chain_of", "orient"):
            r[key] = r[key][:n]
        for key in ("len", "sum", "end_a", "end_b", "seq_woff"):
            r[key] = r[key][:c]
        r["n_chains"] = c
        return r

    def graph_nodes(self):
        """Compact node export (pb_graph_nodes / pb_graph_fetch_nodes / pb_graph_fetch_chain_records); call after graph_chains()."""
        W = words_for_k(self.k)
        nn = C.c_uint64()
        self._chk(self.lib.pb_graph_nodes(self.h, C.byref(nn)))
        m = nn.value
        r = dict(idx=np.zeros(max(m, 1), dtype=np.uint32), meta=np.zeros(max(m, 1), dtype=np.uint32), chain=np.zeros(max(m, 1), dtype=np.uint32),
                 nbr=np.zeros((max(m, 1), 2), dtype=np.uint32), nb8=np.zeros((max(m, 1), 8), dtype=np.uint32), keys=np.zeros((max(m, 1), W), dtype=np.uint64))
        self._chk(self.lib.pb_graph_fetch_nodes(self.h, r["idx"].ctypes.data, r["meta"].ctypes.data, r["chain"].ctypes.data, r["nbr"].ctypes.data,
                                                r["nb8"].ctypes.data, r["keys"].ctypes.data))
        return {key: v[:m] for key, v in r.items()}

    # ---- multi-k rounds (assemble.cpp:392-498) ----
    def reads_filter(self, k: int, edge_keys: np.ndarray) -> int:
        """pickupReadMatchedEdgeKmer (counter.h:869-910): edge_keys uint64 [n, W] canonical; returns the number of reads kept."""
        ek = np.ascontiguousarray(edge_keys, dtype=np.uint64)
        kept = C.c_uint64()
        self._chk(self.lib.pb_reads_filter(self.h, k, ek.ctypes.data, ek.shape[0] if ek.size else 0, C.byref(kept)))
        return kept.value

    def reads_restore(self):
        self._chk(self.lib.pb_reads_restore(self.h))

    def contig_table_build(self, k: int, bases: np.ndarray, offsets: np.ndarray, values: np.ndarray) -> int:
        """saveContig<NEXT> (graph.h:1655-1771): contig strings (codes 0..3) with one value each -> the resident next-k table."""
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        v = np.ascontiguousarray(values, dtype=np.uint16)
        nw = C.c_uint64()
        self._chk(self.lib.pb_contig_table_build(self.h, k, b.ctypes.data, o.ctypes.data, v.ctypes.data, len(v), C.byref(nw)))
        return nw.value

    def count_considering_contigs(self, k: int):
        """makeKmerReadDistributionConsideringPreviousGraph (counter.h:662-750); afterwards retain() / fetch_all() work on the result."""
        hist = np.zeros(PB_OCC_BINS, dtype=np.uint64)
        mx, nd = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.pb_count_considering_contigs(self.h, k, hist.ctypes.data, C.byref(mx), C.byref(nd)))
        self.k, self.occ_hist, self.max_occ, self.n_distinct = k, hist, mx.value, nd.value
        return hist

    # ---- kmer_divide's look-ups (kmer_divide.cpp:150-197) ----
    def occ_table_load(self, k: int, keys: np.ndarray, counts: np.ndarray):
        kk = np.ascontiguousarray(keys, dtype=np.uint64)
        cc = np.ascontiguousarray(counts, dtype=np.uint16)
        self._chk(self.lib.pb_occ_table_load(self.h, k, kk.ctypes.data, cc.ctypes.data, len(cc)))

    def occ_lookup(self, bases: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        out = np.zeros(max(int(o[-1] - o[0]), 1), dtype=np.uint16)
        self._chk(self.lib.pb_occ_lookup(self.h, b.ctypes.data, o.ctypes.data, len(o) - 1, out.ctypes.data))
        return out[:int(o[-1] - o[0])]

    # ---- node-parallel cleaning decisions (graph.h:829-926, 2032-2082, 712-746) ----
    def nodes_clean_decide(self, k, head, tail, length, cov, out, tkeys, tids, jkeys, jcov, jout, branch_th, length_cut=0, coverage_cut=0):
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        head, tail, tkeys, jkeys = a(head, np.uint64), a(tail, np.uint64), a(tkeys, np.uint64), a(jkeys, np.uint64)
        length, cov, out = a(length, np.uint64), a(cov, np.uint16), a(out, np.uint8)
        tids, jcov, jout = a(tids, np.uint32), a(jcov, np.uint16), a(jout, np.uint8)
        dec = np.zeros(max(len(length), 1), dtype=np.uint8)
        self._chk(self.lib.pb_nodes_clean_decide(self.h, k, len(length), head.ctypes.data, tail.ctypes.data, length.ctypes.data, cov.ctypes.data, out.ctypes.data,
                                                 len(tids), tkeys.ctypes.data, tids.ctypes.data, len(jcov), jkeys.ctypes.data, jcov.ctypes.data, jout.ctypes.data,
                                                 float(branch_th), int(length_cut), int(coverage_cut), dec.ctypes.data))
        return dec[:len(length)]

    def edit_distances(self, bases: np.ndarray, offsets: np.ndarray) -> np.ndarray:
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = (len(o) - 1) // 2
        d = np.zeros(max(n, 1), dtype=np.uint64)
        self._chk(self.lib.pb_edit_distances(self.h, b.ctypes.data, o.ctypes.data, n, d.ctypes.data))
        return d[:n]

    def gap_graphs(self, k: int, min_cov: int, seq_first: np.ndarray, seq_off: np.ndarray, codes: np.ndarray) -> dict:
        """pb_gap_graphs_build + _fetch: the local de Bruijn graphs of a batch of gaps (counter.h:1015-1056, graph.h:2116-2345); nodes in the
        reference's insertion order, as arrays (node_first, kind, cov, out, length, word_off, words)."""
        sf = np.ascontiguousarray(seq_first, dtype=np.uint64)
        so = np.ascontiguousarray(seq_off, dtype=np.uint64)
        cd = np.ascontiguousarray(codes, dtype=np.uint8)
        n_gaps, n_seqs = len(sf) - 1, len(so) - 1
        nn, nw = C.c_uint64(), C.c_uint64()
        self._chk(self.lib.pb_gap_graphs_build(self.h, k, min_cov, n_gaps, sf.ctypes.data, n_seqs, so.ctypes.data, cd.ctypes.data, C.byref(nn), C.byref(nw)))
        out = {"node_first": np.zeros(n_gaps + 1, dtype=np.uint64), "kind": np.zeros(nn.value, dtype=np.uint8), "cov": np.zeros(nn.value, dtype=np.uint16),
               "out": np.zeros(nn.value, dtype=np.uint8), "length": np.zeros(nn.value, dtype=np.uint32), "word_off": np.zeros(nn.value + 1, dtype=np.uint64),
               "words": np.zeros(max(nw.value, 1), dtype=np.uint64)}
        self._chk(self.lib.pb_gap_graphs_fetch(self.h, *(out[n].ctypes.data for n in ("node_first", "kind", "cov", "out", "length", "word_off", "words"))))
        out["words"] = out["words"][:nw.value]
        return out

    def kernel_ms(self, stage: str) -> float:
        ms = C.c_float()
        self._chk(self.lib.pb_last_kernel_ms(self.h, stage.encode(), C.byref(ms)))
        return ms.value
