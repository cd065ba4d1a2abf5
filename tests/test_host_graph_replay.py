"""Host logic without a GPU: the product's DFS replay over adjacency arrays (what pb_graph_adjacency returns) must build
the same Straight/Junction tables -- same keys, strands, sequences, coverages, link flags, same container iteration order
-- as the k-mer-level DFS of makeInitialBruijnGraph (graph.h:289-579) restated in the oracle.  Includes inputs with
hairpins / palindromic k-mers (even k), cycles (circular genome reads) and dead ends."""
import numpy as np
import pytest

from conftest import case_reads


@pytest.mark.parametrize("k,min_occ", [(32, 2), (21, 1), (45, 2), (75, 1), (31, 3)])
def test_replay_equals_kmer_level_dfs(k, min_occ, manifest, oracle_built):
    codes, off = case_reads(manifest["g5k"])
    assert oracle_built.replay_check(codes.reshape(-1), off, k, min_occ) == 0


def test_replay_on_cycles_and_hairpins(oracle_built):
    rng = np.random.default_rng(3)
    reads = []
    circ = rng.integers(0, 4, 400).astype(np.uint8)               # circular sequence: a cycle in the graph
    doubled = np.concatenate([circ, circ, circ[:60]])
    for s in range(0, 700, 7):
        reads.append(doubled[s:s + 120])
    half = rng.integers(0, 4, 90).astype(np.uint8)                # hairpin: sequence followed by its reverse complement
    hp = np.concatenate([half, (3 - half)[::-1]])
    for s in range(0, 100, 3):
        reads.append(hp[s:s + 80])
    reads.append(np.tile(np.array([0, 3], dtype=np.uint8), 60))   # (AT)n: every 2nd k-mer is its own reverse complement
    reads.append(np.zeros(100, dtype=np.uint8))                   # poly-A self loop
    lens = np.array([len(r) for r in reads], dtype=np.uint64)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    codes = np.concatenate(reads)
    for k in (12, 16, 21, 32, 33, 40):
        assert oracle_built.replay_check(codes, off, k, 1) == 0, k
