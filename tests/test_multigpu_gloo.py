"""N>1 path on CPU: world_size 2 over gloo.  Reads are range-partitioned, every rank counts its range, the
(key,count) records go to their owner rank in one all_to_all (platanus_b_amd.multigpu), owners merge, the
histogram is all-reduced.  The union of the owner shards must equal the single-process table."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


@pytest.mark.parametrize("k,port,mode", [(32, 29541, "split"), (45, 29542, "split"), (32, 29543, "v1")])
def test_owner_exchange_world2_matches_single_process(k, port, mode, tmp_path, oracle_built):
    out = tmp_path / "res.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_gloo_worker.py"), str(k), str(out), mode]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.load(open(out))
    from platanus_b_amd import synth
    r1, r2 = synth.make_reads(20_000, 30, 9, n_rate=0.001)
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    oc = oracle_built.OracleCounter()
    oc.count(codes.reshape(-1), np.arange(n + 1, dtype=np.uint64) * np.uint64(L), k)
    hist, _ = oc.occ_hist()
    keys, counts = oc.retain(1)
    assert res["n"] == len(keys)
    assert res["hist"] == [int(x) for x in hist]
    want = sorted([[int(w) for w in kk] + [int(c)] for kk, c in zip(keys, counts)])
    assert res["records"] == want
    # gather of the retained shards (cutoff 2): the sorted set the graph stage gets, in the oracle's order
    k2, c2 = oc.retain(2)
    assert res["retained2"] == [[int(w) for w in kk] + [int(c)] for kk, c in zip(k2, c2)]


def test_owner_function_is_the_reference_stripe_range():
    """owner = ((key & 1023) * world) >> 10: contiguous ranges of the reference's 1024 stripes (counter.h:461)."""
    from platanus_b_amd.multigpu import owner_of
    stripes = np.arange(1024, dtype=np.uint64)
    for world in (1, 2, 4, 8):
        o = owner_of(stripes | np.uint64(0xABCD0000), world)
        assert (np.diff(o) >= 0).all() and o.min() == 0 and o.max() == world - 1
        assert all((o == r).sum() == 1024 // world for r in range(world))
