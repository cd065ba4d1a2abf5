"""How bench.py starts (no GPU needed): `python bench.py --gpus N` launches its own ranks as a child `torch.distributed.run`, and a
WORLD_SIZE that contradicts --gpus is an error, not something to ignore."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

_CLEAN = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")


def test_bench_refuses_a_world_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr and not p.stdout.strip()


def test_bench_gpus_n_launches_its_ranks_as_a_child():
    """Without a GPU the two ranks it starts each end with the product's "needs a GPU" message: that they ran at all is the point
    (round 2's bench.py printed a hint and exited).  On a GPU box the GPU suite covers the same path with a real run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("covered by tests/test_bench_contract.py on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in _CLEAN}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--genome", "100000"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and not p.stdout.strip()
    assert p.stderr.count("bench.py needs a GPU") >= 2, p.stderr[-1500:]
    assert "launch with" not in p.stderr
