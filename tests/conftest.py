import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def case_reads(case):
    """Reads of a golden case in the reference's -t 1 temp-file order: all of file 1, then all of file 2."""
    from platanus_b_amd import synth
    r1, r2 = synth.make_reads(case["genome"], case["cov"], case["seed"], err=case["err"], n_rate=case["n_rate"], het=case.get("het", 0.0))
    codes = np.concatenate([r1, r2])
    n, L = codes.shape
    return codes, np.arange(n + 1, dtype=np.uint64) * np.uint64(L)


@pytest.fixture(scope="session")
def oracle_built():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
