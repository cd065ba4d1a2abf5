"""The C-ABI library loads and exports every symbol include/pbgpu.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pbgpu.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from platanus_b_amd import binding
    if not os.path.exists(binding.LIB_PATH):
        g.build()
    lib = ctypes.CDLL(binding.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/pbgpu.h but not exported"


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product path must fail loudly, never fall back to the oracle."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from platanus_b_amd.binding import GpuCounter, PbError
    with pytest.raises(PbError):
        GpuCounter(0)


def test_product_package_never_imports_oracle():
    """No source file of the product includes, imports, loads or executes anything under oracle/ (comments may name it)."""
    pkg = os.path.join(ROOT, "platanus_b_amd")
    bad = re.compile(r'#\s*include\s*[<"][^>"]*oracle|^\s*(from|import)\s+oracle|pyoracle|liboracle|pb_oracle|oracle/_ref|CDLL\([^)]*oracle', re.M)
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h", ".inc")) or fn == "Makefile":
                src = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert not bad.search(src), os.path.join(dirpath, fn)
