"""Writes tests/golden/gapclose/<case>.txt: what the REFERENCE's own countKmerForGapClose + makeBruijnGraphForGapClose leave in their tables
for the synthetic gaps of tests/gapclose_cases.py (oracle/_ref/gapclose_ref: our harness ref_harness/gapclose_ref.cpp over the reference's
sources where they lie; `make -C oracle ref`).  Run in the build container (needs /root/reference): python tests/golden/make_gapclose_golden.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle          # noqa: E402
import gapclose_cases                # noqa: E402

out_dir = os.path.join(ROOT, "tests", "golden", "gapclose")
os.makedirs(out_dir, exist_ok=True)
for name in gapclose_cases.CASES:
    k, mc, first, off, codes = gapclose_cases.make_gaps(name)
    dumps = pyoracle.gap_graphs_reference(k, mc, first, off, codes)
    with open(os.path.join(out_dir, name + ".txt"), "w") as f:
        for g, d in enumerate(dumps):
            f.write(f"G {g}\n{d}")
    print(name, "k", k, "gaps", len(dumps), "lines", sum(d.count("\n") for d in dumps))

for name, (kl, kh, mc, case) in gapclose_cases.ASSEMBLE_CASES.items():
    k, _, first, off, codes = gapclose_cases.make_gaps(case)
    dumps = pyoracle.gap_assemble_reference(kl, kh, mc, first, off, codes)          # GapCloseDBG<kl, kh>::gapAssemble of the reference itself
    with open(os.path.join(out_dir, name + ".txt"), "w") as f:
        for g, d in enumerate(dumps):
            f.write(f"G {g}\n{d}")
    print(name, "gaps", len(dumps), "lines", sum(d.count("\n") for d in dumps))

for name in gapclose_cases.CLOSE_CASES:
    kl, kh, first, off, codes, flanks = gapclose_cases.make_gaps_with_flanks(name)
    dumps = pyoracle.gap_assemble_reference(kl, kh, 2, first, off, codes, flanks)      # ... and the reference's closing decisions (gapClose.cpp:716-760 without OLC)
    with open(os.path.join(out_dir, name + ".txt"), "w") as f:
        for g, d in enumerate(dumps):
            f.write(f"G {g}\n{d}")
    print(name, "gaps", len(dumps), "lines", sum(d.count("\n") for d in dumps))
