"""Development aid: degenerate inputs through the GPU CLI, the oracle CLI and (when it was built) the real reference: exit codes and
output files must agree.  usage: python tools/edge_cli.py"""
import filecmp
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GPU = os.path.join(ROOT, "platanus_b_amd", "bin", "platanus_b")
ORC = os.path.join(ROOT, "oracle", "pb_oracle")
REF = os.path.join(ROOT, "oracle", "_ref", "platanus_b")


def fq(reads):
    return "".join(f"@r{i}\n{r}\n+\n{'I' * len(r)}\n" for i, r in enumerate(reads))


CASES = {
    "reads_shorter_than_k": fq(["ACGTACGTAC", "TTGCA", "GGGGCCCCAAAATTTT"]),
    "single_read": fq(["ACGTTGCAAGGCTTAACCGGTTAAGGCCTTAGCAGCATCGACTAGCTAGCATCAGCTACGACTACGACTAGCATCGACTACGATCAGCACTAGC"]),
    "only_n": fq(["N" * 100, "N" * 150]),
    "empty_read_between": "@a\n\n+\n\n" + fq(["ACGTTGCAAGGCTTAACCGGTTAAGGCCTTAGCAGCATCGACTAGCTAGCATCAGC"] * 6),
    "identical_reads": fq(["ACGTTGCAAGGCTTAACCGGTTAAGGCCTTAGCAGCATCGACTAGCTAGCATCAGCTACGACTACGACTAGCATCGACTACGATCAGCACTAGC"] * 40),
    "homopolymer": fq(["A" * 150] * 30 + ["T" * 150] * 30),
    "palindromes": fq(["ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT" * 2] * 25 + ["GAATTCGAATTCGAATTCGAATTCGAATTCGAATTCGAATTCGAATTC" * 2] * 25),
    "empty_file": "",
    "lower_case_second_line": fq(["acgttgcaaggcttaaccggttaaggccttagcagcatcgactagctagcatcagc"] * 6),
    "not_a_read_file": "hello\nworld\n",
    "fastq_without_plus": "@r0\nACGT\nIIII\n",
    "fasta_lower_case_later": ">a\nACGTTGCAAGGCTTAACCGGTTAAGGCCTTAGCAGCATCGACTAGCTAGCATCAGC\n" + ">b\nacgttgcaaggcttaaccggttaaggccttagcagcatcgactagctagcatcagc\n" * 5,
    "read_of_499999_bases": "LONG:499999",
    "read_of_500000_bases": "LONG:500000",
    "eleven_input_files": "MULTI:11",
    "missing_file": None,
    "gzip_under_plain_name": b"gz",
}
READS = fq(["ACGTTGCAAGGCTTAACCGGTTAAGGCCTTAGCAGCATCGACTAGCTAGCATCAGCTACGACTACGACTAGCATCGACTACGATCAGCACTAGC"] * 20)


def main():
    bins = [("gpu", GPU), ("oracle", ORC)] + ([("reference", REF)] if os.path.exists(REF) else [])
    bad = 0
    for name, text in CASES.items():
        with tempfile.TemporaryDirectory() as td:
            files = ["../in.fq"]
            if text == b"gz":
                import gzip
                with gzip.open(os.path.join(td, "in.fq"), "wt") as f:
                    f.write(READS)
            elif isinstance(text, str) and text.startswith("MULTI:"):       # more files than the CLI parses at once (8): FASTQ, FASTA and gzip mixed
                import gzip
                import random
                rnd = random.Random(11)
                genome = "".join(rnd.choice("ACGT") for _ in range(3000))
                n_files = int(text[6:])
                for fi in range(n_files):
                    reads = []
                    for _ in range(150):
                        a = rnd.randrange(0, len(genome) - 100)
                        reads.append(genome[a:a + 100])
                    fname = os.path.join(td, f"in{fi}.fq")
                    body = fq(reads) if fi % 3 != 1 else "".join(f">r{i}\n{r}\n" for i, r in enumerate(reads))
                    if fi % 4 == 3:
                        with gzip.open(fname, "wt") as f:
                            f.write(body)
                    else:
                        open(fname, "w").write(body)
                files = [f"../in{fi}.fq" for fi in range(n_files)]
            elif isinstance(text, str) and text.startswith("LONG:"):      # the longest read the reference accepts is 499,999 bases (common.h:465-467)
                import random
                rnd = random.Random(7)
                seq = "".join(rnd.choice("ACGT") for _ in range(int(text[5:])))
                open(os.path.join(td, "in.fq"), "w").write(fq([seq] * 10))
            elif text is not None:
                open(os.path.join(td, "in.fq"), "w").write(text)
            res = {}
            for who, b in bins:
                d = os.path.join(td, who)
                os.mkdir(d)
                # (a 500 kbp read would send the multi-k schedule beyond k = 160: single k for the two long-read cases)
                extra = ["-s", "0"] if isinstance(text, str) and text.startswith("LONG:") else []
                p = subprocess.run([b, "assemble", "-k", "21", *extra, "-t", "1", "-m", "1", "-o", "out", "-f", *files], cwd=d, capture_output=True, text=True,
                                   timeout=300)
                res[who] = (p.returncode, d, p.stderr[-300:])
            base = res["reference"] if "reference" in res else res["oracle"]
            line = [f"{who}: rc {res[who][0]} files {sorted(os.listdir(res[who][1]))}" for who, _ in bins]
            ok = True
            for who, _ in bins:
                if res[who][0] != base[0]:
                    ok = False
                if res[who][0] == 0 and base[0] == 0:
                    for fn in os.listdir(base[1]):
                        f2 = os.path.join(res[who][1], fn)
                        if not os.path.exists(f2) or not filecmp.cmp(os.path.join(base[1], fn), f2, shallow=False):
                            ok = False
                            line.append(f"{who}: {fn} differs")
            bad += not ok
            print(("ok       " if ok else "MISMATCH ") + name + " | " + " | ".join(line), flush=True)
            if not ok:
                for who, _ in bins:
                    print(f"    {who} stderr tail: {res[who][2]!r}")
    print(f"{len(CASES)} cases, {bad} mismatches")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
