#!/bin/bash
# Development aid: wall time of `assemble -kmer_occ_only` on the bench workload as plain FASTQ and as .gz (the reference's own way in:
# one `gzip -cd` pipe per file, common.cpp:120-156).  usage (GPU box, repo root): bash tools/gz_ingest_time.sh
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
D=$(mktemp -d)
python3 - "$D" <<'P'
import sys
sys.path.insert(0, ".")
from platanus_b_amd import synth
synth.write_pair_files(sys.argv[1] + "/in", 5_000_000, 100.0, 1)
P
B=$PWD/platanus_b_amd/bin/platanus_b
cd "$D"
ls -la in_1.fq in_2.fq | awk '{print $5, $9}'
for rep in 1 2; do echo "plain FASTQ:"; ( time "$B" assemble -kmer_occ_only -o plain -f in_1.fq in_2.fq > /dev/null 2> plain.err ) 2>&1 | grep real; done
( time gzip -1 -k in_1.fq ) 2>&1 | grep real; gzip -1 -k in_2.fq
ls -la in_1.fq.gz | awk '{print $5, $9}'
( time gzip -cd in_1.fq.gz > /dev/null ) 2>&1 | grep real
for rep in 1 2; do echo "gzip input:"; ( time "$B" assemble -kmer_occ_only -o gz -f in_1.fq.gz in_2.fq.gz > /dev/null 2> gz.err ) 2>&1 | grep real; done
PB_TIMING=1 "$B" assemble -kmer_occ_only -o gz2 -f in_1.fq.gz in_2.fq.gz 2>&1 | grep -A12 "STAGE WALL" | head -14
md5sum plain_kmer_occ.bin gz_kmer_occ.bin | awk '{print $1}'
cd /; rm -rf "$D"
