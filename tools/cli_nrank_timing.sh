#!/bin/bash
# Development aid: `assemble -gpus N -share_gpu` on the bench workload (5 Mbp, 100x), k0 counted both ways; stage wall times (PB_TIMING).
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
N=${1:-2}
D=$(mktemp -d)
python3 - "$D" <<'P'
import sys
sys.path.insert(0, ".")
from platanus_b_amd import synth
synth.write_pair_files(sys.argv[1] + "/in", 5_000_000, 100.0, 1)
P
for design in 0 1; do
  for rep in 1 2; do
    echo "== ranks $N dist_design=$design rep $rep"
    ( cd "$D" && PB_TIMING=1 "$OLDPWD/platanus_b_amd/bin/platanus_b" assemble -gpus "$N" -share_gpu -gpu_opt dist_design=$design -o out$design -f in_1.fq in_2.fq 2>&1 | grep -A40 "STAGE WALL" | grep -v "^$" | head -40 )
  done
done
( cd "$D" && PB_TIMING=1 "$OLDPWD/platanus_b_amd/bin/platanus_b" assemble -o one -f in_1.fq in_2.fq 2>&1 | grep -A30 "STAGE WALL" | head -30 )
md5sum "$D"/out0_contig.fa "$D"/out1_contig.fa "$D"/one_contig.fa
rm -rf "$D"
