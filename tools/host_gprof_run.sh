#!/bin/bash
# development: the CLI's HOST code under gprof (platanus_b_amd/bin/platanus_b_pg: the host sources compiled with -pg) on a 40 Mbp x 100x set --
# where the node-level graph stages (DFS replay, node cleaning) stop being negligible.  Leaves gpurun_out/r3_gmon.out (gprof bin/platanus_b_pg <that>).
set -e
D=$(mktemp -d /tmp/pg_XXXX)
python3 - <<PY
import sys
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
from concurrent.futures import ProcessPoolExecutor
import nrank_full_configs as m
G = ${1:-40000000}; cov = 100.0; n_pairs = int(G * cov) // 300; parts = 16
jobs = [("$D", G, cov, 5, p, n_pairs * p // parts, n_pairs * (p + 1) // parts) for p in range(parts)]
with ProcessPoolExecutor(16) as ex:
    list(ex.map(m.write_part, jobs))
PY
cd $D
PB_TIMING=1 /root/repo/platanus_b_amd/bin/platanus_b_pg assemble -k 32 -m 256 -o out -f in*.fq 2> log.txt > /dev/null
sed -n '/STAGE WALL/,$p' log.txt | head -30
cp gmon.out /root/repo/gpurun_out/r3_gmon.out 2>/dev/null || true
cd /; rm -rf $D
