#!/bin/bash
# development: stage times of the device CLI on BASELINE configs[2] (5 Mbp x 100x, full multi-k)
set -e
D=$(mktemp -d /tmp/cfg3_XXXX)
python3 -c "
import sys; sys.path.insert(0,'.')
from platanus_b_amd import synth
synth.write_pair_files('$D/in', ${1:-5000000}, ${2:-100}, 1)
"
# usage: tools/time_cfg3.sh [genome] [coverage] [extra CLI arguments, e.g. -device_parse]
EXTRA="${@:3}"
for i in 1 2 3; do
  t0=$(date +%s.%N)
  PB_TIMING=1 platanus_b_amd/bin/platanus_b assemble -k 32 $EXTRA -o $D/out -f $D/in_1.fq $D/in_2.fq 2> $D/log$i > /dev/null
  t1=$(date +%s.%N)
  python3 -c "
import sys, re
t0, t1 = float(sys.argv[1]), float(sys.argv[2])
log = open(sys.argv[3]).read()
a = float(re.search(r'epoch at entry\t([0-9.]+)', log).group(1)); b = float(re.search(r'epoch at return\t([0-9.]+)', log).group(1))
print(f'run $i: wall {t1 - t0:.3f} s = start-up {a - t0:.3f} + run {b - a:.3f} + teardown {t1 - b:.3f}')" $t0 $t1 $D/log$i
done
sed -n "/STAGE WALL/,\$p" $D/log3
md5sum $D/out_contig.fa $D/out_32merFrq.tsv
rm -rf "$D"
