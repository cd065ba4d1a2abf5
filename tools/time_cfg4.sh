#!/bin/bash
# development: the one-GPU CLI on BASELINE configs[4] (100 Mbp x 100x, -m 256): stage wall times with the library's own pass report
# (-gpu_opt verbose=1), then the same command under rocprofv3 --kernel-trace --stats (device time per kernel against the stage walls).
# usage (GPU box): [NOPROF=1] [EXTRA="-gpu_opt ..."] bash tools/time_cfg4.sh [genome bp] [tag]
#   -> gpurun_out/<tag>_stages.txt, <tag>_host_samples.txt (tools/host_samples.py), <tag>_kernel_stats.csv
set -e
G=${1:-100000000}
TAG=${2:-r3_cfg4}
ROOT=$(pwd)
D=$(mktemp -d /tmp/c4_XXXX)
python3 - <<PY
import sys
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
from concurrent.futures import ProcessPoolExecutor
import nrank_full_configs as m
G = $G; cov = 100.0; n_pairs = int(G * cov) // 300; parts = 16
jobs = [("$D", G, cov, 5, p, n_pairs * p // parts, n_pairs * (p + 1) // parts) for p in range(parts)]
with ProcessPoolExecutor(16) as ex:
    list(ex.map(m.write_part, jobs))
PY
mkdir -p $ROOT/gpurun_out
cd $D
for i in 1 2; do
  PB_HOST_SAMPLES=$D/samples$i.txt PB_TIMING=1 $ROOT/platanus_b_amd/bin/platanus_b assemble -k 32 -m 256 -gpu_opt verbose=1 $EXTRA -o out$i -f in*.fq 2> log$i.txt > /dev/null
done
cp $D/samples2.txt $ROOT/gpurun_out/${TAG}_host_samples.txt
{ grep "^\[pbgpu\]" log2.txt | cut -c1-400; sed -n '/STAGE WALL/,$p' log2.txt; md5sum out2_contig.fa out2_32merFrq.tsv; } > $ROOT/gpurun_out/${TAG}_stages.txt
echo "stages written"
if [ "$NOPROF" = "1" ]; then cd /; rm -rf $D; exit 0; fi
PB_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof -- $ROOT/platanus_b_amd/bin/platanus_b assemble -k 32 -m 256 -o out3 -f in*.fq 2> log3.txt > /dev/null
S=$(ls $D/prof/*/*kernel_stats.csv | head -1)
cp "$S" $ROOT/gpurun_out/${TAG}_kernel_stats.csv
sed -n '/STAGE WALL/,$p' log3.txt > $ROOT/gpurun_out/${TAG}_stages_profiled.txt
cd /; rm -rf $D
