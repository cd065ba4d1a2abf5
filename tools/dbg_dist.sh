set -e
cd /tmp && rm -rf dd && mkdir dd && cd dd
python3 -c "
import sys; sys.path.insert(0,'$GRAFT_REPO_ROOT')
from platanus_b_amd import synth
synth.write_pair_files('in', 200000, 40, 7, err=0.01, n_rate=0.001)
"
B=$GRAFT_REPO_ROOT/platanus_b_amd/bin/platanus_b
$B assemble -k 32 -t 1 -m 1 -n 2 -o s1 -f in_1.fq in_2.fq 2> s1.err
$B assemble -k 32 -t 1 -m 1 -n 2 -gpus 2 -share_gpu -o s2 -f in_1.fq in_2.fq 2> s2.err
diff <(grep -v "^platanus\|assemble\|Vm\|^$" s1.err) <(grep -v "^platanus\|assemble\|Vm\|^$" s2.err) | head -40
md5sum s1_contig.fa s2_contig.fa
echo ---- first lines; head -30 s1.err | grep -v "^$"; echo ----; head -32 s2.err | grep -v "^$"
