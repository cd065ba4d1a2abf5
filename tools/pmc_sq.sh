#!/bin/bash
# SQ counter passes of the counting pass (bench.py, configs[1]); one rocprofv3 run per pass (8 SQ slots), program directly after "--".
# usage (on the GPU box): bash tools/pmc_sq.sh <outdir>      -> <outdir>/pass{1,2,3}/.../*counter_collection.csv
set -e
out=${1:-gpurun_out/pmc_sq_r03}
mkdir -p "$out"
export TMPDIR=/tmp
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P2="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAVES"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace --output-format csv -d "$out/pass$i" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > "$out/pass$i.log" 2>&1
  echo "pass $i done"
done
