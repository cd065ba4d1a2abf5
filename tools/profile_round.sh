#!/bin/bash
# Round profiles (run on the GPU box from the repo root): kernel trace of bench.py, SQ counter passes, HBM traffic passes.
# usage: bash tools/profile_round.sh r03     -> gpurun_out/prof_r03/...; copy the summaries into profiles/ afterwards (tools/profile_collect.py)
set -e
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
# 1. per-kernel time of the default bench command (with its CPU baseline, end-to-end CLI legs and shard shapes)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py > "$out/bench_profiled.json" 2> "$out/bench_profiled.err"
echo "trace done"
# 2. the same command without the profiler
python3 bench.py > "$out/bench_plain.json" 2> "$out/bench_plain.err"
echo "plain done"
# 3. SQ counters of the counting kernels (three passes of 8 counters)
bash tools/pmc_sq.sh "$out/pmc_sq" > "$out/pmc_sq.log" 2>&1
python3 tools/pmc_sq_summary.py "$out/pmc_sq" "$out/pmc_sq_summary.json"
echo "sq done"
# 4. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/pmc_$c" -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > "$out/pmc_$c.log" 2>&1
done
f=$(ls $out/pmc_FETCH_SIZE/*/*counter_collection.csv | head -1); w=$(ls $out/pmc_WRITE_SIZE/*/*counter_collection.csv | head -1)
python3 tools/pmc_traffic.py "$f" "$w" "$out/hbm_traffic_superk.json" > "$out/pmc_traffic.log"
echo "traffic done"
