#!/bin/bash
# measurement: the graph part of GapCloseDBG<20, 80>::gapAssemble for a batch of synthetic gaps through the product (tests/native/gap_assemble_host.cpp:
# host/gap_graphs.hpp gap_assemble_batch) and, for a bounded share of the gaps, through the reference's own code (oracle/_ref/gapclose_ref INPUT 80).
# usage (GPU box): bash tools/gap_assemble_time.sh [n_gaps]      -> gpurun_out/r3_gap_assemble_time.txt
set -e
N=${1:-20000}
D=$(mktemp -d /tmp/gapasm_XXXX)
g++ -std=c++17 -O2 -Wno-sign-compare -o $D/host tests/native/gap_assemble_host.cpp -Lplatanus_b_amd -lpbgpu -lpthread -Wl,-rpath,$(pwd)/platanus_b_amd -Wl,-rpath-link,/opt/rocm/lib
python3 - <<PY
import struct, numpy as np
rng = np.random.default_rng(9)
def write(path, n):
    seqs, first = [], [0]
    for _ in range(n):
        G = int(rng.integers(200, 1500)); genome = rng.integers(0, 4, G).astype(np.uint8)
        for _ in range(max(1, G * 20 // 150)):
            L = min(150, G); s = int(rng.integers(0, G - L + 1)); r = genome[s:s + L].copy()
            e = rng.random(L) < 0.01; r[e] = (r[e] + 1) & 3
            seqs.append(r)
        first.append(len(seqs))
    off = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    with open(path, "wb") as f:
        f.write(struct.pack("<IqQ", 20, 2, n)); f.write(np.array(first, dtype=np.uint64).tobytes())
        f.write(struct.pack("<Q", len(seqs))); f.write(off.tobytes()); f.write(np.concatenate(seqs).tobytes())
write("$D/all.bin", $N)
rng = np.random.default_rng(9)
write("$D/part.bin", min($N, 2000))
PY
mkdir -p gpurun_out
{
  echo "product, $N gaps (k = 20 / 80, min coverage 2):"; $D/host $D/all.bin 80 time
  echo "product, the first 2000 gaps:"; $D/host $D/part.bin 80 time
  if [ -x oracle/_ref/gapclose_ref ]; then
    echo "reference GapCloseDBG<20, 80>::gapAssemble, the first 2000 gaps, one thread (process start + table dump included):"
    t0=$(date +%s.%N); oracle/_ref/gapclose_ref $D/part.bin 80 > $D/ref_dump.txt; t1=$(date +%s.%N)
    python3 -c "print(f'{$t1 - $t0:.2f} s wall')"
    $D/host $D/part.bin 80 > $D/own_dump.txt
    cmp $D/ref_dump.txt $D/own_dump.txt && echo "dumps identical (2000 gaps)"
  fi
} > gpurun_out/r3_gap_assemble_time.txt 2>&1
rm -rf $D
