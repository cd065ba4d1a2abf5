"""bench.py prints ONE JSON line with the fields the driver and the judge read (metric / value / unit / n_gpus / steps / warmup /
ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config, plus `roofline`); checked on a small workload."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--genome", "300000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline"):
        assert key in d, key
    assert d["metric"] == "k-mers/sec" and d["unit"] == "k-mers/s" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "u64"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # value = windows / wall time; achieved = algorithmic bytes / kernel time of the same pass (kernel time <= wall time per step)
    inst = d["config"]["kmer_instances_per_gpu"]
    assert abs(d["value"] - inst / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert r["kernel_ms"] <= d["ms_per_step"] and abs(r["achieved"] - inst * 12.25 / (r["kernel_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-6
    assert r["kernels_ms"]["part1"] > 0                   # the pass bench.py measures is the super-k-mer pass


@pytest.mark.parametrize("design", ["records", "counts"])
def test_bench_n_rank_step_on_a_world_of_one(design):
    """The N > 1 step of bench.py (both designs) with torch.distributed over RCCL on a world of one: the same code the driver's
    `--gpus N` runs execute, minus the other ranks."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--genome", "300000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--exchange-with-one-rank", "--exchange", design], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.strip()][-1])
    assert d["n_gpus"] == 1 and d["config"]["torch_distributed"] == {"backend": "nccl", "world_size": 1}
    assert d["value"] > 0 and d["roofline"]["kernel_ms"] > 0
    assert d["config"]["distinct_kmers"] > 200_000


def test_bench_starts_its_own_ranks_and_times_the_n_rank_cli():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the driver's scaling runs call it) starts its ranks itself:
    a child `python -m torch.distributed.run ...` before anything touches the GPU.  Here N = 1 is forced through the same launch code
    (--self-launch); the N > 1 step runs on a world of one and the end-to-end leg runs the N-rank CLI (two ranks sharing the device)
    beside the single-GPU CLI on the FASTQ files the rank wrote, comparing every output file."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--self-launch", "--exchange-with-one-rank", "--genome", "300000",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["torch_distributed"] == {"backend": "nccl", "world_size": 1} and d["value"] > 0
    e = d["end_to_end"]
    assert "error" not in e, e
    assert e["assemble_gpus2_wall_s"] > 0 and e["assemble_gpus1_wall_s"] > 0
    assert e["outputs_identical_to_the_single_gpu_run"] is True and "contig.fa" in e["output_md5"]

