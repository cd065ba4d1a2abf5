"""A dead or failed rank ends an N-rank run instead of hanging it (no GPU needed: the launcher side of include/pbgpu.h makes no HIP
call).  tests/native/dist_liveness.cpp is compiled against libpbgpu.so and misbehaves on request; the CLI itself is checked for what
it says and returns when every rank fails the same way."""
import os
import signal
import subprocess
import time

import pytest

from conftest import ROOT

LIB_DIR = os.path.join(ROOT, "platanus_b_amd")
CLI = os.path.join(LIB_DIR, "bin", "platanus_b")


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    import __graft_entry__
    __graft_entry__.build()
    exe = str(tmp_path_factory.mktemp("native") / "dist_liveness")
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "native", "dist_liveness.cpp"), "-L" + LIB_DIR, "-lpbgpu", "-lpthread",
                    "-Wl,-rpath," + LIB_DIR, "-Wl,-rpath-link,/opt/rocm/lib"], check=True)
    return exe


def _alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    with open(f"/proc/{pid}/stat") as f:          # a zombie is not alive
        return f.read().rsplit(")", 1)[1].split()[0] != "Z"


@pytest.mark.parametrize("world", [2, 4])
def test_all_ranks_well(driver, world):
    p = subprocess.run([driver, str(world), "ok"], capture_output=True, text=True, timeout=30)
    assert p.returncode == 0 and p.stderr == ""


@pytest.mark.parametrize("world", [2, 3])
def test_rank_killed_while_the_others_wait_in_a_barrier(driver, world):
    """SIGKILL for rank 1 while every other rank sits in the next barrier: the launcher leaves within seconds, non-zero, and says which
    rank died of what (round 2's pthread barrier waited forever)."""
    p = subprocess.Popen([driver, str(world), "hang1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pids = {}
    for _ in range(2):
        word, pid = p.stdout.readline().split()
        pids[word] = int(pid)
    time.sleep(0.3)                               # the others are inside the barrier by now
    t0 = time.time()
    os.kill(pids["ready"], signal.SIGKILL)
    out, err = p.communicate(timeout=15)
    assert time.time() - t0 < 5
    assert p.returncode == 2 and err.strip() == "Error: rank 1 was killed by signal 9", (p.returncode, err)


def test_ranks_die_with_the_launcher(driver):
    p = subprocess.Popen([driver, "3", "hang1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    pids = {}
    for _ in range(2):
        word, pid = p.stdout.readline().split()
        pids[word] = int(pid)
    assert pids["launcher"] == p.pid
    p.kill()
    p.wait()
    deadline = time.time() + 5
    while _alive(pids["ready"]) and time.time() < deadline:
        time.sleep(0.05)
    assert not _alive(pids["ready"])              # PR_SET_PDEATHSIG: no orphan keeps a GPU


def test_a_rank_with_a_reason_of_its_own_is_the_one_message(driver):
    p = subprocess.run([driver, "4", "fail1"], capture_output=True, text=True, timeout=30)
    assert p.returncode == 7 and p.stderr.strip() == "Error(7): rank 1 has a reason of its own"
    p = subprocess.run([driver, "4", "fail0"], capture_output=True, text=True, timeout=30)
    assert p.returncode == 5 and p.stderr.strip() == "Error(5): the launcher has a reason of its own"


def test_a_rank_that_crashes_stops_the_group(driver):
    t0 = time.time()
    p = subprocess.run([driver, "3", "segv1"], capture_output=True, text=True, timeout=30)
    assert time.time() - t0 < 5
    assert p.returncode == 2 and p.stderr.strip() == "Error: rank 1 was killed by signal 11"


def _cli(args, tmp_path, timeout=60):
    fq = tmp_path / "r.fq"
    if not fq.exists():
        fq.write_text("@r0\nACGTACGTACGTACGTACGTACGTACGTACGTACGTACGTAC\n+\nIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIIII\n")
    return subprocess.run([CLI, "assemble", *args, "-o", str(tmp_path / "out"), "-f", str(fq)], capture_output=True, text=True, timeout=timeout)


def _errors(stderr):
    return [ln for ln in stderr.splitlines() if ln.startswith("Error")]


def test_cli_checks_what_needs_no_gpu_before_it_forks(tmp_path):
    import __graft_entry__
    __graft_entry__.build()
    p = _cli(["-gpus", "3", "-share_gpu", "-kmer_occ_only"], tmp_path)
    assert p.returncode == 2 and len(_errors(p.stderr)) == 1 and "-kmer_occ_only with -gpus > 1" in p.stderr
    p = _cli(["-gpus", "2", "-share_gpu", "-gpu_opt", "no_such_option=1"], tmp_path)
    assert p.returncode == 2 and len(_errors(p.stderr)) == 1 and "no_such_option" in p.stderr
    p = _cli(["-gpu_opt", "count_path=7"], tmp_path)
    assert p.returncode == 2 and len(_errors(p.stderr)) == 1


def test_cli_n_ranks_failing_alike_say_it_once():
    """Without a device every rank fails in pb_create: one message, the launcher's, and its exit status -- no rank kills another
    (round 2: the ranks SIGKILLed each other and the status was 2 or -9 by timing)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a host without a GPU")
    import tempfile
    import pathlib
    with tempfile.TemporaryDirectory() as td:
        for _ in range(3):
            p = _cli(["-gpus", "4", "-share_gpu"], pathlib.Path(td))
            assert p.returncode == 2, (p.returncode, p.stderr[-500:])
            assert len(_errors(p.stderr)) == 1 and "no HIP device" in p.stderr, p.stderr[-800:]
