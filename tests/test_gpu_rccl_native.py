"""The communicator of a one-process-per-GPU launch (include/pbsim3_amd.h pbsim_rccl_*: ncclGetUniqueId + ncclCommInitRank),
as a group of ONE -- all this box can offer (RCCL takes one rank per GPU): a whole job driven through it (PBSIM_COMM_ALWAYS=1:
the several-rank protocol with every exchange a real ncclAllGather / ncclAllReduce) delivers the bytes of the job without a
communicator, the id travels through a rendezvous file as well, and the `pbsim` binary's --rank / --world /
--rendezvous mode runs the command line on it.  What a group of one cannot show -- ranks waiting for each other -- is covered
by the host communicator with several contexts on the one GPU (tests/test_gpu_multi.py) and over gloo."""
import os
import subprocess
import sys

import pytest

import harness
from cases import CASES

pytestmark = pytest.mark.gpu
CLI = os.path.join(harness.ROOT, "pbsim3_amd", "bin", "pbsim")


def run_driver(which, tmp_path):
    env = dict(os.environ, PYTHONPATH=harness.ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(harness.ROOT, "tests", "rccl_native_driver.py"), which, str(tmp_path)],
                       capture_output=True, text=True, cwd=harness.ROOT, env=env, timeout=280)
    assert p.returncode == 0 and "RCCL-NATIVE-OK " + which in p.stdout, (p.stdout + p.stderr)[-3000:]


def test_job_through_init_rank_communicator_of_one(tmp_path):
    """tests/rccl_native_driver.py job: a job (many rounds per record) without a communicator, then through ncclCommInitRank
    communicators of one made from an in-process id and from a rendezvous file, PBSIM_COMM_ALWAYS=1: same bytes, plain and as
    gzip members; RCCL counts one rank; the job's exchanges are counted by the communicator; the id file is gone afterwards"""
    run_driver("job", tmp_path)


def test_id_and_argument_errors(tmp_path):
    run_driver("errors", tmp_path)


def test_cli_one_process_per_gpu_mode(tmp_path):
    """`pbsim --rank 0 --world 1 --rendezvous FILE`: the selftest, then a golden case through pbsim_cli_main on that communicator"""
    p = subprocess.run([CLI, "--rank", "0", "--world", "1", "--rendezvous", str(tmp_path / "r1"), "--comm-selftest"],
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "RCCL counts 1): ok" in p.stderr, p.stderr[-2000:]
    case = "wgs_errhmm-ont_quirk"
    work = tmp_path / "w"
    work.mkdir()
    p = subprocess.run([CLI] + harness.resolve(CASES[case]["args"]) + ["--prefix", str(work / "out"), "--no-gzip", "--rank", "0",
                                                                        "--world", "1", "--rendezvous", str(tmp_path / "r2")],
                       capture_output=True, text=True, cwd=str(work), timeout=280)
    assert p.returncode == 0, p.stderr[-4000:]
    outs = harness.collect(str(work))
    outs[".stderr"] = harness.strip_report(p.stderr).encode()
    want = harness.load_manifest()[f"{case}/philox"]
    assert sorted(outs) == sorted(want)
    for k, v in outs.items():
        assert harness.sha(v) == want[k]["sha256"], k
    # under a launcher: rank / world / device from its environment
    env = dict(os.environ, OMPI_COMM_WORLD_RANK="0", OMPI_COMM_WORLD_SIZE="1", OMPI_COMM_WORLD_LOCAL_RANK="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([CLI, "--rendezvous", str(tmp_path / "r3"), "--comm-selftest"], capture_output=True, text=True, timeout=280, env=env)
    assert p.returncode == 0 and "rank 0 of 1 processes, RCCL counts 1): ok" in p.stderr, p.stderr[-2000:]
    # the binary as its own launcher: one child process per rank (here one), golden bytes again
    work2 = tmp_path / "w2"
    work2.mkdir()
    p = subprocess.run([CLI] + harness.resolve(CASES[case]["args"]) + ["--prefix", str(work2 / "out"), "--no-gzip", "--processes", "1"],
                       capture_output=True, text=True, cwd=str(work2), timeout=280)
    assert p.returncode == 0, p.stderr[-4000:]
    outs2 = harness.collect(str(work2))
    outs2[".stderr"] = harness.strip_report(p.stderr).encode()
    assert {k: harness.sha(v) for k, v in outs2.items()} == {k: want[k]["sha256"] for k in want}
    both = subprocess.run([CLI, "--processes", "2", "--devices", "0,0"], capture_output=True, text=True)
    assert both.returncode == 255 and "--processes N starts the ranks itself" in both.stderr
    bad = subprocess.run([CLI, "--rank", "2", "--world", "2", "--rendezvous", "x"], capture_output=True, text=True)
    assert bad.returncode == 255 and "--rank R --world N --rendezvous FILE" in bad.stderr
