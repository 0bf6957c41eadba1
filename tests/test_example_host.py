"""examples/pbsim_min.c, examples/pbsim_ranks.c and examples/pbsim_rccl_rank.c: C99 hosts of the ABI (one record at a time on one
GPU; the whole-genome job on several ranks with a pthread-barrier pbsim_comm and pwrite() sinks; one rank of a
one-process-per-GPU job on the library's own RCCL communicator, pbsim_rccl_comm_create_file).  CPU: they compile and link against the library
with -Wall -Wextra (the header is plain C).  GPU: their FASTQ / MAF equal the oracle's for the same command."""
import os
import shutil
import subprocess

import pytest

import harness

SRC = os.path.join(harness.ROOT, "examples", "pbsim_min.c")
SRC_RANKS = os.path.join(harness.ROOT, "examples", "pbsim_ranks.c")
SRC_RCCL = os.path.join(harness.ROOT, "examples", "pbsim_rccl_rank.c")
LIBDIR = os.path.join(harness.ROOT, "pbsim3_amd", "lib")


def build(tmp_path, src=SRC):
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    import pbsim3_amd.build as b
    b.build()
    exe = str(tmp_path / os.path.basename(src)[:-2])
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(harness.ROOT, "include"), src,
                        "-L", LIBDIR, "-lpbsim3_amd", "-lpthread", "-Wl,-rpath," + LIBDIR, "-o", exe],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


@pytest.mark.parametrize("src", [SRC, SRC_RANKS, SRC_RCCL])
def test_example_compiles_as_c99(src, tmp_path):
    build(tmp_path, src)


@pytest.mark.gpu
def test_example_matches_oracle(tmp_path):
    exe = build(tmp_path)
    args = ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model", "--genome", "INPUT:quirk.fa",
            "--depth", "3", "--seed", "9"]
    want = harness.run_oracle(args, "philox", str(tmp_path))
    r = harness.resolve(args)
    p = subprocess.run([exe, r[r.index("--errhmm") + 1], r[r.index("--genome") + 1], "3", "9"], capture_output=True)
    assert p.returncode == 0, p.stderr[-2000:]
    assert p.stdout == want["_0001.fq"] + want["_0002.fq"]
    assert p.stderr == want["_0001.maf"] + want["_0002.maf"]


@pytest.mark.gpu
@pytest.mark.parametrize("ranks", [1, 3])
def test_ranks_example_matches_oracle(ranks, tmp_path):
    """the job API with a caller-written communicator: N threads' pwrite()s assemble the oracle's files"""
    exe = build(tmp_path, SRC_RANKS)
    args = ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model", "--genome", "INPUT:quirk.fa",
            "--depth", "3", "--seed", "9"]
    (tmp_path / "o").mkdir()
    want = harness.run_oracle(args, "philox", str(tmp_path / "o"))
    r = harness.resolve(args)
    p = subprocess.run([exe, r[r.index("--errhmm") + 1], r[r.index("--genome") + 1], "3", "9", str(ranks),
                        str(tmp_path / "out")], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    for k in ("_0001.fq", "_0002.fq", "_0001.maf", "_0002.maf"):
        assert (tmp_path / ("out" + k)).read_bytes() == want[k], k
    assert p.stdout.count(b"record ") == 2


@pytest.mark.gpu
def test_rccl_rank_example_matches_oracle(tmp_path):
    """one process per GPU in C99 on the library's RCCL communicator (ncclCommInitRank through a rendezvous file), as a world of
    one -- all this box offers: RCCL takes one rank per GPU"""
    exe = build(tmp_path, SRC_RCCL)
    args = ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model", "--genome", "INPUT:quirk.fa",
            "--depth", "3", "--seed", "9"]
    (tmp_path / "o").mkdir()
    want = harness.run_oracle(args, "philox", str(tmp_path / "o"))
    r = harness.resolve(args)
    p = subprocess.run([exe, r[r.index("--errhmm") + 1], r[r.index("--genome") + 1], "3", "9", "0", "1", str(tmp_path / "rdv"),
                        str(tmp_path / "out")], capture_output=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    for k in ("_0001.fq", "_0001.maf", "_0002.fq", "_0002.maf"):
        assert open(str(tmp_path / "out") + k, "rb").read() == want[k], k
    assert b"record 1:" in p.stdout and b"record 2:" in p.stdout and not os.path.exists(tmp_path / "rdv")
