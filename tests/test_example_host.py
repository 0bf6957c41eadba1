"""examples/pbsim_min.c: a C99 host of the ABI.  CPU: it compiles and links against the library with -Wall -Wextra
(the header is plain C).  GPU: its FASTQ / MAF equal the oracle's for the same command."""
import os
import shutil
import subprocess

import pytest

import harness

SRC = os.path.join(harness.ROOT, "examples", "pbsim_min.c")
LIBDIR = os.path.join(harness.ROOT, "pbsim3_amd", "lib")


def build(tmp_path):
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    import pbsim3_amd.build as b
    b.build()
    exe = str(tmp_path / "pbsim_min")
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(harness.ROOT, "include"), SRC,
                        "-L", LIBDIR, "-lpbsim3_amd", "-Wl,-rpath," + LIBDIR, "-o", exe], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


def test_example_compiles_as_c99(tmp_path):
    build(tmp_path)


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
