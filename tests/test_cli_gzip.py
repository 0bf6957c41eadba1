"""CPU test of the CLI's multi-threaded gzip writer (pbsim3_amd/csrc/gzout.cpp): the
output must be a standard (multi-member) .gz whose decompressed bytes are the input."""
import gzip
import os
import subprocess

import numpy as np
import pytest

import harness

CLI = os.path.join(harness.ROOT, "pbsim3_amd", "bin", "pbsim")


@pytest.mark.parametrize("size,threads", [(0, 1), (1, 2), (1 << 20, 3), ((5 << 20) + 12345, 8)])
def test_parallel_gzip_roundtrip(size, threads, tmp_path):
    import pbsim3_amd.build as b
    b.build()
    rng = np.random.default_rng(size + threads)
    data = np.frombuffer(b"ACGT-!\n", dtype=np.uint8)[rng.integers(0, 7, size)].tobytes()
    src = tmp_path / "x.txt"
    src.write_bytes(data)
    p = subprocess.run([CLI, "--gzip-threads", str(threads), "--gzip-file", str(src)], capture_output=True)
    assert p.returncode == 0, p.stderr
    with gzip.open(str(src) + ".gz", "rb") as f:
        assert f.read() == data
    if size:  # the system gzip agrees
        out = subprocess.run(["gzip", "-dc", str(src) + ".gz"], capture_output=True, check=True).stdout
        assert out == data
