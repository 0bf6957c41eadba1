"""Host-side C++ of the product (FASTA / transcript / template / sample-FASTQ readers, model parsers, integer table
builders) under AddressSanitizer + UBSan on the CPU.  GPU sanitizers are not available on the pool; these are the
parts of the library that touch untrusted input files."""
import os
import shutil
import subprocess

import pytest

import harness

CSRC = os.path.join(harness.ROOT, "pbsim3_amd", "csrc")
HERE = os.path.join(harness.ROOT, "tests", "asan")
FLAGS = ["-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
         "-I" + CSRC, "-I" + os.path.join(harness.ROOT, "include")]
MODELS = ["ERRHMM-RSII.model", "ERRHMM-SEQUEL.model", "ERRHMM-ONT.model", "ERRHMM-ONT-HQ.model", "QSHMM-RSII.model",
          "QSHMM-ONT.model"]


def build(tmp_path, driver, source):
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = str(tmp_path / driver)
    p = subprocess.run(["g++"] + FLAGS + [os.path.join(HERE, driver + ".cpp"), os.path.join(CSRC, source), "-o", exe],
                       capture_output=True, text=True)
    if p.returncode != 0 and "sanitize" in p.stderr:
        pytest.skip("no sanitizer runtime")
    assert p.returncode == 0, p.stderr[-2000:]
    return exe


def test_parsers_under_asan(tmp_path):
    exe = build(tmp_path, "parsers_driver", "unit_io.cpp")
    i = os.path.join(harness.GOLDEN, "inputs")
    p = subprocess.run([exe, i + "/sample.fastq", i + "/quirk.fa", i + "/tiny.transcript", i + "/tiny.template", str(tmp_path)],
                       capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "num 153 filt 124 tot 133973" in p.stdout and "same=1" in p.stdout
    assert "rec 1 len 17104" in p.stdout and "tr 12 exp 45" in p.stdout


def test_table_builders_under_asan(tmp_path):
    exe = build(tmp_path, "tables_driver", "host_tables.cpp")
    p = subprocess.run([exe] + [harness.model_path(m) for m in MODELS], capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert p.stdout.count(" ok stride ") == len(MODELS)
    assert "emission_magic bad 0" in p.stdout


def test_comm_abort_releases_waiting_ranks(tmp_path):
    """pbsim_comm.abort of the in-process communicator (csrc/thread_comm.h): a rank that leaves the job between two
    exchanges releases the ranks waiting in the next collective -- they fail instead of hanging (job.cpp)."""
    if not shutil.which("g++"):
        pytest.skip("g++ not available")
    exe = str(tmp_path / "comm_abort")
    cmd = ["g++"] + FLAGS + ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(HERE, "comm_abort_driver.cpp"),
                             "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-o", exe]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0 and "sanitize" in p.stderr:
        pytest.skip("no sanitizer runtime")
    assert p.returncode == 0, p.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    assert "first 8 second_failed 3 third_failed 3" in p.stdout
