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
FLAGS = ["-std=c++17", "-g", "-O1", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
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


def test_mapped_sample_parse_equals_the_fgets_parse(tmp_path):
    """read_sample_fastq (mmap, memchr, per-string sums on threads) against read_sample_fastq_stdio (the reference's fgets loop,
    BUF_SIZE chunks) under ASan: every statistic bit for bit and every kept string, on files with lines longer than BUF_SIZE,
    CR LF, empty lines, a missing final line feed, a truncated record, characters outside '!'..'~', a NUL byte (falls back),
    nothing in range (the same error), and 6 000 records (the threaded path)."""
    import random
    exe = build(tmp_path, "parsers_driver", "unit_io.cpp")
    r = random.Random(7)

    def rec(i, n, lo=5, hi=40, eol="\n"):
        q = "".join(chr(33 + r.randint(lo, hi)) for _ in range(n))
        return "@r%d%s%s%s+%s%s%s" % (i, eol, "A" * n, eol, eol, q, eol)

    files = {}
    files["plain"] = "".join(rec(i, r.randint(20, 3000)) for i in range(200))
    files["long_lines"] = "".join(rec(i, r.choice([10239, 10240, 10241, 20479, 20480, 31000, 150])) for i in range(30))
    files["crlf"] = "".join(rec(i, r.randint(100, 900), eol="\r\n") for i in range(50))
    files["empty_lines"] = rec(0, 300) + "\n\n\n\n" + rec(1, 400) + "\n" + rec(2, 500) + rec(3, 200)
    files["no_final_lf"] = "".join(rec(i, 250) for i in range(5))[:-1]
    files["truncated"] = "".join(rec(i, 250) for i in range(5)) + "@last\nACGT\n+\n"
    files["odd_bytes"] = rec(0, 300, 0, 0) + "@x\nAC\n+\n" + "\x7f\x80\xff \t" * 60 + "\n" + rec(1, 300, 60, 93)
    files["with_nul"] = rec(0, 300) + "@n\nAC\n+\n" + "I" * 150 + "\0" + "I" * 150 + "\n" + rec(1, 300)
    files["nothing_in_range"] = "".join(rec(i, 50, 0, 3) for i in range(10))
    files["many"] = "".join(rec(i, r.randint(100, 1500)) for i in range(6000))
    paths = []
    for name, text in files.items():
        p = tmp_path / (name + ".fastq")
        p.write_bytes(text.encode("latin-1"))
        paths.append(str(p))
    p = subprocess.run([exe, "--cmp"] + paths, capture_output=True, text=True)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert p.stdout.count("same=1") == 2 * len(paths) and "same=0" not in p.stdout
    assert "many.fastq ok=1 kept=6000" in p.stdout


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


def test_mapped_fasta_pass_equals_the_fgets_pass(tmp_path):
    """map_genome + write_ref_record (the file mapped, headers by memchr on threads, line feeds counted per record; the CLI's
    loader since round 4) against split_genome + load_ref_record (the reference's fgets loop, pbsim.cpp:896-991, 997-1033)
    under ASan: record count, lengths, ids, the error text, every .ref file byte for byte, and the record itself.  Files with
    lines at and around BUF_SIZE - 1 = 10239 characters (the fgets chunking shows in .ref), a '>' where a chunk of a long
    line starts (a header to the reference), headers longer than a chunk and than the 128-character id, CR LF, empty lines,
    no final line feed, an empty record, a short record, sequence in front of the first header and a NUL byte (both fall
    back to the fgets pass)."""
    import random
    exe = build(tmp_path, "parsers_driver", "unit_io.cpp")
    r = random.Random(7)

    def seq(n):
        return "".join(r.choice("ACGT") for _ in range(n))

    def lines(s, w):
        return "".join(s[i:i + w] + "\n" for i in range(0, len(s), w))
    files = {
        "plain.fa": ">chr1 desc\n" + lines(seq(5000), 70) + ">chr2\n" + lines(seq(333), 80),
        "long_lines.fa": ">a\n" + seq(10238) + "\n" + seq(10239) + "\n" + seq(10240) + "\n" + seq(20478) + "\n" + seq(30) + "\n>b\n" + seq(25000) + "\n",
        "gt_at_chunk.fa": ">a\n" + seq(10239) + ">" + seq(400) + "\n" + seq(150) + "\n" + seq(10238) + ">" + seq(200) + "\n",
        "long_header.fa": ">" + "h" * 300 + "\n" + lines(seq(400), 60) + ">" + "k" * 10239 + ">tail of the header " + "z" * 11000 + "\n" + lines(seq(500), 60),
        "crlf.fa": ">a\r\n" + "".join(seq(60) + "\r\n" for _ in range(10)) + ">b\r\n" + seq(200) + "\r\n",
        "empty_lines.fa": ">a\n\n" + lines(seq(300), 50) + "\n\n>b\n" + seq(120) + "\n\n",
        "no_final_lf.fa": ">a\n" + lines(seq(400), 80) + seq(77),
        "no_final_lf_long.fa": ">a\n" + seq(10239 * 2),
        "empty_record.fa": ">a\n>b\n" + lines(seq(400), 80),
        "short_record.fa": ">a\n" + lines(seq(400), 80) + ">b\n" + seq(99) + "\n>c\n" + lines(seq(400), 80),
        "short_last.fa": ">a\n" + lines(seq(400), 80) + ">b\n" + seq(50) + "\n",
        "leading_sequence.fa": seq(50) + "\n>a\n" + lines(seq(400), 80),
        "nul_byte.fa": ">a\n" + seq(200) + "\x00" + seq(100) + "\n",
        "header_only_at_end.fa": ">a\n" + lines(seq(400), 80) + ">b",
        "lower_and_iupac.fa": ">a\n" + lines((seq(300) + "nnnnacgtRYKM" + seq(100)).lower(), 61),
    }
    paths = []
    for name, text in files.items():
        (tmp_path / name).write_bytes(text.encode("latin-1"))
        paths.append(str(tmp_path / name))
    big = ">big\n" + lines(seq(3_000_00) * 12, 80) + ">big2 second\n" + lines(seq(100_000) * 5, 97)   # several 16 MiB-ish blocks? no: block logic by piece seams
    (tmp_path / "big.fa").write_bytes(big.encode())
    paths.append(str(tmp_path / "big.fa"))
    out = tmp_path / "o"
    out.mkdir()
    for block in (None, "997", "10239", "64"):     # thread blocks of the scan: the default 16 MiB, and seams inside every line
        env = dict(os.environ) if block is None else dict(os.environ, PBSIM_FASTA_BLOCK=block)
        p = subprocess.run([exe, "--cmpfa", str(out)] + paths, capture_output=True, text=True, env=env)
        assert p.returncode == 0, (block, (p.stdout + p.stderr)[-4000:])
        assert p.stdout.count("same=1") == len(paths) - 2 and p.stdout.count("fallback") == 2, (block, p.stdout)
        assert "Reference is too short" in p.stdout
