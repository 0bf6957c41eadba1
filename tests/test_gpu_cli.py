"""The `pbsim` CLI shim (pbsim3_amd/bin/pbsim, C++ over the C ABI) against the
golden vectors the reference produced in keyed-Philox mode: every output file
(.ref, FASTQ or SAM text, MAF) and the stderr report, byte for byte."""
import os
import subprocess

import pytest

import harness
from cases import CASES

MANIFEST = harness.load_manifest()
CLI = os.path.join(harness.ROOT, "pbsim3_amd", "bin", "pbsim")

pytestmark = pytest.mark.gpu


def run_cli(args, workdir, case=None):
    import pbsim3_amd.build as b
    b.build()
    harness.run_setup([CLI, "--no-gzip"], case, workdir)
    p = subprocess.run([CLI] + harness.resolve(args) + ["--prefix", os.path.join(workdir, "out"), "--no-gzip"],
                       capture_output=True, text=True, cwd=workdir)
    assert p.returncode == 0, p.stderr[-2000:]
    outs = harness.collect(workdir)
    outs[".stderr"] = harness.strip_report(p.stderr).encode()
    return outs


@pytest.mark.parametrize("case", sorted(CASES))
def test_cli_matches_reference_golden(case, tmp_path):
    outs = run_cli(CASES[case]["args"], str(tmp_path), CASES[case])
    want = MANIFEST[f"{case}/philox"]
    assert sorted(outs) == sorted(want), (sorted(outs), sorted(want))
    for k, v in outs.items():
        if harness.sha(v) != want[k]["sha256"]:
            ref = harness.run_oracle(CASES[case]["args"], "philox", str(tmp_path / "o"), case=CASES[case])[k] if (tmp_path / "o").mkdir() is None else b""
            n = next((i for i, (x, y) in enumerate(zip(v, ref)) if x != y), min(len(v), len(ref)))
            raise AssertionError(f"{case}{k}: differs at byte {n} (sizes {len(v)} vs {len(ref)}):\n"
                                 f"  got  {v[max(0, n - 80):n + 60]!r}\n  want {ref[max(0, n - 80):n + 60]!r}")


def test_cli_error_convention(tmp_path):
    """ERROR: message on stderr and exit status 255 like the reference's exit(-1)."""
    p = subprocess.run([CLI, "--strategy", "wgs", "--method", "errhmm", "--genome", "/nonexistent",
                        "--errhmm", "/nonexistent.model"], capture_output=True, text=True)
    assert p.returncode == 255
    assert "ERROR: Cannot open file" in p.stderr


def test_cli_default_outputs_are_gzip(tmp_path):
    """Without --no-gzip the CLI writes <prefix>_NNNN.fq.gz / .maf.gz itself (multi-member gzip)."""
    import gzip
    case = "wgs_errhmm-ont_quirk"
    p = subprocess.run([CLI] + harness.resolve(CASES[case]["args"]) + ["--prefix", str(tmp_path / "out"), "--gzip-threads", "3"],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    want = MANIFEST[f"{case}/philox"]
    for k in ("_0001.fq", "_0002.fq", "_0001.maf", "_0002.maf"):
        with gzip.open(str(tmp_path / ("out" + k + ".gz")), "rb") as f:
            assert harness.sha(f.read()) == want[k]["sha256"], k


@pytest.mark.parametrize("mode", [["--gzip", "host", "--gzip-threads", "2"], []])
def test_cli_sample_method_compressed(tmp_path, mode):
    """--method sample through the default (GPU) and the host compressors: the members inflate to the golden bytes"""
    import gzip
    case = "wgs_sample_quirk"
    p = subprocess.run([CLI] + harness.resolve(CASES[case]["args"]) + ["--prefix", str(tmp_path / "out")] + mode,
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    want = MANIFEST[f"{case}/philox"]
    for k in ("_0001.fq", "_0002.fq", "_0001.maf", "_0002.maf"):
        with gzip.open(str(tmp_path / ("out" + k + ".gz")), "rb") as f:
            assert harness.sha(f.read()) == want[k]["sha256"], k
    assert harness.sha(harness.strip_report(p.stderr).encode()) == want[".stderr"]["sha256"]


def test_cli_transcripts_without_expression(tmp_path):
    """plus = minus = 0 everywhere: the reference simulates nothing and still prints its report (NaN means)"""
    tsv = tmp_path / "t.tsv"
    tsv.write_text("T0\t0\t0\t" + "ACGT" * 40 + "\nT1\t0\t0\t" + "GGCATTA" * 30 + "\n")
    args = ["--strategy", "trans", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-SEQUEL.model", "--transcript", str(tsv),
            "--seed", "1"]
    (tmp_path / "o").mkdir()
    (tmp_path / "p").mkdir()
    want = harness.run_oracle(args, "philox", str(tmp_path / "o"))
    outs = run_cli(args, str(tmp_path / "p"))
    assert sorted(outs) == sorted(want)
    for k in want:
        assert outs[k] == want[k], k
    assert b"nan" in outs[".stderr"] and outs[".fq"] == b""


@pytest.mark.parametrize("case", ["wgs_qshmm_rsii_pass3", "trans_errhmm_sequel", "templ_qshmm_rsii_pass2", "wgs_errhmm_ont_hpbias5"])
@pytest.mark.parametrize("one_writer", ["0", "1"])
def test_cli_gpu_members_from_one_and_two_writer_threads(tmp_path, case, one_writer):
    """Default CLI mode (members compressed on the GPU): with two writer threads (pbsim_set_deflate bit 2) and with one,
    .fq.gz / .maf.gz inflate to the golden text and the .bam is a BGZF container that ends with the EOF marker."""
    import gzip
    import pbsim3_amd as P
    p = subprocess.run([CLI] + harness.resolve(CASES[case]["args"]) + ["--prefix", str(tmp_path / "out")], capture_output=True,
                       text=True, env=dict(os.environ, PBSIM_CLI_ONE_WRITER=one_writer))
    assert p.returncode == 0, p.stderr[-2000:]
    want = MANIFEST[f"{case}/philox"]
    seen = 0
    for n in sorted(os.listdir(tmp_path)):
        raw = open(tmp_path / n, "rb").read()
        if n.endswith((".fq.gz", ".maf.gz")):
            assert harness.sha(gzip.decompress(raw)) == want[n[len("out"):-3]]["sha256"], n
            seen += 1
        elif n.endswith(".bam"):
            assert raw.endswith(P.BGZF_EOF) and gzip.decompress(raw)[:4] == b"BAM\x01", n
            seen += 1
    assert seen >= 2
