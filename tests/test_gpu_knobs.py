"""The environment knobs that stay live in the product (INTEGRATION.md "Environment"): whatever they are set to, the bytes are
the reference's.  Each knob that no other test sets is exercised here on a golden case through the CLI -- several ranks on the
one GPU, tens of rounds per record (a scratch pool of a few MB), so that the path the knob switches is really taken.
("every knob is a code path the parity suite does not cover by default".)"""
import os
import subprocess

import pytest

import harness
from cases import CASES
from test_gpu_multi import CLI, check_against_golden, run_devices

pytestmark = pytest.mark.gpu

CASE = "wgs_errhmm-ont_quirk"


@pytest.mark.parametrize("env", [
    {"PBSIM_JOB_CLEAR": "0"},            # every round exchanges A, then B + C (the shape of a round that places the cut)
    {"PBSIM_JOB_CLEAR": "1"},            # every round emits its text first and exchanges once; the cut rounds do it again
    {"PBSIM_WALK_LDS_KB": "81"},         # one lane-walk workgroup per CU (the delivered job's occupancy)
    {"PBSIM_WALK_LDS_KB": "41"},
    {"PBSIM_JOB_FIT": "0.02"},           # a sliver of the free HBM for the rounds
    {"PBSIM_DEFLATE_TRACE": "1", "PBSIM_TRACE": "1"},
], ids=lambda e: " ".join("%s=%s" % kv for kv in e.items()))
@pytest.mark.parametrize("ranks", [1, 3])
def test_bytes_do_not_depend_on_the_knob(env, ranks, tmp_path):
    outs = run_devices(CASES[CASE]["args"], str(tmp_path), ranks, scratch_mb=3, env=env)
    if "PBSIM_TRACE" in env:             # the trace lines go to stderr beside the report: compare the files only
        outs.pop(".stderr")
        want = {k: v for k, v in harness.load_manifest()[f"{CASE}/philox"].items() if k != ".stderr"}
        assert sorted(outs) == sorted(want)
        for k, v in outs.items():
            assert harness.sha(v) == want[k]["sha256"], k
    else:
        check_against_golden(outs, CASE)


@pytest.mark.parametrize("ranks", [2, 4])
def test_compressed_ranks_with_a_small_pinned_arena(ranks, tmp_path):
    """several ranks compress their blocks into a pinned arena until their offsets are known: PBSIM_PINNED_ARENA_MB bounds it
    (blocks of 256 MB: 512 MB hold this case's rounds; 64 MB hold none -- every rank must then leave the job with the message)"""
    import gzip
    outs = run_devices(CASES[CASE]["args"], str(tmp_path), ranks, scratch_mb=3, extra=(), env={"PBSIM_PINNED_ARENA_MB": "512"})
    want = harness.load_manifest()[f"{CASE}/philox"]
    for k, v in outs.items():
        if k != ".stderr" and not k.endswith(".ref"):
            v = gzip.decompress(v)
        assert harness.sha(v) == want[k]["sha256"], k
    e = dict(os.environ, PBSIM_PINNED_ARENA_MB="64", PBSIM_SCRATCH_MB="3")
    d2 = tmp_path / "small"
    d2.mkdir()
    p = subprocess.run([CLI] + harness.resolve(CASES[CASE]["args"]) + ["--prefix", str(d2 / "out"), "--devices", ",".join(["0"] * ranks)],
                       capture_output=True, text=True, cwd=str(d2), env=e, timeout=120)
    assert p.returncode != 0 and "PBSIM_PINNED_ARENA_MB" in p.stderr       # refused on every rank, nobody left waiting


def run_cli(args, workdir, env=None, extra=("--no-gzip",)):
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.run([CLI] + harness.resolve(args) + ["--prefix", os.path.join(workdir, "out")] + list(extra),
                       capture_output=True, text=True, cwd=workdir, env=e)
    assert p.returncode == 0, p.stderr[-4000:]
    outs = harness.collect(workdir)
    outs[".stderr"] = harness.strip_report(p.stderr).encode()
    return outs


@pytest.mark.parametrize("env", [{"PBSIM_FASTA_LOADER": "stdio"}, {"PBSIM_CLI_LEAVE_CONTEXT": "0"}, {"PBSIM_CLI_LEAVE_CONTEXT": "1"},
                                 {"PBSIM_NUMA_BIND": "0"}],
                         ids=lambda e: " ".join("%s=%s" % kv for kv in e.items()))
def test_single_rank_cli_knobs(env, tmp_path):
    """the `pbsim` binary itself (one rank, no --devices): the fgets loader instead of the mapped one; an ordinary exit through
    pbsim_destroy instead of _exit (what a profiler needs) -- same files, same report, exit status 0"""
    check_against_golden(run_cli(CASES[CASE]["args"], str(tmp_path), env), CASE)
