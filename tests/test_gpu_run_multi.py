"""pbsim3_amd.run_multi under torchrun (one process per rank, torch.distributed gloo rendezvous, all ranks on the test box's
one GPU): every rank runs pbsim_cli_main with a torch communicator; the files and the stderr report equal the goldens the
reference produced for the same command."""
import os
import subprocess
import sys

import pytest

import harness
from cases import CASES

pytestmark = pytest.mark.gpu
MANIFEST = harness.load_manifest()


@pytest.mark.parametrize("case,batch", [("wgs_errhmm-ont_quirk", "2"), ("wgs_qshmm_rsii_pass3", "3"),
                                        ("wgs_errhmm_ont_hpbias5", "256")])
def test_run_multi_two_ranks(case, batch, tmp_path):
    args = harness.resolve(CASES[case]["args"])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(30500 + os.getpid() % 1000), "-m", "pbsim3_amd.run_multi"] + args + \
          ["--prefix", str(tmp_path / "out"), "--backend", "gloo", "--one-gpu", "--scratch-mb", batch, "--no-gzip"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=harness.ROOT,
                       env=dict(os.environ, PYTHONPATH=harness.ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    outs = harness.collect(str(tmp_path))
    want = MANIFEST[f"{case}/philox"]
    keys = [k for k in want if k.endswith((".fq", ".maf", ".sam", ".ref"))]
    assert keys
    for k in keys:
        assert harness.sha(outs[k]) == want[k]["sha256"], k
    assert not [k for k in outs if ".rank" in k]
    err = p.stderr[p.stderr.index(":::: Simulation parameters"):]      # torchrun's own banner lines come first
    report = harness.strip_report("\n".join(l for l in err.splitlines() if "amdgpu.ids" not in l and not l.startswith(("W0", "W1", "[W", "[E"))))
    assert harness.sha(report.encode()) == want[".stderr"]["sha256"], report


@pytest.mark.parametrize("case,ranks", [("trans_errhmm_ont_hpbias4", 3), ("templ_qshmm_rsii_pass2", 3)])
def test_run_multi_unit_strategies(case, ranks, tmp_path):
    """trans / templ: every rank simulates one contiguous block of the unit set's reads (pbsim_simulate_units_range);
    the blocks stitched in rank order are the golden bytes"""
    args = harness.resolve(CASES[case]["args"])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(32500 + os.getpid() % 1000), "-m", "pbsim3_amd.run_multi"] + args + \
          ["--prefix", str(tmp_path / "out"), "--backend", "gloo", "--one-gpu", "--scratch-mb", "256", "--no-gzip"]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=harness.ROOT,
                       env=dict(os.environ, PYTHONPATH=harness.ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    outs = harness.collect(str(tmp_path))
    want = MANIFEST[f"{case}/philox"]
    keys = [k for k in want if k.endswith((".fq", ".maf", ".sam"))]
    assert keys
    for k in keys:
        assert harness.sha(outs[k]) == want[k]["sha256"], k
    assert not [k for k in outs if ".rank" in k]
    n_reads = outs[".fq"].count(b"\n") // 4 if ".fq" in outs else None
    if n_reads is not None:
        assert ("read num. : %d" % n_reads) in p.stderr


@pytest.mark.parametrize("case,batch", [("wgs_qshmm_rsii_pass3", "3")])
def test_run_multi_gzip_members_stitch(case, batch, tmp_path):
    """--gzip: every rank compresses on its GPU; the stitched members inflate to the golden bytes
    (.bam: a BGZF container whose payload starts with the BAM header and ends with the EOF marker)"""
    import gzip
    args = harness.resolve(CASES[case]["args"])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(31500 + os.getpid() % 1000), "-m", "pbsim3_amd.run_multi"] + args + \
          ["--prefix", str(tmp_path / "out"), "--backend", "gloo", "--one-gpu", "--scratch-mb", batch]
    p = subprocess.run(cmd, capture_output=True, text=True, cwd=harness.ROOT,
                       env=dict(os.environ, PYTHONPATH=harness.ROOT))
    assert p.returncode == 0, p.stderr[-3000:]
    want = MANIFEST[f"{case}/philox"]
    names = sorted(os.listdir(tmp_path))
    assert not [n for n in names if ".rank" in n]
    seen = 0
    for n in names:
        raw = open(tmp_path / n, "rb").read()
        if n.endswith((".fq.gz", ".maf.gz")):
            key = n[len("out"):-3]
            assert harness.sha(gzip.decompress(raw)) == want[key]["sha256"], n
            seen += 1
        elif n.endswith(".bam"):
            import pbsim3_amd as P
            assert raw.endswith(P.BGZF_EOF)
            body = gzip.decompress(raw)
            assert body[:4] == b"BAM\x01" and b"@HD\tVN:1.5" in body[:64]
            seen += 1
    assert seen >= 2
