"""Multi-rank invariance on real hardware: `pbsim --devices 0,0,..` runs one rank per listed device (here: several
contexts on the one GPU of the test box, host-barrier communicator) through the job pipeline of csrc/job.cpp.  Every
output file AND the stderr report must equal what the reference produced for the same command line (golden manifest,
keyed-Philox mode) -- N-GPU output == 1-GPU output == reference, byte for byte."""
import gzip
import os
import subprocess

import pytest

import harness
from cases import CASES

pytestmark = pytest.mark.gpu
MANIFEST = harness.load_manifest()
CLI = os.path.join(harness.ROOT, "pbsim3_amd", "bin", "pbsim")


def run_devices(args, workdir, ranks, scratch_mb=None, extra=("--no-gzip",), case=None, env=None):
    import pbsim3_amd.build as b
    b.build()
    e = dict(os.environ)
    if scratch_mb:
        e["PBSIM_SCRATCH_MB"] = str(scratch_mb)
    e.update(env or {})
    if case:
        harness.run_setup([CLI, "--no-gzip"], case, workdir)
    p = subprocess.run([CLI] + harness.resolve(args) + ["--prefix", os.path.join(workdir, "out"), "--devices",
                                                        ",".join(["0"] * ranks)] + list(extra),
                       capture_output=True, text=True, cwd=workdir, env=e)
    assert p.returncode == 0, p.stderr[-6000:]
    outs = harness.collect(workdir)
    outs[".stderr"] = harness.strip_report(p.stderr).encode()
    return outs


def check_against_golden(outs, case):
    want = MANIFEST[f"{case}/philox"]
    assert sorted(outs) == sorted(want), (sorted(outs), sorted(want))
    for k, v in outs.items():
        assert harness.sha(v) == want[k]["sha256"], (case, k, v[-600:] if k == ".stderr" else len(v))


WGS = sorted(c for c in CASES if c.startswith("wgs_") and "sample" not in c)


@pytest.mark.parametrize("case", WGS)
def test_two_ranks_every_wgs_golden(case, tmp_path):
    check_against_golden(run_devices(CASES[case]["args"], str(tmp_path), 2), case)


@pytest.mark.parametrize("case,ranks,scratch_mb", [
    ("wgs_errhmm-ont_quirk", 3, 3), ("wgs_errhmm-ont_quirk", 8, 4), ("wgs_errhmm-ont-hq_quirk", 4, 3),
    ("wgs_errhmm_rsii_default", 4, 24), ("wgs_qshmm_rsii_pass3", 5, 4), ("wgs_errhmm_ont_hpbias5", 2, 3),
    ("wgs_qshmm_onthq_pass2_hpbias2", 3, 4), ("wgs_errhmm_sequel_pass3", 8, 3)])
def test_many_ranks_many_rounds(case, ranks, scratch_mb, tmp_path):
    """a scratch pool of a few MB per slot forces tens of rounds per record: cuts inside any rank's block, top-up rounds,
    tails on any rank, records overlapping in the pipeline"""
    check_against_golden(run_devices(CASES[case]["args"], str(tmp_path), ranks, scratch_mb), case)


@pytest.mark.parametrize("case,ranks", [("trans_errhmm_sequel", 2), ("trans_errhmm_ont_hpbias4", 3), ("trans_qshmm_rsii", 4),
                                        ("trans_errhmm_rsii_acc98", 3), ("trans_errhmm_sequel_acc99_pass2", 2),
                                        ("templ_errhmm_sequel", 2), ("templ_errhmm_rsii_pass3_hpbias2", 4),
                                        ("templ_qshmm_rsii_pass2", 3), ("trans_qshmm_rsii_readme", 3),
                                        ("templ_qshmm_rsii_readme_pass10", 2), ("trans_errhmm_synthmod", 2)])
def test_unit_strategies(case, ranks, tmp_path):
    """trans / templ: rank r takes the r-th block of the unit set's read numbering (pbsim_simulate_units_range); statistics
    merged with pbsim_stats_merge; files written by byte range"""
    check_against_golden(run_devices(CASES[case]["args"], str(tmp_path), ranks), case)


@pytest.mark.parametrize("case,ranks,scratch_mb", [("wgs_errhmm-ont_quirk", 3, 3), ("wgs_qshmm_rsii_pass3", 2, 4),
                                                   ("trans_errhmm_sequel", 3, None), ("templ_qshmm_rsii_pass2", 2, None)])
def test_compressed_outputs_by_byte_range(case, ranks, scratch_mb, tmp_path):
    """default --gzip gpu: every rank compresses its blocks on the GPU and pwrite()s the members at the offsets the ranks
    agreed on; the files must be valid multi-member gzip / BGZF whose payload is the golden text"""
    import pbsim3_amd as P
    import test_gpu_bam
    work = tmp_path / "w"
    work.mkdir()
    outs = run_devices(CASES[case]["args"], str(work), ranks, scratch_mb, extra=())
    want = MANIFEST[f"{case}/philox"]
    gold = None
    seen = 0
    for n in sorted(os.listdir(work)):
        raw = open(work / n, "rb").read()
        if n.endswith((".fq.gz", ".maf.gz")):
            assert harness.sha(gzip.decompress(raw)) == want[n[len("out"):-3]]["sha256"], n
            seen += 1
        elif n.endswith(".bam"):
            assert raw.endswith(P.BGZF_EOF)
            if gold is None:
                (tmp_path / "g").mkdir()
                gold = harness.run_oracle(CASES[case]["args"], "philox", str(tmp_path / "g"))
            test_gpu_bam.compare_bam_with_sam(raw, gold[n[len("out"):-4] + ".sam"])
            seen += 1
    assert seen >= 2
    assert harness.sha(outs[".stderr"]) == want[".stderr"]["sha256"]


def test_failure_on_one_rank_stops_the_job(tmp_path):
    """a model file that does not exist: every rank fails alike, status 255, nothing hangs"""
    p = subprocess.run([CLI, "--strategy", "wgs", "--method", "errhmm", "--errhmm", "/nonexistent.model", "--genome",
                        os.path.join(harness.GOLDEN, "inputs", "quirk.fa"), "--prefix", str(tmp_path / "o"), "--devices", "0,0"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 255
    assert "ERROR" in p.stderr


@pytest.mark.parametrize("which", ["read", "maf"])
def test_sink_failure_on_either_thread_reaches_the_caller(which):
    """pbsim_set_deflate bit 2 serves the read sink from a second host thread; pbsim_last_error() is thread local, so a
    callback that fails THERE must still surface in the calling thread (untested)"""
    import ctypes as C
    import numpy as np
    import pbsim3_amd as P
    rng = np.random.default_rng(2)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 300_000)].tobytes()
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=1, depth=3.0)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_deflate(7)
        ctx.job_add_record(genome)
        bad = P.REC_TEXT_CB(lambda u, r, t, n, o: 0)
        good = P.REC_TEXT_CB(lambda u, r, t, n, o: 1)
        sink = P.RecordSink(None, bad if which == "read" else good, bad if which == "maf" else good, P.REC_DONE_CB())
        with pytest.raises(P.PbsimError, match="sink aborted \\(%s text\\)" % ("read" if which == "read" else "MAF")):
            P._check(ctx.lib.pbsim_job_run(ctx.h, None, C.byref(sink)))
        # the context stays usable: the same job runs through with a sink that accepts everything
        outs, done = ctx.job_run()
        assert done[1][0].res_num > 0 and len(outs[1][0]) == done[1][1] > 0
        # and the per-record driver with a sink that fails on its second thread
        ctx.set_reference(genome, 1)
        s2 = P.Sink(None, P.SINK_CB(lambda u, t, n: 0 if which == "read" else 1), P.SINK_CB(lambda u, t, n: 0 if which == "maf" else 1))
        with pytest.raises(P.PbsimError, match="sink aborted"):
            P._check(ctx.lib.pbsim_simulate_wgs(ctx.h, C.byref(s2)))
        rt, mt = ctx.simulate_wgs()
        assert len(rt) > 0 and len(mt) > 0


@pytest.mark.parametrize("devices,comm", [("0,0,0", "host"), ("0", "rccl")])
def test_communicators_selftest(devices, comm):
    """C1 / C2 / C3 once on each communicator of the `pbsim` binary: the host barrier with three contexts on the one GPU,
    and RCCL (librccl opened at run time, ncclCommInitAll + ncclAllGather / ncclAllReduce / ncclBroadcast on device memory)
    as a communicator of one -- all this box can offer; more ranks need distinct GPUs"""
    p = subprocess.run([CLI, "--devices", devices, "--comm", comm, "--comm-selftest"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "all-gather ok, all-reduce ok, device broadcast ok" in p.stderr


def test_genome_in_several_jobs(tmp_path):
    """a genome larger than the resident-reference budget runs as several jobs whose record numbering continues, and
    --hp-del-bias != 1 then takes its census over ALL records first (pbsim.cpp:677-696): one record per job here"""
    for case in ("wgs_errhmm_ont_hpbias5", "wgs_qshmm_rsii_pass3"):
        d = tmp_path / case
        d.mkdir()
        outs = run_devices(CASES[case]["args"], str(d), 2, scratch_mb=4, env={"PBSIM_JOB_REF_GB": "0.00001"})
        check_against_golden(outs, case)


@pytest.mark.parametrize("method,targets", [("errhmm", "2000000,500000,1000000"), ("qshmm", "700000,2500000,400000,1200000")])
def test_ranks_that_size_their_rounds_differently_still_agree(method, targets, tmp_path):
    """with the automatic scratch pool a rank sizes its rounds from its OWN free memory; ranks whose GPUs
    report different free bytes then derived different reads-per-round, and their blocks overlapped or left gaps.  The caps
    now go through one MIN over the ranks.  PBSIM_JOB_TARGET_RANKS injects a different batch target per rank (what
    differing hipMemGetInfo values do once the free memory bounds the batch) on a record whose rounds ARE bound by it
    (2 Mbp x depth 6, reads of ~1.2 kb: 10 k reads, caps of 400-1900 reads): the output still equals the oracle's, and
    every rank begins every round with the same block size -- the smallest rank's."""
    import numpy as np
    ranks = targets.count(",") + 1
    rng = np.random.default_rng(31)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 2_000_000)]
    fa = tmp_path / "g.fa"
    with open(fa, "wb") as f:
        f.write(b">chr1\n")
        lines = seq.reshape(-1, 80)
        f.write(np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1).tobytes())
    model = "ERRHMM-ONT.model" if method == "errhmm" else "QSHMM-RSII.model"
    args = ["--strategy", "wgs", "--method", method, "--" + method, harness.model_path(model), "--genome", str(fa),
            "--depth", "6", "--seed", "13", "--length-mean", "1200", "--length-sd", "900"]
    od = tmp_path / "o"
    od.mkdir()
    want = harness.run_oracle(args, "philox", str(od))
    import pbsim3_amd.build as b
    b.build()
    e = dict(os.environ, PBSIM_JOB_TARGET_RANKS=targets, PBSIM_TRACE="1")
    wd = tmp_path / "w"
    wd.mkdir()
    p = subprocess.run([CLI] + args + ["--prefix", str(wd / "out"), "--devices", ",".join(["0"] * ranks), "--no-gzip"],
                       capture_output=True, text=True, cwd=str(wd), env=e)
    assert p.returncode == 0, p.stderr[-4000:]
    begun = {}
    for line in p.stderr.splitlines():
        if line.startswith("[pbsim job r") and " begin rec " in line:
            r = int(line[len("[pbsim job r"):line.index("]")])
            begun.setdefault(r, []).append(line.split("] begin ", 1)[1])
    assert sorted(begun) == list(range(ranks)) and len(begun[0]) >= 3
    assert all(begun[r] == begun[0] for r in begun), begun
    n_per = int(begun[0][2].split("n_per=")[1])      # (the job's first two rounds are a fifth and a half of a full one: ramp-up)
    smallest = min(float(x) for x in targets.split(","))
    want_n = 1.08 * smallest / 1200.0 + 64                                      # the smallest rank's cap decides
    assert 0.85 * want_n < n_per < 1.15 * want_n, (n_per, want_n)                # (E[L] of the length table is ~1200)
    report = "\n".join(l for l in p.stderr.splitlines() if not l.startswith("[pbsim"))
    outs = harness.collect(str(wd))
    outs[".stderr"] = harness.strip_report(report + "\n").encode()
    assert sorted(outs) == sorted(want)
    for k in outs:
        assert outs[k] == want[k], k


def test_job_rerun_and_second_genome_on_one_context():
    """a context's homopolymer census and its Q15 state (an hp == 11 base has been counted) belong to
    ONE run of ONE genome.  Re-running a job gives the same bytes as its first run; a second genome on the same context takes
    its own census (--hp-del-bias != 1) and starts from a clean Q15 state -- each equals a fresh context's output."""
    import numpy as np
    import pbsim3_amd as P
    rng = np.random.default_rng(8)

    def rand(n):
        return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()

    a = rand(400_000)
    a[1000:1015] = ord("A")                      # a run of 15: hp == 11 bases (Q1 / Q15)
    a[5000:5009] = ord("C")
    b_ = rand(300_000)                           # no long runs, other homopolymer census
    for i in range(0, 300_000 - 8, 997):
        b_[i:i + 6] = ord("G")
    ga, gb = a.tobytes(), b_.tobytes()

    def fresh(genome, method, bias):
        p = P.default_params(strategy=P.STRATEGY_WGS, method=method, seed=4, depth=3.0, hp_del_bias=bias)
        with P.Context(p, 0) as ctx:
            (ctx.load_errhmm if method == P.METHOD_ERR else ctx.load_qshmm)(
                harness.model_path("ERRHMM-ONT.model" if method == P.METHOD_ERR else "QSHMM-RSII.model"))
            ctx.job_add_record(genome)
            texts, done = ctx.job_run()
            return bytes(texts[1][0]), bytes(texts[1][1])

    for method, bias in ((P.METHOD_QS, 1.0), (P.METHOD_QS, 3.0), (P.METHOD_ERR, 4.0)):
        want_a, want_b = fresh(ga, method, bias), fresh(gb, method, bias)
        p = P.default_params(strategy=P.STRATEGY_WGS, method=method, seed=4, depth=3.0, hp_del_bias=bias)
        with P.Context(p, 0) as ctx:
            (ctx.load_errhmm if method == P.METHOD_ERR else ctx.load_qshmm)(
                harness.model_path("ERRHMM-ONT.model" if method == P.METHOD_ERR else "QSHMM-RSII.model"))
            ctx.job_add_record(ga)
            for _ in range(2):                   # the same job twice
                texts, _ = ctx.job_run()
                assert (bytes(texts[1][0]), bytes(texts[1][1])) == want_a, (method, bias)
            ctx.job_begin(1)                     # a second genome on the same context
            ctx.job_add_record(gb)
            texts, _ = ctx.job_run()
            assert (bytes(texts[1][0]), bytes(texts[1][1])) == want_b, (method, bias)


SAMPLE = sorted(c for c in CASES if c.startswith("wgs_sample"))


@pytest.mark.parametrize("case", SAMPLE)
@pytest.mark.parametrize("ranks", [2, 3])
def test_sampling_method_sharded_by_string_blocks(case, ranks, tmp_path):
    """--method sample on several ranks (pbsim_simulate_sample_comm, pbsim.cpp:1694-1949): a round gives rank r the r-th run of
    a sweep's strings, the quota test at each read's start is placed by the same prefix gathers as the wgs quota rule.  Every
    file, the stored profile and the stderr report equal what the reference produced for the same command line."""
    outs = run_devices(CASES[case]["args"], str(tmp_path), ranks, case=CASES[case])
    check_against_golden(outs, case)


def test_sampling_method_sharded_compressed_and_larger(tmp_path):
    """a profile of 3 000 strings against a 3 Mbp record at depth 12 on 4 ranks, outputs compressed on the GPU: several sweeps,
    hundreds of strings per rank and round, the cut in the middle of a round; inflated files and report equal the oracle's."""
    import random
    import numpy as np
    r = random.Random(77)
    fq = tmp_path / "prof.fastq"
    with open(fq, "w") as f:
        for i in range(3000):
            n = min(30000, max(120, int(r.gammavariate(1.6, 2500))))
            q = "".join(chr(33 + min(60, max(2, int(r.gauss(14, 6))))) for _ in range(n))
            f.write("@p%d\n%s\n+\n%s\n" % (i, "A" * n, q))
    rng = np.random.default_rng(5)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 3_000_000)]
    fa = tmp_path / "g.fa"
    with open(fa, "wb") as f:
        f.write(b">chr1\n")
        lines = seq.reshape(-1, 60)
        f.write(np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1).tobytes())
    args = ["--strategy", "wgs", "--method", "sample", "--sample", str(fq), "--genome", str(fa), "--depth", "12", "--seed", "19"]
    od = tmp_path / "o"
    od.mkdir()
    want = harness.run_oracle(args, "philox", str(od))
    wd = tmp_path / "w"
    wd.mkdir()
    outs = run_devices(args, str(wd), 4, extra=())
    assert sorted(outs) == sorted(want)
    for k in outs:
        got = gzip.decompress(outs[k]) if k.endswith((".fq", ".maf")) else outs[k]
        assert got == want[k], k
