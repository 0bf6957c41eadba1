"""One rank, a sink that takes each stream FRONT TO BACK (what the CLI's --gzip host / samtools consumers do: a piece whose
offset is not the end of what has arrived fails the job): "on one GPU the offsets simply run up" (include/pbsim3_amd.h).

A record's truncated tail reads travel on a worker of their own (job.cpp tail_worker) while the record's last bulk round may
still be with the bulk worker; ADVICE r4 found that only timing kept the tail behind it.  The sink here is SLOW for bulk pieces
(the tail chain is collected and posted while the last bulk round is still in the callback), so the order holds only if the
tail really waits for its record's bulk bytes."""
import ctypes as C
import os
import time

import pytest

import harness

pytestmark = pytest.mark.gpu


class SequentialSink:
    def __init__(self, P, delay_s):
        self.expect = {}
        self.chunks = {}
        self.violations = []
        self.delay_s = delay_s

        def put(which, rec, text, n, off):
            key = (rec, which)
            if off != self.expect.get(key, 0):
                self.violations.append((rec, which, off, self.expect.get(key, 0)))
                return 0                      # what cli.cpp's Stream::write does: the job fails with "sink aborted"
            if n > 200_000:                   # a bulk piece: be slow (the tail's few KB are not)
                time.sleep(self.delay_s)
            self.chunks.setdefault(key, []).append(C.string_at(text, n))
            self.expect[key] = off + n
            return 1

        self._cbs = (P.REC_TEXT_CB(lambda u, r, t, n, o: put(0, r, t, n, o)), P.REC_TEXT_CB(lambda u, r, t, n, o: put(1, r, t, n, o)),
                     P.REC_DONE_CB(lambda u, rec, st, rb, mb: 1))
        self.sink = P.RecordSink(None, *self._cbs)


@pytest.mark.parametrize("deflate", [0, 7])
@pytest.mark.parametrize("interleave", [1, 3])
def test_tail_bytes_follow_the_bulk_bytes_of_their_record(deflate, interleave, tmp_path):
    import gzip
    import pbsim3_amd as P
    G = 2_000_000
    recs = [harness.synth_bases(G, 40 + i).tobytes() for i in range(3)]
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=3, depth=8.0)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_scratch_bytes(96 << 20)          # a few rounds per record
        ctx.set_deflate(deflate)
        P._check(ctx.lib.pbsim_job_set_interleave(ctx.h, interleave))
        for r in recs:
            ctx.job_add_record(r)
        want, done = ctx.job_run()               # positional sink: the record's streams as they should be
        slow = SequentialSink(P, 0.05)
        ok = ctx.lib.pbsim_job_run(ctx.h, None, C.byref(slow.sink))
        assert slow.violations == [], slow.violations
        P._check(ok)
        assert ctx.job_breakdown()["tail_reads"] >= 1        # (a record of this size ends in a truncated read)
        for rec in want:
            for which in (0, 1):
                got = b"".join(slow.chunks[(rec, which)])
                assert got == bytes(want[rec][which]), (rec, which)
                if deflate:
                    assert len(gzip.decompress(got)) > G


def _slow_stub_dir(tmp_path):
    """`samtools view -b -o FILE -` and `gzip` replaced by consumers that read their stdin SLOWLY into the target (plain bytes):
    the sequential pipes of the reference's own output path (pbsim.cpp:708-730), with back pressure"""
    d = tmp_path / "stubs"
    d.mkdir()
    slow = ("import sys,time\n"
            "out=open(sys.argv[1],'wb')\n"
            "while True:\n"
            "    b=sys.stdin.buffer.read(1<<16)\n"
            "    if not b: break\n"
            "    out.write(b); time.sleep(0.002)\n")
    (d / "slow.py").write_text(slow)
    (d / "samtools").write_text('#!/bin/sh\nexec python3 "%s/slow.py" "$4"\n' % d)
    os.chmod(d / "samtools", 0o755)
    return str(d)


@pytest.mark.parametrize("case", ["wgs_qshmm_rsii_pass3", "wgs_errhmm-ont_quirk"])
def test_cli_sequential_consumers_with_back_pressure(case, tmp_path):
    """The CLI's sequential consumers -- `--samtools` (a pipe) and `--gzip host` (zlib on host threads) -- on ONE rank through the job
    pipeline with a small scratch pool (several rounds per record + truncated tails): a piece that does not start where the stream
    ends fails the job ("sink aborted", cli.cpp Stream::write).  ADVICE r4: only timing kept a record's tail behind its last bulk
    round; here the consumer of the read stream is slow, so the bulk pieces stay in the callback while the tail is ready."""
    import gzip
    import subprocess
    from cases import CASES
    CLI = os.path.join(harness.ROOT, "pbsim3_amd", "bin", "pbsim")
    env = dict(os.environ, PATH=_slow_stub_dir(tmp_path) + ":" + os.environ["PATH"], PBSIM_SCRATCH_MB="3")
    out = tmp_path / "o"
    out.mkdir()
    p = subprocess.run([CLI] + harness.resolve(CASES[case]["args"]) + ["--prefix", str(out / "out"), "--samtools", "--gzip", "host"],
                       capture_output=True, text=True, cwd=str(out), env=env, timeout=240)
    assert p.returncode == 0, p.stderr[-3000:]
    got = harness.collect(str(out))
    want = harness.load_manifest()[f"{case}/philox"]
    for k, v in got.items():
        if k.endswith((".fq", ".maf")):
            v = gzip.decompress(v)               # --gzip host: real gzip files
        assert harness.sha(v) == want[k]["sha256"], k
    assert harness.sha(harness.strip_report(p.stderr).encode()) == want[".stderr"]["sha256"]
