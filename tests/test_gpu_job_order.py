"""One rank, a sink that takes each stream FRONT TO BACK (what the CLI's --gzip host / samtools consumers do: a piece whose
offset is not the end of what has arrived fails the job): "on one GPU the offsets simply run up" (include/pbsim3_amd.h).

A record's truncated tail reads travel on a worker of their own (job.cpp tail_worker) while the record's last bulk round may
still be with the bulk worker; ADVICE r4 found that only timing kept the tail behind it.  The sink here is SLOW for bulk pieces
(the tail chain is collected and posted while the last bulk round is still in the callback), so the order holds only if the
tail really waits for its record's bulk bytes."""
import ctypes as C
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
