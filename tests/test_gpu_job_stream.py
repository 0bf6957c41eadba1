"""pbsim_job_expect: a job whose records are announced and arrive WHILE pbsim_job_run is running, added by another thread
(the job starts before the last record is resident; main() reads its records one at a time,
pbsim.cpp:666-759).  The bytes and statistics must be those of the job whose records were all resident before it started."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

import harness

pytestmark = pytest.mark.gpu

G = [1_200_000, 700_000, 1_500_000, 400_000]


def records():
    return [harness.synth_bases(n, 60 + i).tobytes() for i, n in enumerate(G)]


def params(P, **kw):
    return P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=9, depth=6.0, **kw)


def resident_job(P, recs, **kw):
    with P.Context(params(P, **kw), 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_scratch_bytes(64 << 20)
        for r in recs:
            ctx.job_add_record(r)
        out, done = ctx.job_run()
        return {k: (bytes(v[0]), bytes(v[1])) for k, v in out.items()}, {k: ctx.format_stats(v[0], k) for k, v in done.items()}


@pytest.mark.parametrize("kw", [{}, {"hp_del_bias": 3.0}], ids=["default", "hp-del-bias 3 (census of all records first)"])
@pytest.mark.parametrize("delay", [0.0, 0.08])
def test_records_that_arrive_while_the_job_runs(kw, delay):
    import pbsim3_amd as P
    recs = records()
    want, want_rep = resident_job(P, recs, **kw)
    with P.Context(params(P, **kw), 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_scratch_bytes(64 << 20)
        ctx.job_expect([len(r) for r in recs])
        ctx.job_add_record(recs[0])
        errs = []

        def feed():
            try:
                for r in recs[1:]:
                    time.sleep(delay)
                    ctx.job_add_record(r)
            except Exception as e:      # noqa: BLE001
                errs.append(e)
                ctx.job_feed_abort(str(e))
        th = threading.Thread(target=feed)
        th.start()
        out, done = ctx.job_run()
        th.join()
        assert not errs, errs
        got = {k: (bytes(v[0]), bytes(v[1])) for k, v in out.items()}
        assert got == want
        assert {k: ctx.format_stats(v[0], k) for k, v in done.items()} == want_rep
        if delay and not kw:
            assert ctx.job_breakdown()["slot_wait"] > 0      # the loop did wait for a record (it is counted with the slots)
        # the same context again, every record resident by now: the same bytes
        out2, _ = ctx.job_run()
        assert {k: (bytes(v[0]), bytes(v[1])) for k, v in out2.items()} == want


def test_a_feed_that_gives_up_fails_the_job_and_a_wrong_record_is_refused():
    import pbsim3_amd as P
    recs = records()
    with P.Context(params(P), 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.job_expect([len(r) for r in recs])
        ctx.job_add_record(recs[0])
        th = threading.Thread(target=lambda: (time.sleep(0.05), ctx.job_feed_abort("the FASTA ended early")))
        th.start()
        with pytest.raises(P.PbsimError, match="FASTA ended early"):
            ctx.job_run()
        th.join()
        ctx.job_begin(1)
        ctx.job_expect([len(recs[0]), len(recs[1])])
        ctx.job_add_record(recs[0])
        with pytest.raises(P.PbsimError, match="announced"):
            ctx.job_add_record(recs[2])                      # another length than announced


def test_three_ranks_each_fed_while_the_job_runs():
    import pbsim3_amd as P
    from test_gpu_fullsize import thread_comms
    recs = records()
    want, want_rep = resident_job(P, recs)
    world = 3
    ctxs = []
    for r in range(world):
        ctx = P.Context(params(P), 0)
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_scratch_bytes(24 << 20)
        ctx.job_expect([len(x) for x in recs])
        ctx.job_add_record(recs[0])
        ctxs.append(ctx)
    comms = thread_comms(P, world)
    outs, errs = [None] * world, []

    def feed(r):
        for i, x in enumerate(recs[1:]):
            time.sleep(0.02 * (r + 1))                       # the ranks' records arrive at different times
            ctxs[r].job_add_record(x)

    def run(r):
        try:
            outs[r] = ctxs[r].job_run(comm=comms[r])
        except Exception as e:      # noqa: BLE001
            errs.append((r, e))
    th = [threading.Thread(target=f, args=(r,)) for r in range(world) for f in (feed, run)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for k in want:
        for w in (0, 1):
            text = np.zeros(len(want[k][w]), dtype=np.uint8)
            for r in range(world):                            # holes are zero (other ranks' ranges): the pieces overlay
                piece = np.frombuffer(bytes(outs[r][0][k][w]), dtype=np.uint8)
                assert len(piece) == len(text)
                assert not np.any((text != 0) & (piece != 0)), "two ranks delivered the same bytes"
                text |= piece
            text = text.tobytes()
            assert bytes(text) == want[k][w], (k, w)
    assert {k: ctxs[0].format_stats(v[0], k) for k, v in outs[0][1].items()} == want_rep
    for c in ctxs:
        c.close()
