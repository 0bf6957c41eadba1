"""tools/replay_ranks.py: a rank of an N-rank job run ALONE against "virtual ranks" -- a communicator that answers gathers A
and B for the other ranks from a table of per-read (raw length, pass-0 bases) and the reference's quota rule
(pbsim.cpp:3792-3800).  The harness checks the rank's own values against the table at every exchange; here its output is
checked too: the text pieces the N replayed ranks deliver tile the one-GPU job's text exactly."""
import os
import sys

import numpy as np
import pytest

import harness

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(harness.ROOT, "tools"))


def genome(n, seed):
    return harness.synth_bases(n, seed).tobytes()


@pytest.mark.parametrize("method,world,target", [("errhmm", 3, 250_000), ("errhmm", 8, 120_000), ("qshmm", 4, 400_000)])
def test_replayed_ranks_tile_the_one_gpu_output(method, world, target, monkeypatch):
    import ctypes as C
    import torch
    import pbsim3_amd as P
    import replay_ranks as RR
    qs = method == "qshmm"
    model = "QSHMM-RSII.model" if qs else "ERRHMM-ONT.model"
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_QS if qs else P.METHOD_ERR, seed=5, depth=6.0,
                         len_mean=1500.0, len_sd=1100.0, pass_num=2 if qs else 1)
    G = 1_500_000
    # rounds of `target` bases per rank (the test hook that sizes the rounds; a pool too small for a block would make the job
    # retry with halved caps -- every rank's business, which virtual ranks do not model and refuse)
    monkeypatch.setenv("PBSIM_JOB_TARGET_RANKS", str(target))
    recs = [torch.frombuffer(bytearray(genome(G, 10 + i)), dtype=torch.uint8).cuda() for i in range(2)]
    tables = RR.build_tables(P, harness, p, model, qs, recs, G, 0)
    with P.Context(p, 0) as ctx:
        ctx.set_scratch_bytes(256 << 20)
        (ctx.load_qshmm if qs else ctx.load_errhmm)(harness.model_path(model))
        for t in recs:
            ctx.job_add_record_device(t.data_ptr(), G)
        want, done = ctx.job_run()
        want = {k: (bytes(v[0]), bytes(v[1])) for k, v in want.items()}
        pieces = {k: ([], []) for k in want}
        checked = rounds = 0
        for r in range(world):
            got = {k: ([], []) for k in want}

            def put(which, rec, text, n, off, got=got):
                got[rec][which].append(C.string_at(text, n))
                return 1
            cbs = (P.REC_TEXT_CB(lambda u, rec, t, n, o: put(0, rec, t, n, o)), P.REC_TEXT_CB(lambda u, rec, t, n, o: put(1, rec, t, n, o)),
                   P.REC_DONE_CB(lambda u, rec, st, rb, mb: 1))
            sink = P.RecordSink(None, *cbs)
            vr = RR.VirtualRanks(P, ctx, r, world, tables)
            ok = ctx.lib.pbsim_job_run(ctx.h, C.byref(vr.comm), C.byref(sink))
            assert vr.error is None, vr.error
            P._check(ok)
            checked += vr.checked
            rounds += ctx.job_counters()["rounds"]
            for k in got:
                for w in (0, 1):
                    pieces[k][w].extend(got[k][w])
        # exchange A of every round a rank popped, and B of the rounds that could touch the quota, were checked against the table
        # (rounds begun behind a cut are dropped unexchanged; a round clear of the quota has no B)
        assert rounds // 2 <= checked < 2 * rounds and rounds >= 2 * world
        for k in want:
            for w in (0, 1):
                text = want[k][w]
                at = sorted((text.find(pc), len(pc)) for pc in pieces[k][w] if pc)
                pos = 0
                for start, n in at:                                 # the ranks' pieces tile the record's text: no gap, no overlap
                    assert start == pos, (method, world, k, w, start, pos)
                    pos += n
                assert pos == len(text), (method, world, k, w)


def test_arena_blocks_stay_from_job_to_job(monkeypatch, capfd):
    """The page-locked blocks a rank of a several-rank job compresses its rounds into stay with the context from job to job
    (ctx.h DfLane::arena_trim: a block goes back after sixteen idle jobs, not after one).  Rounds 2-5 gave every untouched block
    back at the end of every job: which slots and how many blocks a job touches varies from job to job, and giving a 256 MB
    block back and page-locking it again cost tens of milliseconds each -- replayed ranks 25-95 ms behind the others
    (profiles/r05z_replay_arena_churn.txt).  Here: four ranks replayed one after the other, jobs with large rounds and jobs with
    small rounds in turn, three passes; every block is allocated in the first pass, none goes back.  (A guard, not a reproduction:
    at test size a round is one piece per lane, and the churn needed the several 83 MB pieces per round of a full-size job.)"""
    import ctypes as C
    import torch
    import pbsim3_amd as P
    import replay_ranks as RR
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=5, depth=6.0, len_mean=1500.0, len_sd=1100.0)
    G, world = 1_500_000, 4
    monkeypatch.setenv("PBSIM_JOB_TARGET_RANKS", "200000")
    monkeypatch.setenv("PBSIM_PINNED_BLOCK_KB", "64")      # (test hook: a round's ~200 KB of members take several blocks, as 256 MB
    recs = [torch.frombuffer(bytearray(genome(G, 10 + i)), dtype=torch.uint8).cuda() for i in range(2)]   # blocks do at full size)
    tables = RR.build_tables(P, harness, p, "ERRHMM-ONT.model", False, recs, G, 0)
    with P.Context(p, 0) as ctx:
        ctx.set_scratch_bytes(256 << 20)
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_deflate(7)
        for t in recs:
            ctx.job_add_record_device(t.data_ptr(), G)
        cbs = (P.REC_TEXT_CB(lambda u, rec, t, n, o: 1), P.REC_TEXT_CB(lambda u, rec, t, n, o: 1), P.REC_DONE_CB(lambda u, rec, st, rb, mb: 1))
        sink = P.RecordSink(None, *cbs)
        monkeypatch.setenv("PBSIM_TRACE", "1")
        new_blocks = []
        for _ in range(3):
            capfd.readouterr()
            for target in ("400000", "100000"):          # jobs with large rounds, then jobs with small ones: the small ones leave
                monkeypatch.setenv("PBSIM_JOB_TARGET_RANKS", target)      # most blocks of a lane untouched
                for r in range(world):
                    vr = RR.VirtualRanks(P, ctx, r, world, tables)
                    ok = ctx.lib.pbsim_job_run(ctx.h, C.byref(vr.comm), C.byref(sink))
                    assert vr.error is None, vr.error
                    P._check(ok)
            err = capfd.readouterr().err
            assert "[pbsim arena] trim" not in err
            new_blocks.append(err.count("[pbsim arena] new block"))
        assert new_blocks[0] >= 2 and new_blocks[1:] == [0, 0], new_blocks
