"""The communicator of a one-process-per-GPU launch (include/pbsim3_amd.h pbsim_rccl_*: ncclGetUniqueId + ncclCommInitRank),
as a group of ONE -- all this box can offer (RCCL takes one rank per GPU): a whole job driven through it (PBSIM_COMM_ALWAYS=1:
the several-rank protocol with every exchange a real ncclAllGather / ncclAllReduce) delivers the bytes of the job without a
communicator, the id travels through a rendezvous file as well, and the `pbsim` binary's --rank / --world /
--rendezvous mode runs the command line on it.  What a group of one cannot show -- ranks waiting for each other -- is covered
by the host communicator with several contexts on the one GPU (tests/test_gpu_multi.py) and over gloo."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import harness
import pbsim3_amd as P
from cases import CASES

pytestmark = pytest.mark.gpu
CLI = os.path.join(harness.ROOT, "pbsim3_amd", "bin", "pbsim")


def job_bytes(ctx, comm_ref):
    reads, mafs, done = {}, {}, {}

    def on(store):
        def cb(user, rec, text, n, off):
            store.setdefault(rec, []).append((off, C.string_at(text, n)))
            return 1
        return cb

    def on_done(user, rec, st, rb, mb):
        done[rec] = (st.contents.res_num, st.contents.res_len_total, rb, mb)
        return 1
    cbs = (P.REC_TEXT_CB(on(reads)), P.REC_TEXT_CB(on(mafs)), P.REC_DONE_CB(on_done))
    sink = P.RecordSink(None, *cbs)
    P._check(ctx.lib.pbsim_job_run(ctx.h, comm_ref, C.byref(sink)))
    cat = lambda d: {r: b"".join(t for _, t in sorted(v)) for r, v in d.items()}   # noqa: E731
    return cat(reads), cat(mafs), done


def test_job_through_init_rank_communicator_of_one(tmp_path, monkeypatch):
    rng = np.random.default_rng(5)
    recs = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].tobytes() for n in (300_000, 180_000)]
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=3, depth=4.0)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_scratch_bytes(6 << 20)          # many rounds per record
        for r in recs:
            ctx.job_add_record(r)
        want = job_bytes(ctx, None)
        monkeypatch.setenv("PBSIM_COMM_ALWAYS", "1")
        for make in (lambda: P.RcclComm.create(0, 1, 0, lambda ident: ident),
                     lambda: P.RcclComm.from_file(str(tmp_path / "rdv"), 0, 1, 0)):
            cm = make()
            assert (cm.comm.rank, cm.comm.world) == (0, 1)
            info0 = cm.info()
            assert info0["ranks_seen"] == 1 and info0["rank"] == 0 and info0["device"] == 0
            got = job_bytes(ctx, cm.ref)
            assert got == want
            assert cm.info()["collectives"] > info0["collectives"] + 6     # the job's exchanges went through RCCL
            lat = P.comm_latency(cm.ref, 8, 200, 20)
            assert 0 < lat["all_gather_us"] < 5000 and 0 < lat["all_reduce_us"] < 5000
            # compressed members through the several-rank delivery (arena + offsets from the exchanges): same payload
            ctx.set_deflate(7)
            gz = job_bytes(ctx, cm.ref)
            ctx.set_deflate(0)
            import gzip
            assert {r: gzip.decompress(v) for r, v in gz[0].items()} == want[0]
            assert {r: gzip.decompress(v) for r, v in gz[1].items()} == want[1]
            cm.close()
    assert not os.path.exists(tmp_path / "rdv")      # rank 0 removes the id file once every rank has joined
    assert sum(v[0] for v in want[2].values()) > 100


def test_id_and_argument_errors():
    lib = P.load()
    assert lib.pbsim_rccl_unique_id(None, 0) == P.RCCL_ID_BYTES
    assert not lib.pbsim_rccl_comm_create(b"x" * 5, 5, 0, 1, 0)
    assert b"128 bytes" in lib.pbsim_last_error()
    out = (C.c_int64 * 4)()
    fake = P.make_comm(0, 1, lambda a: a.reshape(1, -1), lambda a, op: a)
    assert lib.pbsim_rccl_comm_info(C.byref(fake), out) == 0
    with pytest.raises(P.PbsimError, match="did not publish"):
        os.environ["PBSIM_RENDEZVOUS_TIMEOUT_S"] = "0.2"
        try:
            P.RcclComm.from_file("/tmp/pbsim_no_such_rendezvous_%d" % os.getpid(), 1, 2, 0)
        finally:
            del os.environ["PBSIM_RENDEZVOUS_TIMEOUT_S"]


def test_cli_one_process_per_gpu_mode(tmp_path):
    """`pbsim --rank 0 --world 1 --rendezvous FILE`: the selftest, then a golden case through pbsim_cli_main on that communicator"""
    p = subprocess.run([CLI, "--rank", "0", "--world", "1", "--rendezvous", str(tmp_path / "r1"), "--comm-selftest"],
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0 and "RCCL counts 1): ok" in p.stderr, p.stderr[-2000:]
    case = "wgs_errhmm-ont_quirk"
    work = tmp_path / "w"
    work.mkdir()
    p = subprocess.run([CLI] + harness.resolve(CASES[case]["args"]) + ["--prefix", str(work / "out"), "--no-gzip", "--rank", "0",
                                                                        "--world", "1", "--rendezvous", str(tmp_path / "r2")],
                       capture_output=True, text=True, cwd=str(work), timeout=280)
    assert p.returncode == 0, p.stderr[-4000:]
    outs = harness.collect(str(work))
    outs[".stderr"] = harness.strip_report(p.stderr).encode()
    want = harness.load_manifest()[f"{case}/philox"]
    assert sorted(outs) == sorted(want)
    for k, v in outs.items():
        assert harness.sha(v) == want[k]["sha256"], k
    bad = subprocess.run([CLI, "--rank", "2", "--world", "2", "--rendezvous", "x"], capture_output=True, text=True)
    assert bad.returncode == 255 and "--rank R --world N --rendezvous FILE" in bad.stderr
