"""Bodies of tests/test_gpu_rccl_native.py, run in a CHILD process each: RCCL writes a banner to the stdout of any process that
initialises it (at that process's exit, behind pytest's own summary -- the driver reads the tail of pytest's output).
usage: python tests/rccl_native_driver.py job|errors TMPDIR"""
import ctypes as C
import gzip
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np  # noqa: E402

import harness  # noqa: E402
import pbsim3_amd as P  # noqa: E402

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


def job(tmp_path):
    rng = np.random.default_rng(5)
    recs = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].tobytes() for n in (300_000, 180_000)]
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=3, depth=4.0)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_scratch_bytes(6 << 20)          # many rounds per record
        for r in recs:
            ctx.job_add_record(r)
        want = job_bytes(ctx, None)
        os.environ["PBSIM_COMM_ALWAYS"] = "1"
        for make in (lambda: P.RcclComm.create(0, 1, 0, lambda ident: ident),
                     lambda: P.RcclComm.from_file(os.path.join(tmp_path, "rdv"), 0, 1, 0)):
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
            assert {r: gzip.decompress(v) for r, v in gz[0].items()} == want[0]
            assert {r: gzip.decompress(v) for r, v in gz[1].items()} == want[1]
            cm.close()
    assert not os.path.exists(os.path.join(tmp_path, "rdv"))      # rank 0 removes the id file once every rank has joined
    assert sum(v[0] for v in want[2].values()) > 100


def errors(tmp_path):
    lib = P.load()
    assert lib.pbsim_rccl_unique_id(None, 0) == P.RCCL_ID_BYTES
    assert not lib.pbsim_rccl_comm_create(b"x" * 5, 5, 0, 1, 0)
    assert b"128 bytes" in lib.pbsim_last_error()
    out = (C.c_int64 * 4)()
    fake = P.make_comm(0, 1, lambda a: a.reshape(1, -1), lambda a, op: a)
    assert lib.pbsim_rccl_comm_info(C.byref(fake), out) == 0
    os.environ["PBSIM_RENDEZVOUS_TIMEOUT_S"] = "0.2"
    try:
        P.RcclComm.from_file("/tmp/pbsim_no_such_rendezvous_%d" % os.getpid(), 1, 2, 0)
        raise AssertionError("a rank without rank 0's file must fail")
    except P.PbsimError as e:
        assert "did not publish" in str(e), e



if __name__ == "__main__":
    {"job": job, "errors": errors}[sys.argv[1]](sys.argv[2])
    print("RCCL-NATIVE-OK", sys.argv[1])
