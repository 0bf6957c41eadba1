"""The BENCHMARKED path at the BENCHMARKED size, content-checked.

bench.py times pbsim_job_run + pbsim_set_deflate(7) + the delivery threads + the wave walker at its default split on 750 Mbp
records at depth 20 and, at that size, only counts bytes.  Here the same path delivers one such record (15 Gbases, 1.7 M
reads, 63 GB of text, ~2 M gzip members) and every member is looked at: each carries the CRC-32 and the length of the 32 KiB
of text it holds (RFC 1952), and CRC-32 composes -- crc(A || B) = shift(crc(A), |B|) xor crc(B) -- so the members of a
stream, folded in offset order, give the CRC-32 and length of the record's WHOLE FASTQ resp. MAF text whatever the batch
partition, the rank count or the walker that produced it.  That pair must equal what the REFERENCE ITSELF wrote for the same record (round 5: tests/golden/fullsize.json, produced by
running pbsim.cpp with the keyed stream on harness.synth_bases(750 Mbp) for 15 CPU-minutes), and it and the record's statistics
must be identical for

    one rank | one rank without the wave walker | one rank with a small scratch pool (other batches) | three ranks

and the statistics equal pbsim_simulate_wgs's (the per-record driver).  Every 997th member is inflated with zlib and checked
against its own trailer (tests/member_walk.c, zlib), so the CRCs are CRCs of what the members really hold.  A second test compares 300 Mbases delivered
through the same path -- compressed sinks, default split -- with the oracle byte for byte."""
import ctypes as C
import gzip
import os
import struct
import threading
import zlib

import numpy as np
import pytest

import harness

pytestmark = pytest.mark.gpu

G = 750_000_000


# ---- the members of a piece -> (members, CRC-32, length) of the text they hold: tests/member_walk.c (host cc + zlib) ----
_helper = None


def helper():
    global _helper
    if _helper is None:
        import subprocess
        import tempfile
        d = tempfile.mkdtemp(prefix="member_walk_")
        so = os.path.join(d, "member_walk.so")
        subprocess.run(["cc", "-O2", "-shared", "-fPIC", os.path.join(harness.ROOT, "tests", "member_walk.c"), "-lz", "-o", so],
                       check=True)
        lib = C.CDLL(so)
        lib.walk_members.restype = C.c_long
        lib.walk_members.argtypes = [C.c_void_p, C.c_long, C.POINTER(C.c_ulong), C.POINTER(C.c_long), C.c_long, C.c_long]
        lib.fold.restype = C.c_ulong
        lib.fold.argtypes = [C.c_ulong, C.c_ulong, C.c_long]
        lib.walk_members(None, 0, C.byref(C.c_ulong(0)), C.byref(C.c_long(0)), 0, 0)   # builds its tables (once, before threads)
        _helper = lib
    return _helper


def walk_piece(ptr, n, first_index=0, sample_every=997):
    crc, ln = C.c_ulong(0), C.c_long(0)
    k = helper().walk_members(ptr, n, C.byref(crc), C.byref(ln), first_index, sample_every)
    assert k >= 0, "bad member at byte %d of a piece (framing, or a trailer that does not match its data)" % (-1 - k)
    return k, crc.value, ln.value


def test_member_walk_against_zlib():
    rng = np.random.default_rng(3)
    parts = [rng.integers(65, 70, n, dtype=np.uint8).tobytes() for n in (32768, 32768, 5, 32768, 1, 20000, 32768)]

    def member(data):
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        z = co.compress(data) + co.flush()
        return (bytes([31, 139, 8, 4, 0, 0, 0, 0, 0, 255, 6, 0, 66, 67, 2, 0]) + struct.pack("<H", 18 + len(z) + 8 - 1) + z +
                struct.pack("<II", zlib.crc32(data), len(data)))
    raw = b"".join(member(p) for p in parts)
    buf = C.create_string_buffer(raw, len(raw))
    assert walk_piece(C.cast(buf, C.c_void_p), len(raw), 0, 1) == (len(parts), zlib.crc32(b"".join(parts)), sum(map(len, parts)))
    bad = bytearray(raw)
    bad[-8] ^= 1                                     # a trailer that does not describe its data
    buf = C.create_string_buffer(bytes(bad), len(bad))
    with pytest.raises(AssertionError):
        walk_piece(C.cast(buf, C.c_void_p), len(bad), 0, 1)


class MemberSink:
    """receives (offset, members) pieces of the two streams of one record, possibly from several ranks and threads"""

    def __init__(self, P):
        self.P = P
        self.lock = threading.Lock()
        self.pieces = ([], [])       # per stream: (offset, nbytes, members, crc, text length)
        self.done = {}
        helper()

    def sink_for(self, rank):
        P = self.P

        def put(which, rec, text, n, off):
            k, crc, ln = walk_piece(C.cast(text, C.c_void_p), n)     # (ctypes releases the GIL for the walk)
            with self.lock:
                self.pieces[which].append((off, n, k, crc, ln))
            return 1

        def fin(user, rec, st, rb, mb):
            s = P.Stats()
            C.memmove(C.byref(s), st, C.sizeof(P.Stats))
            with self.lock:
                self.done[rank] = (s, rb, mb)
            return 1

        cbs = (P.REC_TEXT_CB(lambda u, r, t, n, o: put(0, r, t, n, o)), P.REC_TEXT_CB(lambda u, r, t, n, o: put(1, r, t, n, o)),
               P.REC_DONE_CB(fin))
        sink = P.RecordSink(None, *cbs)
        sink._keep = cbs
        return sink

    def digest(self):
        """per stream: (CRC-32 of the whole text, its length, compressed bytes, members)"""
        out = []
        for which in (0, 1):
            crc = ln = at = members = 0
            for off, n, k, pcrc, pln in sorted(self.pieces[which], key=lambda x: x[0]):
                assert off == at, "the pieces of a stream do not tile it"
                crc = helper().fold(crc, pcrc, pln)
                ln += pln
                members += k
                at += n
            assert at == self.done[0][1 + which]
            out.append((crc, ln, at, members))
        return out


def thread_comms(P, world):
    """pbsim_comm for `world` contexts in this process, one Python thread each: a barrier over shared lists"""
    bar = threading.Barrier(world)
    slots = [None] * world

    def make(rank):
        def all_gather(arr):
            slots[rank] = np.array(arr, dtype=np.int64)
            bar.wait()
            out = np.stack(slots)
            bar.wait()
            return out

        def all_reduce(arr, op):
            slots[rank] = np.array(arr, dtype=np.int64)
            bar.wait()
            stack = np.stack(slots)
            out = stack.sum(0) if op == P.OP_SUM else stack.min(0) if op == P.OP_MIN else stack.max(0)
            bar.wait()
            return out

        return P.make_comm(rank, world, all_gather, all_reduce, None, abort=bar.abort)
    return [make(r) for r in range(world)]


def stats_key(s):
    return (s.res_num, s.res_len_total, s.res_len_min, s.res_len_max, s.res_sub_num, s.res_ins_num, s.res_del_num,
            struct.pack("<dd", s.res_accuracy_mean, s.res_accuracy_sd), struct.pack("<dd", s.res_len_mean, s.res_len_sd))


def synth_record(case):
    """the record the REFERENCE was run on (tests/golden/make_fullsize.py): harness.synth_bases, here on the GPU"""
    import torch
    from fullsize_cases import FULLSIZE
    length, seed = FULLSIZE[case]["record"]
    t = harness.synth_bases_torch(length, seed, "cuda")
    torch.cuda.synchronize()
    return t


C1 = "c1_errhmm_ont_750m_d20"


@pytest.fixture(scope="module")
def record():
    t = synth_record(C1)
    assert t.numel() == G
    yield t
    del t


def reference_digest(case, P=None, ctx=None):
    """what the reference itself wrote for `case` (tests/golden/fullsize.json): [(crc, bytes) of the read stream, of the MAF
    stream] and its stderr report.  The reference's SAM stream starts with the two header lines (pbsim.cpp:721-722), which the
    record sink does not receive (the CLI writes them): their CRC is folded in front of the records' here."""
    e = harness.load_fullsize()[case]
    rk = ".fq" if ".fq" in e else ".sam"
    return [(int(e[rk]["crc32"], 16), e[rk]["bytes"]), (int(e[".maf"]["crc32"], 16), e[".maf"]["bytes"])], e["stderr"]


def run_job(P, record, world=1, scratch_gib=None, env=None, method="errhmm", model="ERRHMM-ONT.model", depth=20.0, pass_num=1):
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        msink = MemberSink(P)
        ctxs = []
        for r in range(world):
            qs = method == "qshmm"
            p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_QS if qs else P.METHOD_ERR, seed=1, depth=depth,
                                 pass_num=pass_num)
            ctx = P.Context(p, 0)
            (ctx.load_qshmm if qs else ctx.load_errhmm)(harness.model_path(model))
            if scratch_gib:
                ctx.set_scratch_bytes(int(scratch_gib * (1 << 30)))
            ctx.set_deflate(7)
            ctx.job_add_record_device(record.data_ptr(), record.numel())
            ctxs.append(ctx)
        comms = thread_comms(P, world) if world > 1 else [None]
        errs = [None] * world

        def one(r):
            sink = msink.sink_for(r)
            ok = ctxs[r].lib.pbsim_job_run(ctxs[r].h, C.byref(comms[r]) if comms[r] is not None else None, C.byref(sink))
            if not ok:
                errs[r] = ctxs[r].lib.pbsim_last_error().decode(errors="replace")

        th = [threading.Thread(target=one, args=(r,)) for r in range(1, world)]
        for t in th:
            t.start()
        one(0)
        for t in th:
            t.join()
        assert errs == [None] * world, errs
        counters = [c.job_counters() for c in ctxs]
        report = ctxs[0].format_stats(msink.done[0][0], 1)
        sam_header = ctxs[0].job_sam_header(1) if pass_num > 1 else b""
        for c in ctxs:
            c.close()
        assert all(stats_key(msink.done[r][0]) == stats_key(msink.done[0][0]) for r in range(world))   # merged: same everywhere
        digest = msink.digest()
        if sam_header:   # crc(header || records) = fold(crc(header), crc(records), |records|)
            crc, ln, at, members = digest[0]
            digest[0] = (helper().fold(zlib.crc32(sam_header), crc, ln), ln + len(sam_header), at, members)
        run_job.last_report = report
        return digest, stats_key(msink.done[0][0]), counters
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.timeout(900)
def test_benchmarked_path_content_is_invariant(record):
    import pbsim3_amd as P
    base_digest, base_stats, counters = run_job(P, record)
    (fq_crc, fq_n, fq_gz, fq_members), (maf_crc, maf_n, maf_gz, maf_members) = base_digest
    # ---- against the REFERENCE: pbsim.cpp itself (keyed stream) wrote this FASTQ and this MAF for this record, 15 CPU-minutes
    # of it (tests/golden/make_fullsize.py); every variant below must reproduce the same pair
    want, ref_report = reference_digest(C1)
    assert [(fq_crc, fq_n), (maf_crc, maf_n)] == want, "the 15-Gbase record differs from the reference's own output"
    assert run_job.last_report.rstrip("\n") in ref_report
    n_reads, bases = base_stats[0], base_stats[1]
    assert 15_000_000_000 <= bases < 15_000_000_000 + 1_000_000 and 1_500_000 < n_reads < 1_900_000
    assert counters[0]["rounds"] >= 3 and fq_members > 900_000 and maf_members > 900_000
    # FASTQ: "@S1_<n>\n" + bases + "\n+S1_<n>\n" + '!' x bases + "\n": 2 x bases + per-read framing
    assert 2 * bases < fq_n < 2 * bases + 40 * n_reads and maf_n > 2 * bases
    assert fq_gz < 0.2 * fq_n and maf_gz < 0.35 * maf_n
    variants = {
        "lane walker only (PBSIM_COOP_LEN=-1)": dict(env={"PBSIM_COOP_LEN": "-1"}),
        "6 GiB scratch pool (other batches)": dict(scratch_gib=6),
        "three ranks": dict(world=3),
        "three ranks, 3 GiB scratch pools": dict(world=3, scratch_gib=3),
    }
    for name, kw in variants.items():
        digest, stats, cnt = run_job(P, record, **kw)
        assert stats == base_stats, name
        assert [d[:2] for d in digest] == [(fq_crc, fq_n), (maf_crc, maf_n)], name
        if kw.get("world", 1) > 1:
            assert all(c["reads_delivered"] > 0 for c in cnt), name            # every rank delivered blocks
            assert sum(c["reads_delivered"] for c in cnt) == n_reads
    # the per-record driver reaches the same statistics (its own batching, no job pipeline)
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=1, depth=20.0)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_reference_device(record.data_ptr(), record.numel(), 1)
        ctx.simulate_wgs(collect=False)
        assert stats_key(ctx.stats()) == base_stats


@pytest.mark.timeout(900)
@pytest.mark.parametrize("case,runs", [
    ("c4_errhmm_onthq_120m_d60", [dict(world=1), dict(world=8)]),
    # one record of the BASELINE genome at the BASELINE depth: 45 Gbases, 5.1 M reads, 190 GB of text
    ("c4_errhmm_onthq_750m_d60", [dict(world=1), dict(world=4, scratch_gib=3)]),
])
def test_configs4_onthq_depth60_equals_the_reference(case, runs, tmp_path):
    """BASELINE configs[4] (ERRHMM-ONT-HQ, depth 60) -- on a 120 Mbp record (7.2 Gbases; one rank and eight ranks: contexts on the
    one GPU, host communicator) and on a 750 Mbp record (45 Gbases; one rank and four): the FASTQ and MAF the reference wrote, by
    CRC-32 and length, and its report."""
    import pbsim3_amd as P
    rec = synth_record(case)
    want, ref_report = reference_digest(case)
    for kw in runs:
        digest, stats, cnt = run_job(P, rec, model="ERRHMM-ONT-HQ.model", depth=60.0, **kw)
        assert [d[:2] for d in digest] == want, kw
        assert run_job.last_report.rstrip("\n") in ref_report
        if kw["world"] > 1:
            assert all(c["reads_delivered"] > 0 for c in cnt)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("case,runs", [
    ("c2_qshmm_rsii_20m_d20_pass10", [dict(world=1), dict(world=1, env={"PBSIM_COOP_LEN": "-1"}), dict(world=3)]),
    ("c2_qshmm_rsii_60m_d20_pass10", [dict(world=1), dict(world=2, scratch_gib=6)]),
])
def test_configs2_qshmm_pass10_equals_the_reference(case, runs, tmp_path):
    """BASELINE configs[2] (QSHMM-RSII --pass-num 10, depth 20) on a 20 Mbp record (4 G subread bases) and a 60 Mbp one (12 G, 72 GB
    of SAM text): the SAM text (the native BAM records are checked field by field in test_gpu_bam.py) and the MAF the reference
    wrote, by CRC-32 and length."""
    import pbsim3_amd as P
    rec = synth_record(case)
    want, ref_report = reference_digest(case)
    for kw in runs:
        digest, stats, cnt = run_job(P, rec, method="qshmm", model="QSHMM-RSII.model", depth=20.0, pass_num=10, **kw)
        assert [d[:2] for d in digest] == want, kw
        assert run_job.last_report.rstrip("\n") in ref_report


def test_compressed_job_path_matches_oracle_300_mbases(tmp_path):
    """15 Mbp x depth 20 through pbsim_job_run + pbsim_set_deflate(7) with the default lane / wave split (one batch of
    ~33 k reads: reads of half a mean length and more go to the wave walker): inflated, the two streams equal the
    oracle's files byte for byte; so do the statistics."""
    import pbsim3_amd as P
    rng = np.random.default_rng(99)
    seq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 15_000_000)]
    fa = tmp_path / "g.fa"
    with open(fa, "wb") as f:
        f.write(b">chr1\n")
        lines = seq.reshape(-1, 100)
        f.write(np.concatenate([lines, np.full((lines.shape[0], 1), 10, np.uint8)], axis=1).tobytes())
    args = ["--strategy", "wgs", "--method", "errhmm", "--errhmm", harness.model_path("ERRHMM-ONT.model"),
            "--genome", str(fa), "--depth", "20", "--seed", "21"]
    od = tmp_path / "o"
    od.mkdir()
    want = harness.run_oracle(args, "philox", str(od))
    assert len(want["_0001.fq"]) > 600_000_000
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=21, depth=20.0)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        ctx.set_deflate(7)
        ctx.job_add_record(seq.tobytes())
        texts, done = ctx.job_run()
        st = done[1][0]
        rep = ctx.format_stats(st, 1)
    for which, key in ((0, "_0001.fq"), (1, "_0001.maf")):
        got = gzip.decompress(bytes(texts[1][which]))      # multi-member
        assert len(got) == len(want[key]) and got == want[key], key
    assert rep.rstrip("\n") in want[".stderr"].decode()
