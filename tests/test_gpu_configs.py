"""The BASELINE.json configurations that the golden cases do not reach on their own:

  configs[2]  wgs qshmm QSHMM-RSII --pass-num 10 (multi-pass / HiFi BAM path): byte-exact vs the oracle on the quirk genome
              (SAM text and native BAM), and a property test on a larger record (every subread's MAF rows consistent with its
              SAM record and with the genome; only pass 0 counts towards the quota, pbsim.cpp:2296-2298)
  configs[3]  trans errhmm ERRHMM-SEQUEL on a synthetic 100 000-transcript expression profile (full size): read count =
              sum of the expression values, strand split by the plus count (pbsim.cpp:4516-4522), start + length inside
              the transcript, MAF rows consistent with the transcripts
  configs[4]  wgs errhmm ERRHMM-ONT-HQ --depth 60 sharded over 2, 4 and 8 ranks: files and the full stderr report vs the
              oracle (N-rank bytes == 1-rank bytes == CPU restatement)
"""
import os
import subprocess

import numpy as np
import pytest

import harness
import test_gpu_bam
from test_gpu_multi import CLI, run_devices

pytestmark = pytest.mark.gpu
QUIRK = "INPUT:quirk.fa"
COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(b):
    return b.translate(COMP)[::-1]


def oracle(args, d):
    os.makedirs(d, exist_ok=True)
    return harness.run_oracle(args, "philox", str(d))


def cli(args, workdir, extra=()):
    os.makedirs(workdir, exist_ok=True)
    p = subprocess.run([CLI] + harness.resolve(args) + ["--prefix", os.path.join(workdir, "out")] + list(extra),
                       capture_output=True, text=True, cwd=workdir)
    assert p.returncode == 0, p.stderr[-3000:]
    outs = harness.collect(str(workdir))
    outs[".stderr"] = harness.strip_report(p.stderr).encode()
    return outs


# ---------------------------------------------------------------------------------------------------- configs[2]
PASS10 = ["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model", "--genome", QUIRK,
          "--depth", "20", "--pass-num", "10", "--seed", "1", "--length-mean", "1500", "--length-sd", "1100"]


def test_pass_num_10_matches_oracle(tmp_path):
    want = oracle(PASS10, tmp_path / "o")
    got = cli(PASS10, str(tmp_path / "t"), ["--no-gzip"])
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], k
    # default outputs: native BAM records, BGZF-framed on the GPU
    cli(PASS10, str(tmp_path / "b"))
    for k in [k for k in want if k.endswith(".sam")]:
        test_gpu_bam.compare_bam_with_sam((tmp_path / "b" / ("out" + k[:-4] + ".bam")).read_bytes(), want[k])
    # and sharded over three ranks with small rounds
    (tmp_path / "m").mkdir()
    multi = run_devices(PASS10, str(tmp_path / "m"), 3, scratch_mb=6)
    for k in want:
        assert multi[k] == want[k], k


def test_pass_num_10_properties_on_a_larger_record(tmp_path):
    """200 kbp x depth 20 x 10 passes = 40 M subread bases through the job pipeline (several rounds)"""
    import pbsim3_amd as P
    rng = np.random.default_rng(5)
    G = 200_000
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, G)].tobytes()
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_QS, seed=3, depth=20.0, pass_num=10)
    with P.Context(p, 0) as ctx:
        ctx.set_scratch_bytes(96 << 20)
        ctx.load_qshmm(harness.model_path("QSHMM-RSII.model"))
        ctx.job_add_record(genome)
        outs, done = ctx.job_run()
    st, rb, mb = done[1]
    sam, maf = bytes(outs[1][0]), bytes(outs[1][1])
    assert len(sam) == rb and len(maf) == mb
    recs = [l.split(b"\t") for l in sam.split(b"\n") if l]
    blocks = [b.split(b"\n") for b in maf.split(b"\n\n") if b]
    assert len(recs) == len(blocks) == st.res_num * 10 == st.res_pass_num
    quota = int(20.0 * G)
    pass0 = []
    total = 0
    for i, (f, blk) in enumerate(zip(recs, blocks)):
        read, h = i // 10 + 1, i % 10
        assert f[0] == b"S1/%d/%d" % (read, h) and f[1:9] == [b"4", b"*", b"0", b"255", b"*", b"*", b"0", b"0"]
        seq, qual = f[9], f[10]
        assert len(seq) == len(qual) > 0
        tags = {t[:2]: t for t in f[11:]}
        assert tags[b"ip"].count(b",") == len(seq) == tags[b"pw"].count(b",") and tags[b"zm"] == b"zm:i:%d" % read
        assert tags[b"qe"] == b"qe:i:%d" % (len(seq) - 1)
        assert blk[0] == b"a"
        r, q = blk[1].split(), blk[2].split()
        assert r[1] == b"ref" and r[4] == b"+" and int(r[5]) == G and q[1] == f[0] and int(q[3]) == len(seq) == int(q[5])
        start, size = int(r[2]), int(r[3])
        assert len(r[6]) == len(q[6])
        assert r[6].replace(b"-", b"") == genome[start:start + size]
        bases = q[6].replace(b"-", b"")
        assert (bases if q[4] == b"+" else revcomp(bases)) == seq
        assert q[4] == (b"+" if read % 2 == 1 else b"-")          # pbsim.cpp:2199-2203: strand by read parity
        total += len(seq)
        if h == 0:
            pass0.append(len(seq))
    assert total == st.res_len_total
    # only pass 0 counts towards the quota (pbsim.cpp:2296-2298): the loop ends with the read that reaches it
    assert sum(pass0) >= quota > sum(pass0[:-1])


# ---------------------------------------------------------------------------------------------------- configs[3]
def test_trans_100k_transcripts_properties():
    import pbsim3_amd as P
    rng = np.random.default_rng(1)
    n = 100_000
    lens = np.exp(rng.uniform(np.log(300), np.log(12000), n)).astype(np.int64)
    plus = rng.geometric(1 / 21.0, n) - 1
    minus = rng.geometric(1 / 1.1, n) - 1
    allseq = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, int(lens.sum()))].tobytes()
    offs = np.concatenate([[0], np.cumsum(lens)])
    seqs = [allseq[offs[i]:offs[i + 1]] for i in range(n)]
    ids = ["T%d" % i for i in range(n)]
    reads_of = plus + minus
    first_read = np.concatenate([[0], np.cumsum(reads_of)])       # 0-based global index of each transcript's first read
    R = int(reads_of.sum())
    p = P.default_params(strategy=P.STRATEGY_TRANS, method=P.METHOD_ERR, seed=1)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-SEQUEL.model"))
        ctx.set_transcripts(ids, [int(x) for x in plus], [int(x) for x in minus], seqs)
        assert ctx.unit_reads() == R
        ctx.simulate_trans(collect=False)                         # the whole job, text left on the GPU
        st = ctx.stats()
        assert st.res_num == R and st.res_len_min >= 1 and st.res_len_max <= 2 * int(lens.max())
        assert 0.80 < st.res_accuracy_mean < 0.90
        for first, cnt in ((1, 3000), (R // 2, 3000), (R - 2999, 3000)):
            fq, maf = ctx.simulate_units_range(first, cnt)
            names = fq.split(b"\n")[0::4][:cnt]
            blocks = [b.split(b"\n") for b in maf.split(b"\n\n") if b]
            assert len(blocks) == cnt
            for k, blk in enumerate(blocks):
                g = first - 1 + k                                 # 0-based global read index
                u = int(np.searchsorted(first_read, g, side="right") - 1)
                i = g - int(first_read[u]) + 1
                r, q = blk[1].split(), blk[2].split()
                assert names[k] == b"@S_%d" % (g + 1) == b"@" + q[1]
                assert r[1] == ids[u].encode() and int(r[5]) == int(lens[u])
                start, size = int(r[2]), int(r[3])
                assert 0 <= start and start + size <= int(lens[u]) and size >= 1
                assert q[4] == (b"+" if i <= plus[u] else b"-")   # pbsim.cpp:4516-4522
                assert r[6].replace(b"-", b"") == seqs[u][start:start + size]


# ---------------------------------------------------------------------------------------------------- configs[4]
ONTHQ60 = ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT-HQ.model", "--genome", QUIRK,
           "--depth", "60", "--seed", "1", "--length-mean", "1500", "--length-sd", "1100"]


@pytest.mark.parametrize("ranks,scratch_mb", [(2, 8), (4, 5), (8, 4)])
def test_onthq_depth_60_sharded_matches_oracle(ranks, scratch_mb, tmp_path):
    want = oracle(ONTHQ60, tmp_path / "o")
    (tmp_path / "m").mkdir()
    got = run_devices(ONTHQ60, str(tmp_path / "m"), ranks, scratch_mb=scratch_mb)
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], (k, got[k][-400:] if k == ".stderr" else len(got[k]))
