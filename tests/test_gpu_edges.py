"""Edge cases and size-independent properties on the GPU.

Edge cases run the product and the oracle (keyed-Philox mode) on the same seeded
inputs and compare bytes.  The large run is checked through properties that hold
at any size (the oracle would take minutes there): every MAF record is
self-consistent with its FASTQ record and with the genome."""
import os

import numpy as np
import pytest

import harness
import product

pytestmark = pytest.mark.gpu
ONT = ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model"]
QS = ["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model"]


def _write_fasta(path, recs):
    with open(path, "w", encoding="latin-1") as f:
        for i, s in enumerate(recs, 1):
            f.write(f">r{i}\n")
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + "\n")


def _rand_seq(rng, n, alphabet="ACGT"):
    return "".join(np.array(list(alphabet))[rng.integers(0, len(alphabet), n)])


def _compare(args, tmp_path, scratch_mb=None):
    outs, _ = product.run_wgs(harness.resolve(args), scratch_mb=scratch_mb)
    want = harness.run_oracle(args, "philox", str(tmp_path))
    for k, v in outs.items():
        assert v == want[k], (k, len(v), len(want[k]))


EDGE = {
    # reads longer than the record: L >= genome.len -> offset 0, L = genome.len (pbsim.cpp:3804-3806)
    "reads_longer_than_genome": (lambda rng: [_rand_seq(rng, 150), _rand_seq(rng, 100)],
                                 ONT + ["--depth", "30", "--seed", "11"]),
    # short reads down to --length-min 30 (much shorter ones make the reference index freq_accuracy[] with a negative
    # accuracy, pbsim.cpp:4004-4005: memory corruption there, skipped here); quota truncation floors at len_min (pbsim.cpp:3797-3799)
    "len_min_30": (lambda rng: [_rand_seq(rng, 3000)],
                  ONT + ["--depth", "4", "--seed", "12", "--length-min", "30", "--length-mean", "60", "--length-sd", "30"]),
    # depth below one read: the first read is already truncated
    "quota_below_one_read": (lambda rng: [_rand_seq(rng, 20000)], ONT + ["--depth", "0.05", "--seed", "13"]),
    # a genome of homopolymers only (hp 10/11 oscillation everywhere, Q1) and N runs
    "homopolymers": (lambda rng: ["".join(c * n for c, n in zip("ACGTN" * 40, rng.integers(1, 30, 200)))],
                     ONT + ["--depth", "20", "--seed", "14", "--length-mean", "300", "--length-sd", "200"]),
    # runs spanning several 4096-base tiles of the hp kernels: 9001 x A (odd -> hp 11), 12000 x C (even -> hp 10), 10000 x N (hp 1)
    "long_runs": (lambda rng: [_rand_seq(rng, 3000) + "A" * 9001 + _rand_seq(rng, 500) + "C" * 12000 + "G" + "N" * 10000 +
                               _rand_seq(rng, 2500)],
                  ONT + ["--depth", "8", "--seed", "19", "--length-mean", "2500", "--length-sd", "1500", "--hp-del-bias", "3"]),
    # IUPAC soup: non-ACGT substitution branch on nearly every substitution (pbsim.cpp:3947-3949)
    "iupac": (lambda rng: [_rand_seq(rng, 5000, "ACGTRYKMSWN")],
              ONT + ["--depth", "10", "--seed", "15", "--length-mean", "500", "--length-sd", "300"]),
    # long id prefix and many tiny reads: MAF column padding rules (pbsim.cpp:4030-4078)
    "id_prefix_padding": (lambda rng: [_rand_seq(rng, 1200)],
                          ONT + ["--depth", "200", "--seed", "16", "--length-mean", "130", "--length-sd", "40",
                                 "--id-prefix", "Sample_ABC.x"]),
    # QSHMM: leading insertions at column 0 (Q15), classes without a model (freq2qc), 5 passes
    "qshmm_pass5": (lambda rng: [_rand_seq(rng, 4000) + "A" * 11 + _rand_seq(rng, 500)],
                    QS + ["--depth", "3", "--seed", "17", "--pass-num", "5", "--length-mean", "400", "--length-sd", "300",
                          "--accuracy-mean", "0.80"]),
    # bytes >= 0x80 in the record (runs of them too: hp 11 of a non-ASCII byte): bit 7 of the sequence bytes cannot
    # carry the hp == 11 flag, the walks must fall back to the hp byte array (k_hp_breaks high_bytes)
    "non_ascii_bytes": (lambda rng: [_rand_seq(rng, 1500) + "\xc4" * 13 + _rand_seq(rng, 700, "ACGT\x80\xe9") + "A" * 11 +
                                     _rand_seq(rng, 1500) + "\xff" * 12 + _rand_seq(rng, 300)],
                        ONT + ["--depth", "25", "--seed", "21", "--length-mean", "400", "--length-sd", "250"]),
    # scratch pool far too small for one default batch: the driver must shrink batches, same bytes
    "many_small_batches": (lambda rng: [_rand_seq(rng, 600000)],
                           ONT + ["--depth", "15", "--seed", "18", "--length-mean", "1000", "--length-sd", "700"]),
}


@pytest.mark.parametrize("name", sorted(EDGE))
def test_edge_case_matches_oracle(name, tmp_path):
    make, args = EDGE[name]
    rng = np.random.default_rng(abs(hash(name)) % 2**31)
    fa = tmp_path / "g.fa"
    _write_fasta(fa, make(rng))
    _compare(args + ["--genome", str(fa)], tmp_path, scratch_mb=8 if name == "many_small_batches" else None)


def test_large_run_properties(tmp_path):
    """20 Mbp x depth 5 (100 Mbases, ~11 k reads): properties that hold at any size."""
    rng = np.random.default_rng(99)
    genome = _rand_seq(rng, 20_000_000)
    fa = tmp_path / "big.fa"
    with open(fa, "w") as f:
        f.write(">chr1\n")
        f.write("\n".join(genome[i:i + 80] for i in range(0, len(genome), 80)))
        f.write("\n")
    args = harness.resolve(ONT + ["--depth", "5", "--seed", "21", "--genome", str(fa)])
    outs, stats = product.run_wgs(args)
    fq = outs["_0001.fq"].split(b"\n")
    maf = outs["_0001.maf"].split(b"\n")
    n = stats[0].res_num
    assert len(fq) == 4 * n + 1 and len(maf) == 4 * n + 1
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    total = 0
    gb = genome.encode()
    for r in range(n):
        rid, seq, plus, qual = fq[4 * r:4 * r + 4]
        assert rid == b"@S1_%d" % (r + 1) and plus == b"+S1_%d" % (r + 1)
        assert len(seq) == len(qual) and set(qual) <= {ord("!")}
        a, ref_line, read_line, blank = maf[4 * r:4 * r + 4]
        assert a == b"a" and blank == b""
        rf = ref_line.split()
        rd = read_line.split()
        assert rf[0] == b"s" and rf[1] == b"ref" and rf[4] == b"+" and int(rf[5]) == len(genome)
        start, span, ref_row = int(rf[2]), int(rf[3]), rf[6]
        strand, read_row = rd[4], rd[6]
        assert rd[1] == b"S1_%d" % (r + 1) and int(rd[3]) == len(seq) == int(rd[5])
        assert strand == (b"+" if (r + 1) % 2 == 1 else b"-")
        assert len(ref_row) == len(read_row)
        # the reference row without gaps is the genome segment; the read row without gaps is the read
        assert ref_row.replace(b"-", b"") == gb[start:start + span]
        got = read_row.replace(b"-", b"")
        if strand == b"-":
            got = got.translate(comp)[::-1]
        assert got == seq
        total += len(seq)
    assert total == stats[0].res_len_total
    quota = int(5 * len(genome))
    assert quota <= total < quota + 200_000
    # error budget of the default accuracy distribution (classes 63..89, mean ~0.85)
    err = stats[0].res_sub_rate + stats[0].res_ins_rate + stats[0].res_del_rate
    assert 0.10 < err < 0.25 and 0.80 < stats[0].res_accuracy_mean < 0.90
    assert abs((1 - stats[0].res_accuracy_mean) - err) < 0.02


def test_multi_batch_quota_pipeline_matches_oracle(tmp_path):
    """3 records x 2 Mbp x depth 12 with a 64 MiB scratch pool: ~10 pipelined batches per
    record, the quota cut in the last one and the serial truncated tail -- bytes equal the oracle."""
    rng = np.random.default_rng(4242)
    fa = tmp_path / "g.fa"
    _write_fasta(fa, [_rand_seq(rng, 2_000_000) for _ in range(3)])
    args = ONT + ["--depth", "12", "--seed", "77", "--genome", str(fa)]
    _compare(args, tmp_path, scratch_mb=64)


def test_full_size_record_is_invariant_under_batching():
    """BASELINE configs[1] size (one 750 Mbp record x depth 20 = 15 Gbases, 1.7 M reads): the same reads whatever
    the batch partition -- a 10 GiB and a 40 GiB scratch pool cut the record into different batches, yet every counter,
    the quota-cut read count and the ordered double sum behind the mean accuracy are identical.  Property test: no oracle
    can walk 15 Gbases in seconds."""
    import torch
    import pbsim3_amd as P
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    idx = torch.randint(0, 4, (750_000_000,), device="cuda", dtype=torch.uint8, generator=g)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device="cuda")
    genome = torch.empty_like(idx)
    for lo in range(0, idx.numel(), 1 << 27):      # gather in pieces: index tensors are 8 bytes per element
        genome[lo:lo + (1 << 27)] = lut[idx[lo:lo + (1 << 27)].long()]
    del idx
    results = []
    for gib in (10, 40):
        p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=1, depth=20.0)
        with P.Context(p, 0) as ctx:
            ctx.set_scratch_bytes(gib << 30)
            ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
            ctx.set_reference_device(genome.data_ptr(), genome.numel(), 1)
            ctx.simulate_wgs(collect=False)
            st = ctx.stats()
            results.append((st.res_num, st.res_len_total, st.res_len_min, st.res_len_max, st.res_sub_num,
                            st.res_ins_num, st.res_del_num, round(st.res_accuracy_mean, 12)))
        torch.cuda.synchronize()
    assert results[0] == results[1], results
    n, total = results[0][0], results[0][1]
    assert 15_000_000_000 <= total < 15_000_000_000 + 1_000_000 and 1_500_000 < n < 1_900_000
    sub, ins, dele = (results[0][k] / total for k in (4, 5, 6))
    assert 0.02 < sub + ins + dele < 0.2


def test_two_contexts_alive_at_once(tmp_path):
    """Two contexts in one process (different seeds and methods), used alternately record by record: each produces what it
    produces alone.  Nothing in the library is process-global except the last-error string."""
    import pbsim3_amd as P
    from pbsim3_amd import args as A
    SHORT = ["--length-mean", "1200", "--length-sd", "900"]
    recs = A.read_fasta(os.path.join(harness.GOLDEN, "inputs", "quirk.fa"))[0]
    a_args = harness.resolve(["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model",
                              "--genome", "INPUT:quirk.fa", "--depth", "4", "--seed", "31"] + SHORT)
    b_args = harness.resolve(["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model",
                              "--genome", "INPUT:quirk.fa", "--depth", "3", "--seed", "32", "--pass-num", "2"] + SHORT)
    alone_a, _ = product.run_wgs(a_args)
    alone_b, _ = product.run_wgs(b_args)
    pa, _ = A.parse(a_args)
    pb, _ = A.parse(b_args)
    got_a, got_b = {}, {}
    with P.Context(pa, 0) as ca, P.Context(pb, 0) as cb:
        ca.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        cb.load_qshmm(harness.model_path("QSHMM-RSII.model"))
        for i, r in enumerate(recs, 1):
            ca.set_reference(r, i)
            cb.set_reference(r, i)
            rt_b, mt_b = cb.simulate_wgs()
            rt_a, mt_a = ca.simulate_wgs()
            got_a["_%04d.fq" % i], got_a["_%04d.maf" % i] = rt_a, mt_a
            got_b["_%04d.sam" % i], got_b["_%04d.maf" % i] = cb.sam_header() + rt_b, mt_b
    assert got_a == alone_a and got_b == alone_b


def test_prefetched_reference_is_adopted_and_changes_nothing():
    """pbsim_prefetch_reference_device: the next record is uploaded and prepared beside the current simulation; the
    bytes of every record equal a run without prefetch, also when a prefetch is dropped (other pointer) or unused"""
    import torch
    import pbsim3_amd as P
    from pbsim3_amd import args as A
    argv = harness.resolve(ONT + ["--genome", "INPUT:quirk.fa", "--depth", "6", "--seed", "31"])
    p, a = A.parse(argv)
    recs = A.read_fasta(a["--genome"])[0]
    rng = np.random.default_rng(5)
    recs = recs + [_rand_seq(rng, 30000).encode() + b"A" * 13 + _rand_seq(rng, 9000).encode()]
    dev = [torch.frombuffer(bytearray(r), dtype=torch.uint8).cuda() for r in recs]
    decoy = torch.zeros(1000, dtype=torch.uint8, device="cuda") + 65

    def run(mode):
        outs = []
        with P.Context(p, 0) as ctx:
            ctx.load_errhmm(a["--errhmm"])
            for i, t in enumerate(dev):
                ctx.set_reference_device(t.data_ptr(), t.numel(), i + 1)
                if mode == "prefetch" and i + 1 < len(dev):
                    ctx.prefetch_reference_device(dev[i + 1].data_ptr(), dev[i + 1].numel())
                if mode == "decoy":
                    ctx.prefetch_reference_device(decoy.data_ptr(), decoy.numel())
                outs.append(ctx.simulate_wgs())
        return outs

    plain = run("none")
    assert run("prefetch") == plain
    assert run("decoy") == plain

    # the host-pointer variant (pbsim_prefetch_reference): same pointer and length at pbsim_set_reference adopt the copy
    import ctypes as C
    bufs = [C.create_string_buffer(r, len(r)) for r in recs]
    outs = []
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(a["--errhmm"])
        for i, b in enumerate(bufs):
            P._check(ctx.lib.pbsim_set_reference(ctx.h, C.cast(b, C.c_void_p), len(recs[i]), i + 1))
            if i + 1 < len(bufs):
                P._check(ctx.lib.pbsim_prefetch_reference(ctx.h, C.cast(bufs[i + 1], C.c_char_p), len(recs[i + 1])))
            outs.append(ctx.simulate_wgs())
    assert outs == plain


def test_records_added_as_fasta_lines(tmp_path):
    """pbsim_job_add_record_lines: a record handed over as its FASTA lines (line feeds included) and squeezed on the GPU
    (k_lines_count / k_lines_squeeze: get_genome_seq's copy loop, pbsim.cpp:1014-1033) gives the bytes of the same record
    handed over squeezed -- line widths around the kernels' 16-byte loads and 4096-byte tiles, empty lines, CR LF, no final
    line feed, a record of one long line."""
    import pbsim3_amd as P
    rng = np.random.default_rng(3)

    def bases(n):
        return bytes(np.frombuffer(b"ACGTacgtN", dtype=np.uint8)[rng.integers(0, 9, n)])

    def wrap(s, w):
        return b"".join(s[i:i + w] + b"\n" for i in range(0, len(s), w))
    a, b, c, d = bases(300_007), bases(120_000), bases(65_536 * 3), bases(50_001)
    shaped = [wrap(a, 80), wrap(b[:60_000], 15) + b"\n\n" + wrap(b[60_000:], 4096) + b[:0], wrap(c, 4095)[:-1],
              b"".join(d[i:i + 61] + b"\r\n" for i in range(0, len(d), 61)), bases(200_000)]
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=9, depth=4.0, len_mean=1500.0, len_sd=1000.0)
    outs = []
    for as_lines in (False, True):
        with P.Context(p, 0) as ctx:
            ctx.set_scratch_bytes(64 << 20)
            ctx.load_errhmm(harness.model_path("ERRHMM-SEQUEL.model"))
            for s in shaped:
                if as_lines:
                    ctx.job_add_record_lines(s)
                else:
                    ctx.job_add_record(s.replace(b"\n", b""))
            texts, done = ctx.job_run()
            outs.append({k: (bytes(v[0]), bytes(v[1]), done[k][0].res_num) for k, v in texts.items()})
    assert len(outs[0]) == len(shaped) and outs[0] == outs[1]
    with P.Context(p, 0) as ctx:       # a wrong length is refused, not simulated
        with pytest.raises(P.PbsimError, match="not the number of bytes"):
            P._check(ctx.lib.pbsim_job_add_record_lines(ctx.h, shaped[0], len(shaped[0]), len(a) - 1))


@pytest.mark.parametrize("method", ["errhmm", "qshmm"])
def test_interleaved_records_change_nothing(method, monkeypatch):
    """pbsim_job_set_interleave / PBSIM_JOB_INTERLEAVE: the rounds of k records alternate (the record furthest behind first)
    instead of running record by record -- the same bytes at the same offsets and the same statistics, with more records than
    the window holds, records of very different sizes, tails of several records due at once (two chain slots), tiny pools."""
    import pbsim3_amd as P
    recs = [harness.synth_bases(n, 40 + i).tobytes() for i, n in enumerate((400_000, 90_000, 650_000, 120_000, 300_000, 100_000, 510_000))]
    qs = method == "qshmm"
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_QS if qs else P.METHOD_ERR, seed=21, depth=5.0,
                         len_mean=1300.0, len_sd=1000.0, pass_num=2 if qs else 1)
    outs = {}
    for k in (1, 2, 3, 5, 9):
        monkeypatch.setenv("PBSIM_JOB_INTERLEAVE", str(k))
        with P.Context(p, 0) as ctx:
            ctx.set_scratch_bytes(12 << 20)
            (ctx.load_qshmm if qs else ctx.load_errhmm)(harness.model_path("QSHMM-RSII.model" if qs else "ERRHMM-ONT.model"))
            for r in recs:
                ctx.job_add_record(r)
            texts, done = ctx.job_run()
            outs[k] = {i: (bytes(v[0]), bytes(v[1]), tuple(getattr(done[i][0], f[0]) for f in done[i][0]._fields_), done[i][1:])
                       for i, v in texts.items()}
    assert len(outs[1]) == len(recs)
    for k in outs:
        assert outs[k] == outs[1], k


@pytest.mark.parametrize("case", ["wgs_errhmm-ont_quirk", "wgs_qshmm_rsii_pass3", "wgs_errhmm_rsii_default"])
@pytest.mark.parametrize("factor", ["1.02", "1.2", "2"])
def test_rows_laid_out_tighter_than_two_lengths(case, factor, monkeypatch):
    """Scratch rows of factor x length + 64 columns (ctx.h scratch_factor): 2 is the reference's bound; at 1.02 most batches hold
    a read that runs out of row and are walked again at 2 inside pbsim_batch_walk_end -- the bytes are the goldens' either way
    (and the re-walks really happen)."""
    import pbsim3_amd as P
    from cases import CASES
    from test_gpu_parity import MANIFEST
    monkeypatch.setenv("PBSIM_SCRATCH_FACTOR", factor)
    args = harness.resolve(CASES[case]["args"])
    p, a = product.params_from_args(args)
    outs = {}
    with P.Context(p, 0) as ctx:
        ctx.set_scratch_bytes(product.scratch_mb_for(case) << 22)
        (ctx.load_errhmm if p.method == P.METHOD_ERR else ctx.load_qshmm)(a["--errhmm" if p.method == P.METHOD_ERR else "--qshmm"])
        for r in product.read_fasta(a["--genome"]):
            ctx.job_add_record(r)
        texts, done = ctx.job_run()
        state = ctx.scratch_state()
        for i in sorted(texts):
            rt = bytes(texts[i][0])
            outs["_%04d.%s" % (i, "fq" if p.pass_num == 1 else "sam")] = (ctx.job_sam_header(i) if p.pass_num > 1 else b"") + rt
            outs["_%04d.maf" % i] = bytes(texts[i][1])
    gold = MANIFEST[f"{case}/philox"]
    for k, v in outs.items():
        assert harness.sha(v) == gold[k]["sha256"], (case, factor, k)
    assert state[0] == float(factor)
    assert (state[2] > 0) == (state[1] > float(factor)), state      # walked twice exactly when a read needed more than its row
    assert state[2] > 0 or factor != "1.02"


def test_rows_follow_what_the_reads_need():
    """without the knob: a context starts at the reference's 2 x length, learns from its first walk (the job's probe) what the
    model's reads take, and lays the job out with that + 0.08 -- about 1.2 for ERRHMM-ONT at the default accuracy"""
    import pbsim3_amd as P
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=3, depth=8.0)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
        assert ctx.scratch_state()[0] == 2.0
        ctx.job_add_record(harness.synth_bases(3_000_000, 5).tobytes())
        ctx.job_run(collect=False)
        f, need, rewalks = ctx.scratch_state()
        assert 1.05 < need < 1.4 and abs(f - (need + 0.08)) < 1e-9 and rewalks == 0, (f, need, rewalks)
