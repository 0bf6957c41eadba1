"""k_walk_errhmm_coop (one wave per long read) against k_walk_errhmm (one lane per read): the two walks of the ERRHMM
path must produce the same bytes whatever share of the reads each of them takes.  PBSIM_COOP_LEN picks the share: -1 none,
0 every read, n the reads of at least n bases (rounded up to the sort's 256-base bucket); the default is three mean lengths,
and every read of a small batch.  The golden cases (small batches: every read on the wave walker by default) are repeated
on the lane walker."""
import numpy as np
import pytest

import harness
import product
from cases import CASES
from test_gpu_parity import MANIFEST, WGS_ERR

pytestmark = pytest.mark.gpu


def genome(n, seed):
    rng = np.random.default_rng(seed)
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
    for p in rng.integers(0, n - 40, n // 2000):          # homopolymers past the hp == 11 class, and some non-ACGT bases
        g[p:p + int(rng.integers(8, 30))] = g[p]
    for p in rng.integers(0, n - 8, n // 5000):
        g[p:p + int(rng.integers(1, 6))] = ord("N")
    return g.tobytes()


def run(coop, monkeypatch, records, model="ERRHMM-ONT.model", **kw):
    import pbsim3_amd as P
    if coop is None:
        monkeypatch.delenv("PBSIM_COOP_LEN", raising=False)
    else:
        monkeypatch.setenv("PBSIM_COOP_LEN", str(coop))
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, **kw)
    with P.Context(p, 0) as ctx:
        ctx.set_scratch_bytes(256 << 20)
        ctx.load_errhmm(harness.model_path(model))
        for r in records:
            ctx.job_add_record(r)
        outs, done = ctx.job_run()
    texts = {k: (bytes(v[0]), bytes(v[1])) for k, v in outs.items()}
    stats = {k: tuple(getattr(v[0], f[0]) for f in v[0]._fields_) + tuple(v[1:]) for k, v in done.items()}
    return texts, stats


RUNS = {
    "ont_default": dict(seed=7, depth=15.0),
    "onthq_hp_bias": dict(seed=3, depth=10.0, model="ERRHMM-ONT-HQ.model", hp_del_bias=2.0),
    "sequel_pass3_short_reads": dict(seed=5, depth=6.0, model="ERRHMM-SEQUEL.model", pass_num=3, len_mean=3000.0, len_sd=2500.0),
    "rsii_high_accuracy": dict(seed=11, depth=8.0, model="ERRHMM-RSII.model", accuracy_mean=0.99),
    "ont_low_accuracy_deletions": dict(seed=13, depth=8.0, accuracy_mean=0.70, sub_ratio=1, ins_ratio=1, del_ratio=8),
    "ont_short_min_length": dict(seed=17, depth=8.0, len_mean=400.0, len_sd=300.0, len_min=1),
}


@pytest.mark.parametrize("name", sorted(RUNS))
def test_wave_walker_matches_lane_walker(name, monkeypatch):
    recs = [genome(1_500_000, 1), genome(700_000, 2)]
    want = run(-1, monkeypatch, recs, **RUNS[name])
    assert sum(len(a) + len(b) for a, b in want[0].values()) > 10_000_000
    for coop in (0, None, 4096, 20000):
        got = run(coop, monkeypatch, recs, **RUNS[name])
        assert got[1] == want[1], (name, coop)
        for k in want[0]:
            assert got[0][k] == want[0][k], (name, coop, k)


@pytest.mark.parametrize("case", WGS_ERR + ["wgs_errhmm_sequel_pass3"])
@pytest.mark.parametrize("coop", [-1, 512])
def test_goldens_with_another_share(case, coop, monkeypatch):
    monkeypatch.setenv("PBSIM_COOP_LEN", str(coop))
    outs, _ = product.run_wgs_job(harness.resolve(CASES[case]["args"]), scratch_mb=product.scratch_mb_for(case))
    gold = MANIFEST[f"{case}/philox"]
    for k, v in outs.items():
        assert harness.sha(v) == gold[k]["sha256"], (case, coop, k)


def test_split_on_several_ranks(tmp_path, monkeypatch):
    """the rounds of a multi-rank job with both walkers at work: thread ranks on one GPU vs the oracle"""
    from test_gpu_multi import run_devices
    args = ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT-HQ.model", "--genome", "INPUT:quirk.fa",
            "--depth", "40", "--seed", "4", "--length-mean", "1500", "--length-sd", "1100"]
    (tmp_path / "o").mkdir()
    want = harness.run_oracle(args, "philox", str(tmp_path / "o"))
    for coop in ("1024", "2560"):
        monkeypatch.setenv("PBSIM_COOP_LEN", coop)
        d = tmp_path / ("m" + coop)
        d.mkdir()
        got = run_devices(args, str(d), 3, scratch_mb=6)
        for k in want:
            assert got[k] == want[k], (coop, k)


# ---- k_walk_qshmm_coop (round 3): the QSHMM walk with one wave per task, against the lane walker and the goldens ----------
def run_qs(coop, monkeypatch, records, model="QSHMM-RSII.model", **kw):
    import pbsim3_amd as P
    if coop is None:
        monkeypatch.delenv("PBSIM_COOP_LEN", raising=False)
    else:
        monkeypatch.setenv("PBSIM_COOP_LEN", str(coop))
    p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_QS, **kw)
    with P.Context(p, 0) as ctx:
        ctx.set_scratch_bytes(256 << 20)
        ctx.load_qshmm(harness.model_path(model))
        for r in records:
            ctx.job_add_record(r)
        outs, done = ctx.job_run()
        waves = ctx.prof_wave_launches()
    # the wave walker really took part (or really did not).  QSHMM-ONT-HQ is not eligible: its classes name states above
    # STATE_MAX and not all of its moduli are 100 -- every task stays with the lane walker, whatever the split
    assert (waves > 0) == (coop != -1 and "ONT-HQ" not in model), (coop, waves)
    texts = {k: (bytes(v[0]), bytes(v[1])) for k, v in outs.items()}
    stats = {k: tuple(getattr(v[0], f[0]) for f in v[0]._fields_) + tuple(v[1:]) for k, v in done.items()}
    return texts, stats


QS_RUNS = {
    "rsii_default": dict(seed=7, depth=10.0),
    "rsii_pass3": dict(seed=9, depth=4.0, pass_num=3, len_mean=4000.0, len_sd=3000.0),
    "ont_deletion_heavy": dict(seed=3, depth=8.0, model="QSHMM-ONT.model", sub_ratio=5, ins_ratio=10, del_ratio=85, accuracy_mean=0.80),
    "ont_insertion_heavy": dict(seed=5, depth=8.0, model="QSHMM-ONT.model", sub_ratio=5, ins_ratio=85, del_ratio=10),
    "onthq_many_states": dict(seed=11, depth=8.0, model="QSHMM-ONT-HQ.model", accuracy_mean=0.95),
    "rsii_short_min_length": dict(seed=17, depth=6.0, len_mean=400.0, len_sd=300.0, len_min=1),
}


@pytest.mark.parametrize("name", sorted(QS_RUNS))
def test_qshmm_wave_walker_matches_lane_walker(name, monkeypatch):
    """every byte and every statistic -- the ordered f64 sum behind the accuracy mean and its histogram included -- whatever
    share of the tasks the wave walker takes (0: all of them; n: tasks of at least n bases; default: small batches only)"""
    recs = [genome(900_000, 1), genome(400_000, 2)]
    want = run_qs(-1, monkeypatch, recs, **QS_RUNS[name])
    assert sum(len(a) + len(b) for a, b in want[0].values()) > 5_000_000
    for coop in (0, 4096, None):
        got = run_qs(coop, monkeypatch, recs, **QS_RUNS[name])
        assert got[1] == want[1], (name, coop)
        for k in want[0]:
            assert got[0][k] == want[0][k], (name, coop, k)


QS_GOLD = sorted(c for c in CASES if c.startswith("wgs_qshmm"))


@pytest.mark.parametrize("case", QS_GOLD)
@pytest.mark.parametrize("coop", [0, -1, 512])
def test_qshmm_goldens_on_either_walker(case, coop, monkeypatch):
    """the reference's own QSHMM goldens (all three models, ratios, multi-pass, the Q15 hp-del-bias cases, which keep their
    byte-form hp array and therefore the lane walker) with every task / no task / the long tasks on the wave walker"""
    monkeypatch.setenv("PBSIM_COOP_LEN", str(coop))
    outs, _ = product.run_wgs_job(harness.resolve(CASES[case]["args"]), scratch_mb=product.scratch_mb_for(case))
    gold = MANIFEST[f"{case}/philox"]
    for k, v in outs.items():
        assert harness.sha(v) == gold[k]["sha256"], (case, coop, k)


# ---- units drawn from a counter (PBSIM_COOP_DYNAMIC; the default for batches of 16 k - 250 k tasks) ------------------------
@pytest.mark.parametrize("dynamic", ["1", "0"])
def test_counter_drawn_units_change_nothing(dynamic, monkeypatch):
    """both wave walkers with their units drawn from a counter (forced on small batches, where several classes and partial units
    meet) and dealt round-robin: the lane walker's bytes and statistics"""
    monkeypatch.setenv("PBSIM_COOP_DYNAMIC", dynamic)
    recs = [genome(1_500_000, 1), genome(700_000, 2)]
    for name in ("ont_default", "sequel_pass3_short_reads"):
        want = run(-1, monkeypatch, recs, **RUNS[name])
        for coop in (0, 4096):
            got = run(coop, monkeypatch, recs, **RUNS[name])
            assert got == want, (name, coop)
    recs = [genome(900_000, 1), genome(400_000, 2)]
    for name in ("rsii_default", "ont_deletion_heavy"):
        want = run_qs(-1, monkeypatch, recs, **QS_RUNS[name])
        for coop in (0, 4096):
            got = run_qs(coop, monkeypatch, recs, **QS_RUNS[name])
            assert got == want, (name, coop)


def test_counter_drawn_units_at_their_default_size(monkeypatch):
    """22 000 reads as ONE batch (the counter's default range) through the batch primitives: default split with the counter and
    dealt round-robin, against the lane walker"""
    import zlib
    import pbsim3_amd as P
    g = genome(24_000_000, 5)

    def crc(coop, dynamic):
        for k, v in (("PBSIM_COOP_LEN", coop), ("PBSIM_COOP_DYNAMIC", dynamic)):
            if v is None:
                monkeypatch.delenv(k, raising=False)
            else:
                monkeypatch.setenv(k, v)
        p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=3, depth=8.0)
        with P.Context(p, 0) as ctx:
            ctx.set_scratch_bytes(8 << 30)
            ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
            ctx.set_reference(g, 1)
            ctx.batch_walk(1, 22000)
            info = ctx.batch_finalize(0)
            rt, mt = ctx.batch_fetch(info)
            waves = ctx.prof_wave_launches()
        return (zlib.crc32(rt), len(rt), zlib.crc32(mt), len(mt), info.n_final, info.bases), waves

    want, w0 = crc("-1", None)
    assert w0 == 0 and want[4] > 16384              # the quota leaves the batch in the counter's range
    got, w1 = crc(None, None)
    assert w1 > 0 and got == want
    assert crc(None, "0")[0] == want


# ---- the wave walker's grid and the split rule (round 4): knobs of speed only -----------------------------------------------
@pytest.mark.parametrize("knobs", [{"PBSIM_COOP_WG": "1"}, {"PBSIM_COOP_WG": "3"}, {"PBSIM_COOP_WG": "2048"},
                                   {"PBSIM_COOP_SPLIT_READS": "2000"}, {"PBSIM_COOP_SPLIT_READS": "500000", "PBSIM_COOP_WG": "7"}])
def test_grid_size_and_split_rule_change_nothing(knobs, monkeypatch):
    """one persistent workgroup (every unit of every class through the same eight waves), a few, more than the GPU holds at
    once; the split by the rule at two other batch sizes: the lane walker's bytes and statistics"""
    recs = [genome(1_500_000, 1), genome(700_000, 2)]
    want = {name: run(-1, monkeypatch, recs, **RUNS[name]) for name in ("ont_default", "sequel_pass3_short_reads")}
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    for name in want:
        for coop in ((None, 0) if "PBSIM_COOP_SPLIT_READS" in knobs else (0, 4096)):
            assert run(coop, monkeypatch, recs, **RUNS[name]) == want[name], (name, coop, knobs)
