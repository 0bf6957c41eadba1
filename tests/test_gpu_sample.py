"""The sampling method (--method sample, pbsim.cpp:1694-1949) through the C ABI against the oracle on the same seeded
inputs: chunking under a small scratch pool (chunks must not change a byte), chains of copies, strings longer than the
record, deletion-heavy ratios (every copy shorter than the last), --hp-del-bias, many sweeps over a tiny profile."""
import os
import random

import pytest

import harness
import pbsim3_amd as P
from pbsim3_amd import args as A

pytestmark = pytest.mark.gpu
INPUTS = os.path.join(harness.GOLDEN, "inputs")


@pytest.fixture(autouse=True, params=["default", "-1", "0", "1500"])
def walker_split(request, monkeypatch):
    """which strings get a whole wave (scoop_walk_string, round 3) and which a lane: default = 0 = every string; -1 none;
    1500 = the line waves whose strings all have >= 1500 characters.  Same bytes every time."""
    if request.param == "default":
        monkeypatch.delenv("PBSIM_COOP_LEN", raising=False)
    else:
        monkeypatch.setenv("PBSIM_COOP_LEN", request.param)
    return request.param


def run_product(argv, scratch_mb=None):
    p, a = A.parse(argv)
    quals = A.read_sample_fastq(a["--sample"], p.len_min, p.len_max,
                                int(float(a.get("--accuracy-min", 0.75)) * 100) * 0.01 if "--accuracy-min" in a else 0.75,
                                int(float(a.get("--accuracy-max", 1.0)) * 100) * 0.01 if "--accuracy-max" in a else 1.0)
    outs = {}
    with P.Context(p, 0) as ctx:
        if scratch_mb:
            ctx.set_scratch_bytes(scratch_mb << 20)
        ctx.set_sample_profile(quals)
        recs = A.read_fasta(a["--genome"])[0]
        if p.hp_del_bias != 1:
            for r in recs:
                ctx.add_hp_census(r)
            ctx.finish_hp_census()
        for i, r in enumerate(recs, 1):
            ctx.set_reference(r, i)
            rt, mt = ctx.simulate_sample()
            outs["_%04d.fq" % i] = rt
            outs["_%04d.maf" % i] = mt
    return outs


def check(argv, tmp_path, scratch_mb=None):
    want = harness.run_oracle(argv, "philox", str(tmp_path))
    got = run_product(harness.resolve(argv), scratch_mb)
    for k, v in got.items():
        if v != want[k]:
            n = next((i for i, (x, y) in enumerate(zip(v, want[k])) if x != y), min(len(v), len(want[k])))
            raise AssertionError(f"{k}: differs at byte {n} (sizes {len(v)} vs {len(want[k])}):\n"
                                 f"  got  {v[max(0, n - 80):n + 60]!r}\n  want {want[k][max(0, n - 80):n + 60]!r}")
    return got


BASE = ["--strategy", "wgs", "--method", "sample", "--sample", "INPUT:sample.fastq"]


@pytest.mark.parametrize("scratch_mb", [8, 24, 96])
def test_chunking_does_not_change_bytes(tmp_path, scratch_mb):
    check(BASE + ["--genome", "INPUT:plain.fa", "--depth", "3", "--seed", "11"], tmp_path, scratch_mb)


def test_many_copies_per_string(tmp_path):
    """depth 40 on a 200 kbp record: ~60 copies per string, deletion-heavy so that every copy shrinks"""
    check(BASE + ["--genome", "INPUT:plain.fa", "--depth", "40", "--seed", "12", "--difference-ratio", "5:15:80"], tmp_path, 24)


def test_insertion_heavy_stops_on_the_quality_string(tmp_path):
    """ins >> del: the read ends when the quality string is used up, ref span < string length (pbsim.cpp:1776, 1847)"""
    check(BASE + ["--genome", "INPUT:quirk.fa", "--depth", "6", "--seed", "13", "--difference-ratio", "5:90:5",
                  "--hp-del-bias", "4"], tmp_path)


def test_tiny_profile_many_sweeps(tmp_path):
    """4 short strings against a quota of 30x: one long chain per string, then residue sweeps"""
    r = random.Random(3)
    fq = tmp_path / "tiny.fastq"
    with open(fq, "w") as f:
        for i, n in enumerate([300, 450, 800, 1200]):
            q = "".join(chr(33 + r.randint(8, 30)) for _ in range(n))
            f.write("@r%d\n%s\n+\n%s\n" % (i, "A" * n, q))
    argv = ["--strategy", "wgs", "--method", "sample", "--sample", str(fq), "--genome", "INPUT:quirk.fa",
            "--depth", "30", "--seed", "14"]
    check(argv, tmp_path, 16)


def test_high_quality_profile(tmp_path):
    """Q40 strings: a handful of errors per read, deletion/insertion balance decides each chain link"""
    r = random.Random(4)
    fq = tmp_path / "hifi.fastq"
    with open(fq, "w") as f:
        for i in range(40):
            n = r.randint(500, 4000)
            q = "".join(chr(33 + r.randint(35, 45)) for _ in range(n))
            f.write("@h%d\n%s\n+\n%s\n" % (i, "C" * n, q))
    argv = ["--strategy", "wgs", "--method", "sample", "--sample", str(fq), "--genome", "INPUT:plain.fa",
            "--depth", "8", "--seed", "15", "--difference-ratio", "20:30:50"]
    check(argv, tmp_path)


def test_large_profile_default_split(tmp_path, walker_split):
    """30 000 strings, 470 line waves, a second sweep over every nth string: all on waves (default), all on lanes, and the
    line waves of >= 1500 characters on waves beside the others on lanes in one launch"""
    if walker_split == "0":
        pytest.skip("0 is the default")
    import numpy as np
    rng = np.random.default_rng(5)
    n = 30000
    lens = np.clip(rng.gamma(1.6, 250.0, n), 40, 6000).astype(int)
    fq = tmp_path / "big.fastq"
    with open(fq, "w") as f:
        for i in range(n):
            q = bytes((33 + np.clip(rng.integers(5, 35) + rng.integers(-4, 5, lens[i]), 0, 60)).astype(np.uint8)).decode()
            f.write("@b%d\n%s\n+\n%s\n" % (i, "G" * lens[i], q))
    argv = ["--strategy", "wgs", "--method", "sample", "--sample", str(fq), "--genome", "INPUT:plain.fa",
            "--depth", "85", "--seed", "16", "--length-min", "40"]
    check(argv, tmp_path)


def test_wave_and_lane_walkers_agree_at_scale(monkeypatch, walker_split):
    """40 000 strings with a long tail (up to 60 000 characters), ~330 Mbases, two sweeps: the bytes and the statistics of the
    wave walker (default) equal the lane walker's, which the cases above pin to the oracle; the oracle would need half a
    minute for this one.  Also under --hp-del-bias (the wave walker's third ring)."""
    if walker_split != "default":
        pytest.skip("runs its own splits")
    import zlib
    import numpy as np
    rng = np.random.default_rng(21)
    n = 40_000
    k = (7500.0 / 6500.0) ** 2
    lens = np.clip(rng.gamma(k, 7500.0 / k, n), 100, 60000).astype(np.int64)
    level = rng.integers(6, 32, n)
    noise = rng.integers(-5, 6, int(lens.sum()), dtype=np.int8)
    quals, o = [], 0
    for i in range(n):
        quals.append((np.clip(level[i] + noise[o:o + lens[i]], 0, 93).astype(np.uint8) + 33).tobytes())
        o += int(lens[i])
    g = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 20_000_000)].copy()
    for p in rng.integers(0, len(g) - 40, 4000):   # homopolymers past the hp == 11 class, some non-ACGT bases
        g[p:p + int(rng.integers(8, 30))] = g[p]
    for p in rng.integers(0, len(g) - 8, 1500):
        g[p:p + int(rng.integers(1, 6))] = ord("N")
    genome = g.tobytes()

    def run(split, bias):
        if split is None:
            monkeypatch.delenv("PBSIM_COOP_LEN", raising=False)
        else:
            monkeypatch.setenv("PBSIM_COOP_LEN", split)
        p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_SAMPLE, seed=9, depth=16.5, hp_del_bias=bias)
        with P.Context(p, 0) as ctx:
            ctx.set_scratch_bytes(6 << 30)
            ctx.set_sample_profile(quals)
            if bias != 1:
                ctx.add_hp_census(genome)
                ctx.finish_hp_census()
            ctx.set_reference(genome, 1)
            rt, mt = ctx.simulate_sample()
            st = ctx.stats()
            stats = tuple(getattr(st, f[0]) for f in st._fields_)
        return zlib.crc32(rt), len(rt), zlib.crc32(mt), len(mt), stats

    for bias in (1.0, 3.0):
        want = run("-1", bias)
        assert want[1] > 300_000_000 and want[4][0] > 0
        assert run(None, bias) == want, bias
        assert run("9000", bias) == want, bias


def test_strings_of_one_to_five_characters(tmp_path):
    """--length-min 1: strings shorter than a group of four columns, next to ordinary ones and to strings longer than the record"""
    r = random.Random(6)
    fq = tmp_path / "short.fastq"
    with open(fq, "w") as f:
        for i, n in enumerate([1, 2, 3, 4, 5, 1, 700, 2, 63, 64, 65, 127, 128, 129, 30000, 3, 255, 256, 257]):
            q = "".join(chr(33 + r.randint(4, 40)) for _ in range(n))
            f.write("@s%d\n%s\n+\n%s\n" % (i, "T" * n, q))
    argv = ["--strategy", "wgs", "--method", "sample", "--sample", str(fq), "--genome", "INPUT:quirk.fa",
            "--depth", "9", "--seed", "17", "--length-min", "1", "--accuracy-min", "0.5"]
    check(argv, tmp_path)


def test_second_slot_is_optional(tmp_path, walker_split):
    """the driver sets the next chunk walking on a second slot while the current one is emitted -- unless a second scratch pool
    does not fit the GPU's memory (here: a 150 GiB pool): then the chunks follow each other on one slot, same bytes"""
    if walker_split != "default":
        pytest.skip("one split is enough")
    r = random.Random(8)
    fq = tmp_path / "few.fastq"
    with open(fq, "w") as f:
        for i, n in enumerate([400, 900, 1500, 700, 2500, 350]):
            q = "".join(chr(33 + r.randint(8, 30)) for _ in range(n))
            f.write("@r%d\n%s\n+\n%s\n" % (i, "A" * n, q))
    argv = ["--strategy", "wgs", "--method", "sample", "--sample", str(fq), "--genome", "INPUT:quirk.fa",
            "--depth", "25", "--seed", "19"]
    small = check(argv, tmp_path, 64)
    big = check(argv, tmp_path, 150 * 1024)
    assert small == big
