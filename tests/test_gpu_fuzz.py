"""Seeded random sweep over the parameter space (GPU): product vs oracle, byte for byte.
Parameters stay inside the region where the reference itself is well defined (reads of
at least ~60 bases: shorter ones can drive its freq_accuracy[] index negative)."""
import numpy as np
import pytest

import harness
import product

pytestmark = pytest.mark.gpu

ERR = ["ERRHMM-RSII", "ERRHMM-SEQUEL", "ERRHMM-ONT", "ERRHMM-ONT-HQ"]
QS = ["QSHMM-RSII", "QSHMM-ONT"]


def make_case(k):
    rng = np.random.default_rng(1000 + k)
    n_rec = int(rng.integers(1, 4))
    recs = []
    for _ in range(n_rec):
        n = int(rng.integers(3000, 40000))
        s = np.array(list("ACGT"))[rng.integers(0, 4, n)]
        for _ in range(int(rng.integers(0, 6))):           # homopolymer / N / IUPAC / lower-case patches
            p, ln = int(rng.integers(0, n - 40)), int(rng.integers(2, 30))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                s[p:p + ln] = rng.choice(list("ACGT"))
            elif kind == 1:
                s[p:p + ln] = "N"
            elif kind == 2:
                s[p:p + ln] = rng.choice(list("RYKMSW"), ln)
            else:
                s[p:p + ln] = np.char.lower(s[p:p + ln])
        recs.append("".join(s))
    qs = bool(rng.integers(0, 3) == 0)
    model = str(rng.choice(QS if qs else ERR))
    mean = int(rng.integers(300, 3000))
    args = ["--strategy", "wgs", "--method", "qshmm" if qs else "errhmm", "--qshmm" if qs else "--errhmm",
            f"MODEL:{model}.model", "--depth", str(round(float(rng.uniform(1.0, 6.0)), 2)),
            "--seed", str(int(rng.integers(0, 2**31 - 1))),
            "--length-mean", str(mean), "--length-sd", str(int(mean * rng.uniform(0.3, 1.2))),
            "--length-min", str(int(rng.integers(60, 200))), "--length-max", str(int(rng.integers(5000, 100000))),
            "--accuracy-mean", str(round(float(rng.uniform(0.72, 0.95 if qs else 0.99)), 2)),
            "--pass-num", str(int(rng.choice([1, 1, 1, 2, 4]))),
            "--hp-del-bias", str(rng.choice(["1", "1", "2.5", "7"])),
            "--id-prefix", str(rng.choice(["S", "read", "x.Y_"]))]
    if qs:
        args += ["--difference-ratio", "%d:%d:%d" % tuple(int(x) for x in rng.integers(1, 60, 3))]
    return recs, args


@pytest.mark.parametrize("k", range(16))
def test_random_configuration_matches_oracle(k, tmp_path):
    recs, args = make_case(k)
    fa = tmp_path / "g.fa"
    with open(fa, "w") as f:
        for i, s in enumerate(recs, 1):
            f.write(f">rec{i} fuzz\n")
            w = 50 + 10 * i
            for p in range(0, len(s), w):
                f.write(s[p:p + w] + "\n")
    args = args + ["--genome", str(fa)]
    outs, _ = product.run_wgs(harness.resolve(args))
    want = harness.run_oracle(args, "philox", str(tmp_path))
    for key, v in outs.items():
        assert v == want[key], (k, key, args)
