#!/usr/bin/env python3
"""Synthetic FIC-HMM model files for the parity cases that no shipped model reaches (missing 4).

Every model under /root/reference/data has initial / transition rows that sum to 1 within rounding, so the reference's
cumulative tables end at exactly 1000 (ERRHMM, pbsim.cpp:3715-3789) resp. 100 (QSHMM, :2066-2142) and its draws are
`rand() % 1000` / `% 100` everywhere.  The table builders accept any row: `err_rand_value_{init,tran}` and
`qc_rand_value_{init,emis,tran}` are whatever the cumulative sum rounds to.  These files make the other moduli and the
state counts beyond the wave walkers' limits real:

  SYNTH-ERRHMM-MOD.model     classes 66..92, 3-12 states, IP / TP rows scaled to 1.0 / 0.995 / 0.99 / 0.97 / 0.9
  SYNTH-ERRHMM-S35.model     classes 66..92, rows sum to 1; class 84 has 35 states, class 85 has 50 (= STATE_MAX, pbsim.cpp:43)
  SYNTH-QSHMM-MOD.model      classes 66..92, 4-20 states, 24 quality codes, IP / EP / TP rows scaled likewise

The files are written in the shipped models' own format (`<acc> IP|EP|TP <state> <values %.3e>`, set_errhmm :5640,
set_qshmm :5570) and committed gzip-compressed next to them; the goldens come from running the reference on them
(make_golden.py).  Run only to regenerate."""
import gzip
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "models")
SCALES = [1.0, 0.995, 0.99, 0.97, 0.9]


def sticky_row(rng, n, j, stay):
    """a transition row over n states that stays in state j with probability `stay` (the shipped models are sticky)"""
    w = [rng.random() ** 3 for _ in range(n)]
    w[j] = 0.0
    s = sum(w) or 1.0
    row = [(1.0 - stay) * x / s for x in w]
    row[j] = stay if n > 1 else 1.0
    # a few exact zeros like the real files have tiny entries: the table builders skip `== 0` columns
    for k in range(n):
        if k != j and rng.random() < 0.2:
            row[j] += row[k]
            row[k] = 0.0
    return row


def fmt(vals):
    return " ".join("%.3e" % v for v in vals)


def errhmm(path, seed, states_of, scaled):
    rng = random.Random(seed)
    lines = []
    for acc in range(66, 93):
        n = states_of(acc, rng)
        ip = [rng.random() ** 2 for _ in range(n)]
        s = sum(ip)
        f = rng.choice(SCALES) if scaled else 1.0
        ip = [f * x / s for x in ip]
        for j in range(n):
            lines.append("%d IP %d %.3e" % (acc, j + 1, ip[j]))
        for j in range(n):
            # match | substitution | insertion (the cumulative table, :3741-3762) and the deletion threshold (:3740)
            err = (100 - acc) / 100.0
            sub, ins, dele = (err * rng.uniform(0.02, 0.3), err * rng.uniform(0.2, 1.2), err * rng.uniform(0.1, 0.9))
            if rng.random() < 0.15:
                sub = 0.0
            match = max(0.05, 1.0 - sub - ins - dele)
            lines.append("%d EP %d %s" % (acc, j + 1, fmt([match, sub, ins, dele])))
        for j in range(n):
            f = rng.choice(SCALES) if scaled else 1.0
            row = [f * x for x in sticky_row(rng, n, j, rng.uniform(0.5, 0.95))]
            lines.append("%d TP %d %s" % (acc, j + 1, fmt(row)))
    with gzip.GzipFile(path, "wb", mtime=0) as g:
        g.write(("\n".join(lines) + "\n").encode())


def qshmm(path, seed):
    rng = random.Random(seed)
    lines = []
    nq = 24
    for acc in range(66, 93):
        n = 4 + (acc * 5) % 17
        ip = [rng.random() ** 2 for _ in range(n)]
        s = sum(ip)
        f = rng.choice(SCALES)
        for j in range(n):
            lines.append("%d IP %d %.3e" % (acc, j + 1, f * ip[j] / s))
        for j in range(n):
            centre = rng.uniform(2, nq - 3)
            w = [pow(2.718281828, -((q - centre) / rng.uniform(1.0, 4.0)) ** 2) for q in range(nq)]
            w[0] = 0.0  # quality 0 ('!') carries error probability 1: keep the reads non-degenerate
            s = sum(w)
            f = rng.choice(SCALES)
            lines.append("%d EP %d %s" % (acc, j + 1, fmt([f * x / s for x in w])))
        for j in range(n):
            f = rng.choice(SCALES)
            row = [f * x for x in sticky_row(rng, n, j, rng.uniform(0.4, 0.9))]
            lines.append("%d TP %d %s" % (acc, j + 1, fmt(row)))
    with gzip.GzipFile(path, "wb", mtime=0) as g:
        g.write(("\n".join(lines) + "\n").encode())


def main():
    errhmm(os.path.join(OUT, "SYNTH-ERRHMM-MOD.model.gz"), 41, lambda acc, rng: 3 + (acc * 7) % 10, True)
    errhmm(os.path.join(OUT, "SYNTH-ERRHMM-S35.model.gz"), 42,
           lambda acc, rng: 35 if acc == 84 else 50 if acc == 85 else 3 + (acc * 7) % 10, False)
    qshmm(os.path.join(OUT, "SYNTH-QSHMM-MOD.model.gz"), 43)


if __name__ == "__main__":
    main()
