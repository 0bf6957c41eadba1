#!/usr/bin/env python3
"""Generate the small synthetic inputs used by the golden fixtures.

The generated files are committed (tests/golden/inputs/), so the tests never
depend on numpy's bit-generator staying stable.  Run only to regenerate.

quirk.fa      2 FASTA records exercising SURVEY.md quirks: lower-case runs
              (toupper, pbsim.cpp:1035), an N run (hp=1, pbsim.cpp:1050),
              homopolymers of 10/11/12/13 (Q1: hp 11 <-> out-of-bounds bias),
              IUPAC codes (non-ACGT substitution branch, pbsim.cpp:3947), and a
              line longer than BUF_SIZE-1 = 10239 (fgets chunking, :914).
tiny.transcript  12 transcripts in the reference's TSV format (id, plus, minus,
              seq); one sequence > 10239 chars (continuation chunks, :4447),
              lower-case first bases (Q6), minus-strand expression.
"""
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "inputs")


def rand_seq(rng, n):
    return "".join(np.array(list("ACGT"))[rng.integers(0, 4, n)])


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20251002)

    # ---- quirk.fa ----------------------------------------------------------
    parts = [
        rand_seq(rng, 1500),
        rand_seq(rng, 400).lower(),
        "G" + "A" * 10 + "C",
        rand_seq(rng, 300),
        "G" + "A" * 11 + "C",
        rand_seq(rng, 300),
        "C" + "T" * 12 + "G",
        rand_seq(rng, 300),
        "T" + "C" * 13 + "A",
        rand_seq(rng, 200),
        "N" * 30,
        rand_seq(rng, 500),
        "ACGRYACGTNNACGKMACGT",
        rand_seq(rng, 2000),
    ]
    rec1 = "".join(parts)
    long_line = rand_seq(rng, 10300)  # > BUF_SIZE-1
    rec1_tail = rand_seq(rng, 1200)
    rec2 = rand_seq(rng, 9000) + "A" * 23 + rand_seq(rng, 977)
    with open(os.path.join(OUT, "quirk.fa"), "w") as f:
        f.write(">rec1 quirk record with a very long description " + "x" * 150 + "\n")
        for i in range(0, len(rec1), 70):
            f.write(rec1[i:i + 70] + "\n")
        f.write(long_line + "\n")
        for i in range(0, len(rec1_tail), 70):
            f.write(rec1_tail[i:i + 70] + "\n")
        f.write(">rec2\n")
        for i in range(0, len(rec2), 80):
            f.write(rec2[i:i + 80] + "\n")

    # ---- plain.fa: 200 kbp uniform genome (default read-length parameters) --
    s = rand_seq(rng, 200000)
    with open(os.path.join(OUT, "plain.fa"), "w") as f:
        f.write(">chr1\n")
        for i in range(0, len(s), 80):
            f.write(s[i:i + 80] + "\n")

    # ---- tiny.transcript ----------------------------------------------------
    with open(os.path.join(OUT, "tiny.transcript"), "w") as f:
        lens = [346, 812, 1500, 2300, 12690, 999, 1000, 1001, 3100, 450, 5200, 700]
        for i, n in enumerate(lens):
            seq = rand_seq(rng, n)
            if i % 3 == 0:
                seq = seq[:5].lower() + seq[5:]
            if i == 2:
                seq = seq[:700] + "A" * 11 + seq[711:]
            plus = int(rng.integers(0, 6))
            minus = int(rng.integers(0, 3)) if i % 2 == 0 else 0
            if i == 4:
                plus, minus = 4, 2
            f.write(f"TR{i + 1:03d}\t{plus}\t{minus}\t{seq}\n")


if __name__ == "__main__":
    main()
