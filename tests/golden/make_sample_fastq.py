#!/usr/bin/env python3
"""inputs/sample.fastq: the FASTQ the `--method sample` goldens sample from (the reference ships none).

153 reads: lengths 40..2 600 plus one of 10 400 (> BUF_SIZE-1 = 10 239: the chunked fgets path of get_sample_inf,
pbsim.cpp:1273-1281), per-read quality levels Q3..Q40 with jitter so that some reads fall below --accuracy-min 0.75
and some below --length-min 100 (the filter of pbsim.cpp:1252-1260).  The bases are irrelevant to the simulation
(only the quality strings are sampled) but present, as in a real FASTQ.  Deterministic: random.Random(7)."""
import os
import random

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    r = random.Random(7)
    lens = [r.choice([40, 60, 90]) for _ in range(6)]
    lens += [int(r.lognormvariate(6.6, 0.7)) + 100 for _ in range(146)]
    lens = [min(x, 2600) for x in lens]
    lens.insert(57, 10400)
    with open(os.path.join(HERE, "inputs", "sample.fastq"), "w") as f:
        for i, n in enumerate(lens):
            level = r.choice([3, 5, 8, 10, 12, 15, 20, 25, 30, 35, 40])
            qual = "".join(chr(33 + max(0, min(93, level + r.randint(-5, 5)))) for _ in range(n))
            seq = "".join(r.choice("ACGT") for _ in range(n))
            f.write("@sample_%d\n%s\n+\n%s\n" % (i + 1, seq, qual))


if __name__ == "__main__":
    main()
