"""GPU box: random sweep of the sampling method (product through the ABI vs oracle, byte for byte).
usage: python tools/fuzz_sample.py K0 K1"""
import os, random, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import harness
from test_gpu_sample import run_product

bad = 0
k0, k1 = int(sys.argv[1]), int(sys.argv[2])
for k in range(k0, k1):
    r = random.Random(5000 + k)
    with tempfile.TemporaryDirectory() as td:
        fq = os.path.join(td, "s.fastq")
        with open(fq, "w") as f:
            for i in range(r.randint(3, 150)):
                n = int(r.lognormvariate(r.uniform(4.5, 7.5), 0.8)) + r.choice([1, 30, 101])
                n = min(n, 30000)
                level = r.choice([2, 6, 9, 12, 17, 25, 33, 41])
                q = "".join(chr(33 + max(0, min(93, level + r.randint(-6, 6)))) for _ in range(n))
                f.write("@r%d\n%s\n+\n%s\n" % (i, "A" * n, q))
        fa = os.path.join(td, "g.fa")
        with open(fa, "w") as f:
            for i in range(r.randint(1, 3)):
                n = r.randint(400, 60000)
                s = "".join(r.choice("ACGT") for _ in range(n))
                if r.random() < 0.5:
                    p = r.randint(0, n - 30)
                    s = s[:p] + r.choice("ACGTN") * r.randint(5, 25) + s[p + 25:]
                f.write(">rec%d\n" % (i + 1))
                for p in range(0, len(s), 70):
                    f.write(s[p:p + 70] + "\n")
        args = ["--strategy", "wgs", "--method", "sample", "--sample", fq, "--genome", fa,
                "--depth", str(round(r.uniform(0.3, 25.0), 2)), "--seed", str(r.randint(0, 2**31 - 1)),
                "--difference-ratio", "%d:%d:%d" % (r.randint(1, 60), r.randint(1, 60), r.randint(1, 60)),
                "--hp-del-bias", r.choice(["1", "1", "3", "8.5"]), "--length-min", str(r.choice([100, 30, 250])),
                "--length-max", str(r.choice([1000000, 5000, 20000])), "--accuracy-min", r.choice(["0.75", "0.5", "0.9"]),
                "--id-prefix", r.choice(["S", "smp_"])]
        try:
            want = harness.run_oracle(args, "philox", td)
        except RuntimeError as e:
            print(k, "oracle refused:", str(e)[-90:].replace("\n", " "))
            continue
        split = r.choice([None, None, "-1", "0", "300", "2000"])  # which strings get a wave each (scoop_walk_string), which a lane
        if split is None:
            os.environ.pop("PBSIM_COOP_LEN", None)
        else:
            os.environ["PBSIM_COOP_LEN"] = split
        try:
            got = run_product(args, r.choice([None, 16, 64]))
        except Exception as e:
            print(k, "PRODUCT FAILED", e, args)
            bad += 1
            continue
        for key, v in got.items():
            if v != want[key]:
                print(k, "MISMATCH", key, len(v), len(want[key]), "split", split, args)
                bad += 1
                break
print("swept", k1 - k0, "cases,", bad, "bad")
sys.exit(1 if bad else 0)
