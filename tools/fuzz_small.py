"""GPU box: WGS on tiny records (100..3000 bp, reads mostly clipped to the record, pbsim.cpp:3804-3809) vs the oracle.
usage: python tools/fuzz_small.py K0 K1"""
import os, random, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import harness, product

bad = 0
k0, k1 = int(sys.argv[1]), int(sys.argv[2])
for k in range(k0, k1):
    r = random.Random(77000 + k)
    qs = r.random() < 0.4
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "g.fa")
        with open(fa, "w") as f:
            for i in range(r.randint(1, 4)):
                n = r.choice([100, 101, 127, 128, 129, 255, 256, 257, r.randint(100, 3000)])
                s = "".join(r.choice("ACGT") for _ in range(n))
                f.write(">r%d\n%s\n" % (i, s))
        model = r.choice(["QSHMM-RSII", "QSHMM-ONT"] if qs else ["ERRHMM-RSII", "ERRHMM-SEQUEL", "ERRHMM-ONT", "ERRHMM-ONT-HQ"])
        mean = r.randint(150, 3000)
        args = ["--strategy", "wgs", "--method", "qshmm" if qs else "errhmm", "--qshmm" if qs else "--errhmm",
                "MODEL:%s.model" % model, "--genome", fa, "--depth", str(round(r.uniform(0.5, 40), 2)),
                "--seed", str(r.randint(0, 2**31 - 1)), "--length-mean", str(mean),
                "--length-sd", str(int(mean * r.uniform(0.2, 1.0))), "--length-min", str(r.randint(60, 100)),
                "--pass-num", str(r.choice([1, 1, 3])), "--hp-del-bias", r.choice(["1", "6"])]
        try:
            want = harness.run_oracle(args, "philox", td)
        except RuntimeError as e:
            print(k, "oracle refused:", str(e)[-80:].replace("\n", " "))
            continue
        try:
            outs, _ = product.run_wgs(harness.resolve(args))
        except Exception as e:
            print(k, "PRODUCT FAILED", e, args)
            bad += 1
            continue
        for key, v in outs.items():
            if v != want[key]:
                print(k, "MISMATCH", key, args)
                bad += 1
                break
print("swept", k1 - k0, "cases,", bad, "bad")
sys.exit(1 if bad else 0)
