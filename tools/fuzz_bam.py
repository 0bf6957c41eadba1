"""GPU box: random multi-pass runs through the CLI's native BAM output (GPU records + GPU BGZF) vs the oracle's SAM text.
usage: python tools/fuzz_bam.py K0 K1"""
import os, random, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import harness
from test_gpu_bam import compare_bam_with_sam

CLI = os.path.join(R, "pbsim3_amd", "bin", "pbsim")
bad = 0
k0, k1 = int(sys.argv[1]), int(sys.argv[2])
for k in range(k0, k1):
    r = random.Random(31000 + k)
    qs = r.random() < 0.5
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(td + "/o"), os.makedirs(td + "/p")
        fa = td + "/g.fa"
        with open(fa, "w") as f:
            for i in range(r.randint(1, 2)):
                n = r.choice([r.randint(2000, 20000), r.randint(80000, 150000)])
                s = "".join(r.choice("ACGT") for _ in range(n))
                f.write(">c%d\n" % i)
                for p in range(0, n, 80):
                    f.write(s[p:p + 80] + "\n")
        model = r.choice(["QSHMM-RSII", "QSHMM-ONT"] if qs else ["ERRHMM-RSII", "ERRHMM-SEQUEL", "ERRHMM-ONT"])
        mean = r.choice([r.randint(150, 1500), r.randint(40000, 90000)])
        args = ["--strategy", "wgs", "--method", "qshmm" if qs else "errhmm", "--qshmm" if qs else "--errhmm",
                "MODEL:%s.model" % model, "--genome", fa, "--depth", str(round(r.uniform(0.5, 6), 2)),
                "--seed", str(r.randint(0, 2**31 - 1)), "--length-mean", str(mean), "--length-sd", str(int(mean * 0.5)),
                "--pass-num", str(r.randint(2, 6)), "--accuracy-mean", str(round(r.uniform(0.76, 0.94), 2)),
                "--id-prefix", r.choice(["S", "movie_1", "m54006"])]
        try:
            want = harness.run_oracle(args, "philox", td + "/o")
        except RuntimeError as e:
            print(k, "oracle refused:", str(e)[-80:].replace("\n", " "))
            continue
        p = subprocess.run([CLI] + harness.resolve(args) + ["--prefix", td + "/p/out"], capture_output=True, text=True)
        if p.returncode != 0:
            print(k, "CLI FAILED", p.stderr[-200:], args)
            bad += 1
            continue
        try:
            for key in sorted(x for x in want if x.endswith(".sam")):
                compare_bam_with_sam(open(td + "/p/out" + key[:-4] + ".bam", "rb").read(), want[key])
        except AssertionError as e:
            print(k, "MISMATCH", str(e)[:200], args)
            bad += 1
print("swept", k1 - k0, "cases,", bad, "bad")
sys.exit(1 if bad else 0)
