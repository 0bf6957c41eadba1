"""GPU box: random sweep of the sampling method ON SEVERAL RANKS (pbsim --devices 0,0,..: 2-5 contexts on the one GPU, host
communicator; pbsim_simulate_sample_comm) vs the oracle: every output file and the stderr report, byte for byte; every fourth
case with the default GPU compression (members inflated).   usage: python tools/fuzz_sample_multi.py K0 K1"""
import gzip, os, random, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import harness

CLI = os.path.join(R, "pbsim3_amd", "bin", "pbsim")
bad = 0
k0, k1 = int(sys.argv[1]), int(sys.argv[2])
for k in range(k0, k1):
    r = random.Random(91000 + k)
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(td + "/o"), os.makedirs(td + "/p")
        fq = os.path.join(td, "s.fastq")
        with open(fq, "w") as f:
            for i in range(r.choice([r.randint(3, 40), r.randint(40, 900)])):
                n = int(r.lognormvariate(r.uniform(4.5, 7.5), 0.8)) + r.choice([1, 30, 101])
                n = min(n, 20000)
                level = r.choice([2, 6, 9, 12, 17, 25, 33, 41])
                q = "".join(chr(33 + max(0, min(93, level + r.randint(-6, 6)))) for _ in range(n))
                f.write("@r%d\n%s\n+\n%s\n" % (i, "A" * n, q))
        fa = os.path.join(td, "g.fa")
        with open(fa, "w") as f:
            for i in range(r.randint(1, 3)):
                n = r.choice([r.randint(400, 60000), r.randint(60000, 600000)])
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
            want = harness.run_oracle(args, "philox", td + "/o")
        except RuntimeError as e:
            print(k, "oracle refused:", str(e)[-90:].replace("\n", " "))
            continue
        ranks = r.randint(2, 5)
        zipped = k % 4 == 3
        env = dict(os.environ)
        if r.random() < 0.5:
            env["PBSIM_SCRATCH_MB"] = str(r.choice([64, 256, 1024]))
        cmd = [CLI] + args + ["--prefix", td + "/p/out", "--devices", ",".join(["0"] * ranks)] + ([] if zipped else ["--no-gzip"])
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, cwd=td + "/p", env=env, timeout=120)
        except subprocess.TimeoutExpired:
            print(k, "TIMEOUT", args, "ranks", ranks)
            bad += 1
            continue
        if p.returncode != 0:
            if "do not fit a rank's scratch pool" in p.stderr or "scratch pool too small" in p.stderr:
                print(k, "pool too small for one string's copies, skipped")
                continue
            print(k, "CLI FAILED rc", p.returncode, p.stderr[-1200:], args, ranks, env.get("PBSIM_SCRATCH_MB"))
            bad += 1
            continue
        if zipped:
            for fn in os.listdir(td + "/p"):
                if fn.endswith(".gz"):
                    with open(td + "/p/" + fn, "rb") as f:
                        data = gzip.decompress(f.read())
                    with open(td + "/p/" + fn[:-3], "wb") as f:
                        f.write(data)
                    os.remove(td + "/p/" + fn)
        got = harness.collect(td + "/p")
        got[".stderr"] = harness.strip_report(p.stderr).encode()
        if sorted(got) != sorted(want) or any(got[x] != want[x] for x in got):
            diff = [x for x in want if got.get(x) != want[x]]
            print(k, "MISMATCH", diff, args, "ranks", ranks, env.get("PBSIM_SCRATCH_MB"))
            bad += 1
print("swept", k1 - k0, "cases,", bad, "bad")
sys.exit(1 if bad else 0)
