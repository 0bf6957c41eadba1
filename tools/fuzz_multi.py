"""GPU box: random sweep through the `pbsim` CLI on SEVERAL RANKS (--devices 0,0,..: 1-6 contexts on the one GPU, host
communicator) with small scratch pools (many rounds per record: cuts inside any rank's block, top-up rounds, tails on any rank,
records overlapping in the pipeline, records split over several jobs) vs the oracle: every output file and the stderr report,
byte for byte; every fifth case with the default GPU compression (members inflated).   usage: python tools/fuzz_multi.py K0 K1"""
import gzip, os, random, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import harness

CLI = os.path.join(R, "pbsim3_amd", "bin", "pbsim")
ERR = ["ERRHMM-RSII", "ERRHMM-SEQUEL", "ERRHMM-ONT", "ERRHMM-ONT-HQ"]
QS = ["QSHMM-RSII", "QSHMM-ONT", "QSHMM-ONT-HQ"]


def seq(r, n):
    s = "".join(r.choice("ACGT") for _ in range(n))
    if r.random() < 0.3:
        s = s[:3].lower() + s[3:]
    if r.random() < 0.3 and n > 60:
        p = r.randint(0, n - 40)
        s = s[:p] + r.choice("ACGTN") * r.randint(8, 20) + s[p + 20:]
    return s


bad = 0
k0, k1 = int(sys.argv[1]), int(sys.argv[2])
for k in range(k0, k1):
    r = random.Random(77000 + k)
    strategy = r.choice(["trans", "templ", "wgs"])
    qs = r.random() < 0.4
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(td + "/o"), os.makedirs(td + "/p")
        if strategy == "trans":
            path = td + "/t.tsv"
            with open(path, "w") as f:
                for i in range(r.randint(1, 25)):
                    n = r.choice([r.randint(150, 3000), r.randint(3000, 14000)])
                    f.write("T%d\t%d\t%d\t%s\n" % (i, r.randint(0, 6), r.randint(0, 3), seq(r, n)))
            inp = ["--transcript", path]
        elif strategy == "templ":
            path = td + "/t.fa"
            with open(path, "w") as f:
                for i in range(r.randint(1, 25)):
                    s = seq(r, r.randint(120, 9000))
                    f.write(">tp%d some text\n" % i)
                    w = r.choice([60, 80, 20000])
                    for p in range(0, len(s), w):
                        f.write(s[p:p + w] + "\n")
            inp = ["--template", path]
        else:
            path = td + "/g.fa"
            with open(path, "w") as f:
                for i in range(r.randint(1, 4)):
                    s = seq(r, r.choice([r.randint(2000, 30000), r.randint(30000, 400000)]))
                    f.write(">chr%d\n" % i)
                    for p in range(0, len(s), 70):
                        f.write(s[p:p + 70] + "\n")
            inp = ["--genome", path, "--depth", str(round(r.uniform(0.5, 8), 2))]
        model = r.choice(QS if qs else ERR)
        mean = r.randint(300, 2500)
        args = ["--strategy", strategy, "--method", "qshmm" if qs else "errhmm", "--qshmm" if qs else "--errhmm",
                "MODEL:%s.model" % model] + inp + ["--seed", str(r.randint(0, 2**31 - 1)),
                "--pass-num", str(r.choice([1, 1, 2, 3])), "--hp-del-bias", r.choice(["1", "1", "4"]),
                "--accuracy-mean", str(round(r.uniform(0.75, 0.95 if qs else 0.98), 2))]
        if strategy != "templ":
            args += ["--length-mean", str(mean), "--length-sd", str(int(mean * r.uniform(0.3, 1.1))),
                     "--length-min", str(r.randint(60, 150))]
        if qs:
            args += ["--difference-ratio", "%d:%d:%d" % (r.randint(1, 60), r.randint(1, 60), r.randint(1, 60))]
        try:
            want = harness.run_oracle(args, "philox", td + "/o")
        except RuntimeError as e:
            print(k, "oracle refused:", str(e)[-90:].replace("\n", " "))
            continue
        ranks = r.randint(1, 6)
        if os.environ.get("FUZZ_RANKS"):
            ranks = int(os.environ["FUZZ_RANKS"])
        zipped = k % 5 == 4
        env = dict(os.environ, PBSIM_SCRATCH_MB=str(r.choice([3, 4, 6, 12, 48])))
        if r.random() < 0.25:
            env["PBSIM_JOB_REF_GB"] = "0.0001"   # ~50 kbases of reference per job: the genome runs as several jobs
        if os.environ.get("FUZZ_TORCHRUN"):   # the same job as one process per rank under torchrun (torch.distributed gloo communicator)
            ranks = max(2, min(ranks, 3))
            env["PYTHONPATH"] = R
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
                   "127.0.0.1", "--master-port", str(29600 + k % 300), "-m", "pbsim3_amd.run_multi"] + harness.resolve(args) + \
                  ["--prefix", td + "/p/out", "--backend", "gloo", "--one-gpu"] + ([] if zipped else ["--no-gzip"])
        else:
            cmd = [CLI] + harness.resolve(args) + ["--prefix", td + "/p/out", "--devices", ",".join(["0"] * ranks)] + \
                  ([] if zipped else ["--no-gzip"])
        p = subprocess.run(cmd, capture_output=True, text=True, cwd=td + "/p", env=env)
        if os.environ.get("FUZZ_TORCHRUN") and ":::: Simulation parameters" in p.stderr:   # torchrun's banner comes first
            p.stderr = "\n".join(l for l in p.stderr[p.stderr.index(":::: Simulation parameters"):].splitlines()
                                 if "amdgpu.ids" not in l and not l.startswith(("W0", "W1", "[W", "[E", "[Gloo", "[rank")))
        if p.returncode != 0:
            if "scratch budget exceeded" in p.stderr or "scratch pool too small" in p.stderr:
                print(k, "pool too small for a single read, skipped")
                continue
            print(k, "CLI FAILED rc", p.returncode, p.stderr[-1200:], args, ranks, env.get("PBSIM_SCRATCH_MB"), env.get("PBSIM_JOB_REF_GB"))
            bad += 1
            continue
        if zipped:       # inflate the members; BAM records are compared by tests/test_gpu_bam.py, here only their presence
            for fn in os.listdir(td + "/p"):
                if fn.endswith(".gz"):
                    with open(td + "/p/" + fn, "rb") as f:
                        data = gzip.decompress(f.read())
                    with open(td + "/p/" + fn[:-3], "wb") as f:
                        f.write(data)
                    os.remove(td + "/p/" + fn)
        got = harness.collect(td + "/p")
        got[".stderr"] = harness.strip_report(p.stderr).encode()
        if zipped and any(x.endswith(".sam") for x in want):
            for x in list(want):
                if x.endswith(".sam"):
                    want.pop(x)
                    got.pop(x, None)
        if sorted(got) != sorted(want) or any(got[x] != want[x] for x in got):
            diff = [x for x in want if got.get(x) != want[x]]
            print(k, "MISMATCH", diff, args, "ranks", ranks, env.get("PBSIM_SCRATCH_MB"), env.get("PBSIM_JOB_REF_GB"))
            if os.environ.get("FUZZ_DIFF"):
                for x in diff:
                    a, b = got.get(x, b""), want[x]
                    n = next((i for i, (u, v) in enumerate(zip(a, b)) if u != v), min(len(a), len(b)))
                    ln = a[:n].count(b"\n")
                    print("   ", x, "sizes", len(a), len(b), "first diff at byte", n, "line", ln)
                    print("      got ", a[max(0, n - 70):n + 50])
                    print("      want", b[max(0, n - 70):n + 50])
            bad += 1
print("swept", k1 - k0, "cases,", bad, "bad")
sys.exit(1 if bad else 0)
