"""GPU box: a long one-off sweep of tests/test_gpu_fuzz.py's generator (cases K0..K1), product vs oracle byte for byte.
usage: python tools/fuzz_sweep.py K0 K1"""
import os, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
import harness, product
from test_gpu_fuzz import make_case

bad = 0
k0, k1 = int(sys.argv[1]), int(sys.argv[2])
for k in range(k0, k1):
    recs, args = make_case(k)
    with tempfile.TemporaryDirectory() as td:
        fa = os.path.join(td, "g.fa")
        with open(fa, "w") as f:
            for i, s in enumerate(recs, 1):
                f.write(f">rec{i} fuzz\n")
                w = 50 + 10 * i
                for p in range(0, len(s), w):
                    f.write(s[p:p + w] + "\n")
        a = args + ["--genome", fa]
        try:
            want = harness.run_oracle(a, "philox", td)
        except RuntimeError as e:          # the reference's own limits (e.g. "length parameters are not appropriate")
            print(k, "oracle refused:", str(e)[-80:].replace("\n", " "))
            continue
        try:
            outs, _ = product.run_wgs(harness.resolve(a))
        except Exception as e:
            print(k, "PRODUCT FAILED", e, a)
            bad += 1
            continue
        for key, v in outs.items():
            if v != want[key]:
                print(k, "MISMATCH", key, a)
                bad += 1
                break
print("swept", k1 - k0, "cases,", bad, "bad")
sys.exit(1 if bad else 0)
