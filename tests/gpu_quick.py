"""Ad-hoc GPU check used during development: python tests/gpu_quick.py CASE..."""
import sys, os, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import harness, product
from cases import CASES

for case in sys.argv[1:]:
    args = harness.resolve(CASES[case]["args"])
    t = time.time()
    try:
        outs, stats = product.run_wgs(args)
    except Exception as e:
        print(case, "FAILED:", e)
        continue
    dt = time.time() - t
    with tempfile.TemporaryDirectory() as td:
        want = harness.run_oracle(CASES[case]["args"], "philox", td)
    ok = True
    for k, v in outs.items():
        w = want.get(k)
        if v != w:
            ok = False
            n = next((i for i, (x, y) in enumerate(zip(v, w)) if x != y), min(len(v), len(w)))
            print(case, k, "DIFF at", n, "sizes", len(v), len(w))
            print("  got :", v[max(0, n - 80):n + 40])
            print("  want:", w[max(0, n - 80):n + 40])
    print(case, "OK" if ok else "MISMATCH", "%.2fs" % dt, [s.res_num for s in stats])
