#!/usr/bin/env python3
"""Draw census of the reference: how often each rand() call site of pbsim.cpp is reached, for a few
golden cases, measured by running the reference itself under oracle/ref_shim.h with PBSHIM_CENSUS (keyed-Philox mode: the
stream the product reproduces).  Writes tests/golden/census.json {case: {line: count}}; tests/test_gpu_census.py holds the
product's per-task counters against it.  Needs /root/reference and `make -C oracle`; run only to regenerate."""
import json
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import harness  # noqa: E402
from cases import CASES  # noqa: E402

CENSUS_CASES = ["wgs_errhmm-ont_quirk", "wgs_errhmm_rsii_config0", "wgs_errhmm_synthmod_acc95_hpbias3", "wgs_errhmm_rsii_acc98",
                "wgs_qshmm_rsii_pass3", "wgs_qshmm_synthmod", "trans_errhmm_sequel", "templ_qshmm_rsii_pass2"]


def main():
    out = {}
    for case in CENSUS_CASES:
        with tempfile.TemporaryDirectory() as td:
            args = harness.resolve(CASES[case]["args"])
            seed = args[args.index("--seed") + 1]
            stubs = os.path.join(td, "stubs")
            harness.make_stubs(stubs)
            cf = os.path.join(td, "census.tsv")
            env = dict(os.environ, PATH=stubs + ":" + os.environ["PATH"], PBSHIM_SEED=seed, PBSHIM_MODE="philox", PBSHIM_CENSUS=cf)
            subprocess.run([harness.REF_PHILOX] + args + ["--prefix", os.path.join(td, "out")], env=env, capture_output=True,
                           check=True, cwd=td)
            out[case] = {int(l.split()[0]): int(l.split()[1]) for l in open(cf)}
            print(case, sum(out[case].values()), "draws at", len(out[case]), "sites")
    with open(os.path.join(HERE, "census.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
