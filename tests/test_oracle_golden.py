"""The oracle (oracle/pbsim_oracle.c) against the committed golden vectors,
which were produced by the reference itself (tests/golden/make_golden.py):
glibc mode = unmodified pbsim.cpp, philox mode = pbsim.cpp + oracle/ref_shim.h.
Byte-exact: FASTQ / SAM text / MAF / .ref / stderr report."""
import gzip
import os

import pytest

import harness
from cases import CASES, FULL, MODES

MANIFEST = harness.load_manifest()


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_matches_reference_golden(case, mode, tmp_path):
    outs = harness.run_oracle(CASES[case]["args"], mode, str(tmp_path), case=CASES[case])
    want = MANIFEST[f"{case}/{mode}"]
    assert sorted(outs) == sorted(want), (sorted(outs), sorted(want))
    for k, v in outs.items():
        assert len(v) == want[k]["bytes"], (case, mode, k)
        assert harness.sha(v) == want[k]["sha256"], (case, mode, k)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("case", FULL)
def test_full_vectors_are_consistent_with_manifest(case, mode):
    full = os.path.join(harness.GOLDEN, "full")
    want = MANIFEST[f"{case}/{mode}"]
    seen = 0
    for k in want:
        p = os.path.join(full, f"{case}.{mode}{k}.gz")
        if os.path.exists(p):
            with gzip.open(p, "rb") as f:
                assert harness.sha(f.read()) == want[k]["sha256"]
            seen += 1
    assert seen >= 3
