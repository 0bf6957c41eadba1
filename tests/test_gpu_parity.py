"""Parity proper (GPU): the HIP product, called through the C ABI, against the
oracle in keyed-Philox mode and against the committed golden vectors (which the
reference itself produced through oracle/ref_shim.h).  Byte-exact."""
import pytest

import harness
import product
from cases import CASES

MANIFEST = harness.load_manifest()
WGS_ERR = sorted(c for c in CASES if c.startswith("wgs_errhmm") and "pass3" not in c)   # incl. configs[0] verbatim, the synthetic-moduli / 35- and 50-state models, ultra-long reads

pytestmark = pytest.mark.gpu


def _cmp(outs, want_outs, case):
    for k, v in outs.items():
        assert k in want_outs, (case, k)
        if v != want_outs[k]:
            n = next(i for i, (x, y) in enumerate(zip(v, want_outs[k])) if x != y) if len(v) and len(want_outs[k]) else 0
            raise AssertionError(f"{case}{k}: first difference at byte {n}: "
                                 f"{v[max(0, n - 60):n + 60]!r} != {want_outs[k][max(0, n - 60):n + 60]!r} "
                                 f"(sizes {len(v)} vs {len(want_outs[k])})")


@pytest.mark.parametrize("case", WGS_ERR)
def test_wgs_errhmm_matches_oracle_and_golden(case, tmp_path):
    args = harness.resolve(CASES[case]["args"])
    outs, _ = product.run_wgs(args)
    want = harness.run_oracle(CASES[case]["args"], "philox", str(tmp_path))
    _cmp(outs, want, case)
    gold = MANIFEST[f"{case}/philox"]
    for k, v in outs.items():
        assert harness.sha(v) == gold[k]["sha256"], (case, k)


WGS_OTHER = ["wgs_errhmm_sequel_pass3", "wgs_qshmm_rsii_pass1", "wgs_qshmm_rsii_pass3", "wgs_qshmm_ont_ratio",
             "wgs_qshmm_synthmod", "wgs_qshmm_synthmod_pass3_acc92", "wgs_qshmm_rsii_ultralong_pass2"]


@pytest.mark.parametrize("case", WGS_OTHER)
def test_wgs_qshmm_and_multipass_match_oracle_and_golden(case, tmp_path):
    args = harness.resolve(CASES[case]["args"])
    outs, _ = product.run_wgs(args)
    want = harness.run_oracle(CASES[case]["args"], "philox", str(tmp_path))
    _cmp(outs, want, case)
    gold = MANIFEST[f"{case}/philox"]
    for k, v in outs.items():
        assert harness.sha(v) == gold[k]["sha256"], (case, k)


@pytest.mark.parametrize("case", WGS_ERR + WGS_OTHER + ["wgs_qshmm_onthq_acc95", "wgs_qshmm_onthq_pass2_hpbias2"])
def test_job_pipeline_matches_golden(case):
    """the same commands through pbsim_job_* (all records resident, one pipeline of rounds; a 4 MB scratch pool forces many rounds)"""
    args = harness.resolve(CASES[case]["args"])
    outs, _ = product.run_wgs_job(args, scratch_mb=product.scratch_mb_for(case))
    gold = MANIFEST[f"{case}/philox"]
    for k, v in outs.items():
        assert harness.sha(v) == gold[k]["sha256"], (case, k)
