"""Golden case matrix shared by make_golden.py (generation, needs the compiled
reference under oracle/_ref) and the tests (comparison, needs nothing but the
committed manifest).  Each case is a pbsim command line minus --prefix/--seed
handling; MODEL/INPUT placeholders are resolved by the runner."""

ERR_MODELS = ["ERRHMM-RSII", "ERRHMM-SEQUEL", "ERRHMM-ONT", "ERRHMM-ONT-HQ"]
SHORT = ["--length-mean", "1200", "--length-sd", "900"]

CASES = {}
for m in ERR_MODELS:
    CASES[f"wgs_{m.lower()}_quirk"] = dict(
        args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", f"MODEL:{m}.model",
              "--genome", "INPUT:quirk.fa", "--depth", "6", "--seed", "1"] + SHORT)
CASES["wgs_errhmm_rsii_default"] = dict(
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-RSII.model",
          "--genome", "INPUT:plain.fa", "--depth", "3", "--seed", "7"])
CASES["wgs_errhmm_ont_hpbias5"] = dict(
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model",
          "--genome", "INPUT:quirk.fa", "--depth", "4", "--seed", "3", "--hp-del-bias", "5"] + SHORT)
CASES["wgs_errhmm_rsii_acc98"] = dict(   # classes 73..100: above-range branch + verbatim class 100 (Q3, Q8)
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-RSII.model",
          "--genome", "INPUT:quirk.fa", "--depth", "4", "--seed", "5", "--accuracy-mean", "0.98"] + SHORT)
CASES["wgs_errhmm_ont_sd0"] = dict(      # fixed read length (len_sd == 0 branch, pbsim.cpp:3639)
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model",
          "--genome", "INPUT:quirk.fa", "--depth", "3", "--seed", "2", "--length-mean", "800", "--length-sd", "0"])
CASES["wgs_errhmm_sequel_pass3"] = dict(  # multi-pass SAM text for the error model
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-SEQUEL.model",
          "--genome", "INPUT:quirk.fa", "--depth", "3", "--seed", "4", "--pass-num", "3"] + SHORT)
CASES["wgs_qshmm_rsii_pass1"] = dict(
    args=["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model",
          "--genome", "INPUT:quirk.fa", "--depth", "5", "--seed", "1"] + SHORT)
CASES["wgs_qshmm_rsii_pass3"] = dict(
    args=["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model",
          "--genome", "INPUT:quirk.fa", "--depth", "3", "--seed", "1", "--pass-num", "3"] + SHORT)
CASES["wgs_qshmm_ont_ratio"] = dict(
    args=["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-ONT.model",
          "--genome", "INPUT:quirk.fa", "--depth", "4", "--seed", "9", "--difference-ratio", "39:24:36",
          "--hp-del-bias", "3"] + SHORT)
CASES["trans_errhmm_sequel"] = dict(
    args=["--strategy", "trans", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-SEQUEL.model",
          "--transcript", "INPUT:tiny.transcript", "--seed", "1"])
CASES["trans_errhmm_ont_hpbias4"] = dict(
    args=["--strategy", "trans", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model",
          "--transcript", "INPUT:tiny.transcript", "--seed", "6", "--hp-del-bias", "4"] + SHORT)
CASES["trans_qshmm_rsii"] = dict(
    args=["--strategy", "trans", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model",
          "--transcript", "INPUT:tiny.transcript", "--seed", "8"])

CASES["templ_errhmm_sequel"] = dict(
    args=["--strategy", "templ", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-SEQUEL.model",
          "--template", "INPUT:tiny.template", "--seed", "1"])
CASES["templ_errhmm_rsii_pass3_hpbias2"] = dict(
    args=["--strategy", "templ", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-RSII.model",
          "--template", "INPUT:tiny.template", "--seed", "3", "--pass-num", "3", "--hp-del-bias", "2",
          "--accuracy-mean", "0.97"])
CASES["templ_qshmm_rsii_pass2"] = dict(
    args=["--strategy", "templ", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model",
          "--template", "INPUT:tiny.template", "--seed", "5", "--pass-num", "2"])

# cases whose complete outputs are committed (gzip) in addition to the hashes
FULL = ["wgs_errhmm-ont_quirk", "wgs_qshmm_rsii_pass3", "trans_errhmm_sequel"]

MODES = ["glibc", "philox"]
