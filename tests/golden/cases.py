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

# SURVEY Q7, pinned: QSHMM-ONT-HQ.model has classes with 52 and 56 states; the compiled reference writes states > STATE_MAX
# into the neighbouring rows of its flat tables and loops over 50 states (pbsim.cpp:160-166, 5606-5626, 2073, 2124)
CASES["wgs_qshmm_onthq_acc95"] = dict(
    args=["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-ONT-HQ.model",
          "--genome", "INPUT:quirk.fa", "--depth", "5", "--seed", "11", "--accuracy-mean", "0.95"] + SHORT)
CASES["wgs_qshmm_onthq_pass2_hpbias2"] = dict(
    args=["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-ONT-HQ.model",
          "--genome", "INPUT:quirk.fa", "--depth", "3", "--seed", "12", "--accuracy-mean", "0.97", "--pass-num", "2",
          "--hp-del-bias", "2"] + SHORT)
# SURVEY Q5, pinned: errhmm trans with accuracy class 100 in range -- the verbatim copy (pbsim.cpp:4533) clobbers the
# per-transcript read counter (:4487), so fewer reads than the expression values are made
CASES["trans_errhmm_rsii_acc98"] = dict(
    args=["--strategy", "trans", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-RSII.model",
          "--transcript", "INPUT:tiny.transcript", "--seed", "3", "--accuracy-mean", "0.98"])
CASES["trans_errhmm_sequel_acc99_pass2"] = dict(
    args=["--strategy", "trans", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-SEQUEL.model",
          "--transcript", "INPUT:tiny.transcript", "--seed", "14", "--accuracy-mean", "0.99", "--pass-num", "2"] + SHORT)

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

# ---- sampling method (pbsim.cpp:1694-1949): quality strings come from a FASTQ, tests/golden/make_sample_fastq.py
SAMPLE = ["--strategy", "wgs", "--method", "sample"]
CASES["wgs_sample_plain"] = dict(     # 4 copies per string + a second sweep; default ratio 6:55:39
    args=SAMPLE + ["--sample", "INPUT:sample.fastq", "--genome", "INPUT:plain.fa", "--depth", "3", "--seed", "1"])
CASES["wgs_sample_quirk"] = dict(     # strings longer than the record (clipped to it), quota below the profile's total
    args=SAMPLE + ["--sample", "INPUT:sample.fastq", "--genome", "INPUT:quirk.fa", "--depth", "5", "--seed", "2"])
CASES["wgs_sample_delheavy"] = dict(  # deletions outnumber insertions: every copy is shorter than the one before
    args=SAMPLE + ["--sample", "INPUT:sample.fastq", "--genome", "INPUT:plain.fa", "--depth", "2.5", "--seed", "3",
                   "--difference-ratio", "10:30:60", "--hp-del-bias", "3", "--accuracy-min", "0.8",
                   "--length-min", "150", "--length-max", "5000"])
CASES["wgs_sample_store"] = dict(     # --sample + --sample-profile-id: the filtered profile is written out
    args=SAMPLE + ["--sample", "INPUT:sample.fastq", "--sample-profile-id", "g1", "--genome", "INPUT:quirk.fa",
                   "--depth", "2", "--seed", "4", "--accuracy-max", "0.99"])
CASES["wgs_sample_reuse"] = dict(     # --sample-profile-id alone: the stored profile is read back
    setup=SAMPLE + ["--sample", "INPUT:sample.fastq", "--sample-profile-id", "g1", "--genome", "INPUT:quirk.fa",
                    "--depth", "1", "--seed", "4", "--accuracy-max", "0.99"],
    args=SAMPLE + ["--sample-profile-id", "g1", "--genome", "INPUT:plain.fa", "--depth", "2", "--seed", "5"])

# cases whose complete outputs are committed (gzip) in addition to the hashes
FULL = ["wgs_errhmm-ont_quirk", "wgs_qshmm_rsii_pass3", "trans_errhmm_sequel"]

MODES = ["glibc", "philox"]
