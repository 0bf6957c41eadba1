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

# ---- round 4 (VERDICT r3 "missing" 2, 4, 5; "weak": configs[0] verbatim) -------------------------------------------------
# the reference's own runnable fixtures through its README commands (README.md:119-165; 100 transcripts / 2 260 reads,
# 100 templates), committed gzip-compressed under inputs/ as data; default --seed is the clock, so the seed is named
CASES["trans_qshmm_rsii_readme"] = dict(
    args=["--strategy", "trans", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model",
          "--transcript", "INPUT:sample.transcript", "--seed", "1"])
CASES["trans_errhmm_rsii_readme"] = dict(
    args=["--strategy", "trans", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-RSII.model",
          "--transcript", "INPUT:sample.transcript", "--seed", "1"])
CASES["templ_qshmm_rsii_readme"] = dict(
    args=["--strategy", "templ", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model",
          "--template", "INPUT:sample.template", "--seed", "1"])
CASES["templ_qshmm_rsii_readme_pass10"] = dict(
    args=["--strategy", "templ", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model",
          "--template", "INPUT:sample.template", "--seed", "2", "--pass-num", "10"])
# BASELINE.json configs[0] verbatim (sample/sample.fasta is not in the reference tree: a 1 Mbp record generated from
# integer arithmetic, harness.synth_bases)
CASES["wgs_errhmm_rsii_config0"] = dict(
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-RSII.model", "--depth", "20",
          "--genome", "INPUT:synth_1000000_1.fa", "--seed", "1"])
# models whose cumulative tables do not end at 1000 / 100 (pbsim.cpp:3715-3789, 2066-2142: the moduli are whatever the
# rows round to) and classes with more states than the wave walkers take: tests/golden/make_models.py
CASES["wgs_errhmm_synthmod"] = dict(
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:SYNTH-ERRHMM-MOD.model",
          "--genome", "INPUT:quirk.fa", "--depth", "6", "--seed", "21"] + SHORT)
CASES["wgs_errhmm_synthmod_acc95_hpbias3"] = dict(   # classes above the model's range (Q3) with odd moduli
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:SYNTH-ERRHMM-MOD.model",
          "--genome", "INPUT:quirk.fa", "--depth", "4", "--seed", "22", "--accuracy-mean", "0.95", "--hp-del-bias", "3",
          "--pass-num", "2"] + SHORT)
CASES["wgs_errhmm_synths35"] = dict(
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:SYNTH-ERRHMM-S35.model",
          "--genome", "INPUT:quirk.fa", "--depth", "6", "--seed", "23"] + SHORT)
CASES["trans_errhmm_synthmod"] = dict(
    args=["--strategy", "trans", "--method", "errhmm", "--errhmm", "MODEL:SYNTH-ERRHMM-MOD.model",
          "--transcript", "INPUT:tiny.transcript", "--seed", "24"])
CASES["wgs_qshmm_synthmod"] = dict(
    args=["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:SYNTH-QSHMM-MOD.model",
          "--genome", "INPUT:quirk.fa", "--depth", "5", "--seed", "25"] + SHORT)
CASES["wgs_qshmm_synthmod_pass3_acc92"] = dict(
    args=["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:SYNTH-QSHMM-MOD.model",
          "--genome", "INPUT:quirk.fa", "--depth", "3", "--seed", "26", "--pass-num", "3", "--accuracy-mean", "0.92"] + SHORT)
# reads near the reference's shape limit (FASTQ_LEN_MAX 1 000 000, pbsim.cpp:27): ONT ultra-long settings on a 3 Mbp
# record -- rows of up to 2 M columns through the scratch layout, the text kernels and both walkers
ULTRA = ["--length-mean", "200000", "--length-sd", "150000", "--length-max", "1000000", "--genome", "INPUT:synth_3000000_7.fa"]
CASES["wgs_errhmm_ont_ultralong"] = dict(
    args=["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model", "--depth", "8", "--seed", "31"] + ULTRA)
CASES["wgs_qshmm_rsii_ultralong_pass2"] = dict(
    args=["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model", "--depth", "4", "--seed", "32",
          "--pass-num", "2"] + ULTRA)

# cases whose complete outputs are committed (gzip) in addition to the hashes
FULL = ["wgs_errhmm-ont_quirk", "wgs_qshmm_rsii_pass3", "trans_errhmm_sequel"]

MODES = ["glibc", "philox"]
