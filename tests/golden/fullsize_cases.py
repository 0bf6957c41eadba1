"""The BASELINE.json configurations at (or near) their benchmarked size, as cases the REFERENCE ITSELF has run
(tests/golden/make_fullsize.py -> tests/golden/fullsize.json): what is kept of a run is the CRC-32 and the length of each
output stream plus the stderr report -- the streams themselves are 10-65 GB of text.

A record is `harness.synth_bases(length, seed)`: plain 64-bit integer arithmetic, the same bytes on the CPU (numpy, where the
reference runs) and on the GPU box (numpy or torch, `harness.synth_bases_torch`), so nothing large is committed.

  args      : the reference's command line without --genome / --prefix (MODEL:<name> = tests/golden/models/<name>)
  record    : (length, seed) of the one FASTA record
"""

FULLSIZE = {
    # a small case through the same digest machinery: the CPU suite runs the oracle on it (tests/test_fullsize_digests.py)
    "t0_errhmm_ont_200k_d5": {
        "record": (200_000, 100),
        "args": ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model", "--depth", "5", "--seed", "1"],
    },
    # configs[1]: one of the four 750 Mbp records of the 3 Gbp genome at depth 20 (15 Gbases, 1.7 M reads, 63 GB of text)
    "c1_errhmm_ont_750m_d20": {
        "record": (750_000_000, 101),
        "args": ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT.model", "--depth", "20", "--seed", "1"],
    },
    # configs[4]: ERRHMM-ONT-HQ at depth 60 (7.2 Gbases on a 120 Mbp record; the eight-rank job is run as ranks on the one GPU)
    "c4_errhmm_onthq_120m_d60": {
        "record": (120_000_000, 104),
        "args": ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT-HQ.model", "--depth", "60", "--seed", "1"],
    },
    # configs[4] at the BASELINE record size: one 750 Mbp record at depth 60 (45 Gbases, 5.1 M reads; the reference: 93 CPU-minutes)
    "c4_errhmm_onthq_750m_d60": {
        "record": (750_000_000, 105),
        "args": ["--strategy", "wgs", "--method", "errhmm", "--errhmm", "MODEL:ERRHMM-ONT-HQ.model", "--depth", "60", "--seed", "1"],
    },
    # configs[2] larger: 60 Mbp at depth 20, ten passes (12 G subread bases, 72 GB of SAM text)
    "c2_qshmm_rsii_60m_d20_pass10": {
        "record": (60_000_000, 106),
        "args": ["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model", "--depth", "20", "--pass-num", "10",
                 "--seed", "1"],
    },
    # configs[2]: QSHMM-RSII --pass-num 10 at depth 20 (4 G subread bases on a 20 Mbp record): SAM text + MAF
    "c2_qshmm_rsii_20m_d20_pass10": {
        "record": (20_000_000, 102),
        "args": ["--strategy", "wgs", "--method", "qshmm", "--qshmm", "MODEL:QSHMM-RSII.model", "--depth", "20", "--pass-num", "10",
                 "--seed", "1"],
    },
}
