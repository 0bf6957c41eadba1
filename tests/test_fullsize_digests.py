"""The digest machinery behind the BASELINE-size reference goldens (tests/golden/fullsize.json), on the CPU:

* the small case of fullsize_cases.py was run by the REFERENCE through the same stubs (crcsum.c in place of gzip): the oracle's
  files for the same command must have exactly those CRC-32s and lengths, and the same stderr report;
* harness.synth_bases_torch (what the GPU box generates the 750 Mbp record with) equals harness.synth_bases (what the reference
  was fed) -- here on torch's CPU device, same integer arithmetic;
* every case of fullsize_cases.py has its digests committed."""
import os
import zlib

import numpy as np
import pytest

import harness
from fullsize_cases import FULLSIZE


def test_every_case_has_reference_digests():
    full = harness.load_fullsize()
    for name, case in FULLSIZE.items():
        assert name in full, "run tests/golden/make_fullsize.py %s" % name
        e = full[name]
        assert e["record"] == list(case["record"]) and e["args"] == case["args"]
        streams = [k for k in e if k.startswith(".")]
        assert ".maf" in streams and (".fq" in streams or ".sam" in streams)
        for k in streams:
            assert len(e[k]["crc32"]) == 8 and e[k]["bytes"] > 0
        assert "read num. :" in e["stderr"]


def test_oracle_reproduces_the_reference_digest_of_the_small_case(tmp_path):
    name = "t0_errhmm_ont_200k_d5"
    want = harness.load_fullsize()[name]
    length, seed = FULLSIZE[name]["record"]
    seq = harness.synth_bases(length, seed)
    fa = tmp_path / "g.fa"
    with open(fa, "wb") as f:
        f.write(b">synth_%d_%d\n" % (length, seed))
        rows = seq.reshape(-1, 80)
        f.write(np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1).tobytes())
    od = tmp_path / "o"
    od.mkdir()
    got = harness.run_oracle(FULLSIZE[name]["args"] + ["--genome", str(fa)], "philox", str(od))
    for key in (".fq", ".maf"):
        data = got["_0001" + key]
        assert (len(data), "%08x" % zlib.crc32(data)) == (want[key]["bytes"], want[key]["crc32"]), key
    assert got[".stderr"].decode() == want["stderr"]


@pytest.mark.parametrize("n,seed", [(1, 1), (1000, 7), ((1 << 22) + 12345, 101), (3_000_000, 104)])
def test_synth_bases_torch_equals_numpy(n, seed):
    import torch
    a = harness.synth_bases(n, seed)
    b = harness.synth_bases_torch(n, seed, device="cpu").numpy()
    assert np.array_equal(a, b)
