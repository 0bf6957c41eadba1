"""Native BAM output (GPU emits BAM records, the CLI BGZF-frames them) against the golden SAM
text the reference produced for the same run: every field and tag of every record, plus the
BGZF container rules (SAMv1 4.1: BC subfield, block size, CRC32, ISIZE, EOF marker)."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import harness
from cases import CASES

pytestmark = pytest.mark.gpu
CLI = os.path.join(harness.ROOT, "pbsim3_amd", "bin", "pbsim")
EOF_BLOCK = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
CODES = "=ACMGRSVTWYHKDBN"


def bgzf_decompress(raw):
    out, p, nblocks = [], 0, 0
    assert raw.endswith(EOF_BLOCK)
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04" and raw[p + 10:p + 16] == b"\x06\x00BC\x02\x00"
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        assert bsize <= 65536
        payload = raw[p + 18:p + bsize - 8]
        data = zlib.decompress(payload, -15)
        crc, isize = struct.unpack_from("<II", raw, p + bsize - 8)
        assert isize == len(data) <= 65536 and crc == (zlib.crc32(data) & 0xffffffff)
        out.append(data)
        p += bsize
        nblocks += 1
    assert p == len(raw) and nblocks >= 2
    return b"".join(out)


def parse_bam(b):
    assert b[:4] == b"BAM\x01"
    l_text = struct.unpack_from("<I", b, 4)[0]
    text = b[8:8 + l_text]
    p = 8 + l_text
    assert struct.unpack_from("<I", b, p)[0] == 0  # n_ref
    p += 4
    recs = []
    while p < len(b):
        bs = struct.unpack_from("<I", b, p)[0]
        r = b[p + 4:p + 4 + bs]
        p += 4 + bs
        ref, pos, lname, mapq, bin_, ncig, flag, lseq, nref, npos, tlen = struct.unpack_from("<iiBBHHHIiii", r, 0)
        assert (ref, pos, ncig, nref, npos, tlen, bin_) == (-1, -1, 0, -1, -1, 0, 4680)
        o = 32
        name = r[o:o + lname - 1].decode()
        assert r[o + lname - 1] == 0
        o += lname
        packed = r[o:o + (lseq + 1) // 2]
        seq = "".join(CODES[x >> 4] + CODES[x & 15] for x in packed)[:lseq]
        o += (lseq + 1) // 2
        qual = bytes(x + 33 for x in r[o:o + lseq]).decode()
        o += lseq
        tags = []
        while o < len(r):
            tag, typ = r[o:o + 2].decode(), chr(r[o + 2])
            o += 3
            if typ in "cCsSiI":
                fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[typ]
                v = struct.unpack_from(fmt, r, o)[0]
                o += struct.calcsize(fmt)
                tags.append((tag, "i", v, typ))
            elif typ == "f":
                tags.append((tag, "f", struct.unpack_from("<f", r, o)[0], typ))
                o += 4
            elif typ == "Z":
                e = r.index(b"\0", o)
                tags.append((tag, "Z", r[o:e].decode(), typ))
                o = e + 1
            elif typ == "B":
                sub, n = chr(r[o]), struct.unpack_from("<I", r, o + 1)[0]
                o += 5
                if sub == "C":
                    tags.append((tag, "B:C", list(r[o:o + n]), typ))
                    o += n
                else:
                    assert sub == "f"
                    tags.append((tag, "B:f", list(struct.unpack_from("<%df" % n, r, o)), typ))
                    o += 4 * n
            else:
                raise AssertionError(typ)
        recs.append(dict(name=name, flag=flag, mapq=mapq, seq=seq, qual=qual, tags=tags))
    return text, recs


def smallest(v):
    return ("c" if v >= -128 else "s" if v >= -32768 else "i") if v < 0 else ("C" if v < 256 else "S" if v < 65536 else "I")


@pytest.mark.parametrize("case", ["wgs_qshmm_rsii_pass3", "wgs_errhmm_sequel_pass3", "templ_qshmm_rsii_pass2"])
def test_native_bam_equals_golden_sam(case, tmp_path):
    args = harness.resolve(CASES[case]["args"])
    p = subprocess.run([CLI] + args + ["--prefix", str(tmp_path / "out"), "--gzip-threads", "4"], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    gold_dir = tmp_path / "g"
    gold_dir.mkdir()
    gold = harness.run_oracle(CASES[case]["args"], "philox", str(gold_dir))
    sams = sorted(k for k in gold if k.endswith(".sam"))
    assert sams
    for k in sams:
        compare_bam_with_sam((tmp_path / ("out" + k[:-4] + ".bam")).read_bytes(), gold[k])


def compare_bam_with_sam(bam, sam_text):
    """every header byte, field and tag of the BAM container against the SAM text of the same run"""
    text, recs = parse_bam(bgzf_decompress(bam))
    lines = sam_text.decode().split("\n")
    hdr = [l for l in lines if l.startswith("@")]
    body = [l for l in lines if l and not l.startswith("@")]
    assert text.decode() == "\n".join(hdr) + "\n"
    assert len(recs) == len(body)
    for r, line in zip(recs, body):
        f = line.split("\t")
        # 4-bit codes carry no case (a lower-case first base survives in SAM text only, Q6)
        assert (r["name"], r["flag"], r["mapq"], r["seq"], r["qual"]) == (f[0], int(f[1]), int(f[4]), f[9].upper(), f[10])
        assert [t[0] for t in r["tags"]] == [x[:2] for x in f[11:]]
        for (tag, kind, v, typ), x in zip(r["tags"], f[11:]):
            val = x[5:]
            if kind == "i":
                assert v == int(val) and typ == smallest(int(val)), (tag, v, typ)
            elif kind == "f":
                assert v == np.float32(float(val))
            elif kind == "Z":
                assert v == val
            elif kind == "B:C":
                assert x[5:7] == "C," and v == [int(y) for y in val.split(",")[1:]]
            else:
                assert v == [float(y) for y in val.split(",")[1:]]


@pytest.mark.parametrize("case", ["wgs_qshmm_rsii_pass3", "wgs_errhmm_sequel_pass3", "wgs_qshmm_onthq_pass2_hpbias2",
                                  "trans_errhmm_sequel_acc99_pass2"])
def test_native_bam_through_the_independent_reader(case, tmp_path):
    """The same files through tests/bam_spec_reader.py -- a reader written from SAMv1 4.1 / 4.2 alone that shares nothing
    with the parser above or with the writer (it stands in for samtools, which this image lacks): container framing, every
    mandatory column as SAM text, every optional field with its type letter."""
    import bam_spec_reader as R
    args = harness.resolve(CASES[case]["args"])
    p = subprocess.run([CLI] + args + ["--prefix", str(tmp_path / "out")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    gold_dir = tmp_path / "g"
    gold_dir.mkdir()
    gold = harness.run_oracle(CASES[case]["args"], "philox", str(gold_dir))
    sams = sorted(k for k in gold if k.endswith(".sam"))
    assert sams
    for k in sams:
        text, refs, recs = R.read_bam((tmp_path / ("out" + k[:-4] + ".bam")).read_bytes())
        lines = gold[k].decode().split("\n")
        assert text.decode() == "".join(l + "\n" for l in lines if l.startswith("@")) and refs == []
        body = [l.split("\t") for l in lines if l and not l.startswith("@")]
        assert len(recs) == len(body)
        for a, f in zip(recs, body):
            want = f[:11]
            want[9] = want[9].upper()      # 4-bit codes carry no case (Q6: a lower-case first base exists in SAM text only)
            assert R.sam_fields(a, refs) == want
            assert a["bin"] == R.reg2bin(a["pos"], a["pos"] + 1)
            assert len(a["aux"]) == len(f) - 11
            for (tag, typ, val), col in zip(a["aux"], f[11:]):
                t, ty, v = col.split(":", 2)
                assert tag == t
                if ty == "i":
                    assert val == int(v) and typ == R.smallest_int_type(int(v))
                elif ty == "f":
                    assert typ == "f" and val == np.float32(float(v))
                elif ty == "Z":
                    assert typ == "Z" and val == v
                else:
                    assert ty == "B" and typ == "B" + v[0]
                    nums = v.split(",")[1:]
                    assert val == ([int(x) for x in nums] if v[0] != "f" else [float(np.float32(float(x))) for x in nums])
