"""The independent BAM reader (tests/bam_spec_reader.py) against a file assembled BY HAND from the byte layouts of SAMv1
4.1 / 4.2 -- the literals below were written from the specification's tables, not produced by any encoder in this repo --
and against zlib-made BGZF blocks with the framing mistakes a writer can make."""
import struct
import zlib

import pytest

import bam_spec_reader as R


def bgzf(data, extra_first=b""):
    """one BGZF block around `data`, built with zlib's raw deflate (an encoder that is not the product's)"""
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    z = co.compress(data) + co.flush()
    bc = b"BC" + struct.pack("<HH", 2, 0)           # SLEN = 2, BSIZE patched below
    extra = extra_first + bc
    total = 12 + len(extra) + len(z) + 8
    extra = extra_first + b"BC" + struct.pack("<HH", 2, total - 1)
    return (bytes([31, 139, 8, 4]) + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", len(extra)) + extra + z +
            struct.pack("<II", zlib.crc32(data), len(data)))


EOF_MARKER = bgzf(b"")

# One unmapped read "r1", flag 4, SEQ = ACGTN (odd length: padding nibble), QUAL = "!#%+5" (phred 0 2 4 10 20),
# optional fields  np:i:3 (as 'C')  rq:f:0.5  zm:i:-70000 (as 'i')  ip:B:C,1,2,3   sn:B:f,1.5,2.0   RG:Z:ab
HAND_RECORD = bytes.fromhex(
    "ffffffff"          # refID = -1
    "ffffffff"          # pos = -1
    "03"                # l_read_name (incl. NUL)
    "00"                # mapq
    "4812"              # bin = 4680 = reg2bin(-1, 0)
    "0000"              # n_cigar_op
    "0400"              # flag = 4
    "05000000"          # l_seq
    "ffffffff"          # next_refID
    "ffffffff"          # next_pos
    "00000000"          # tlen
    "723100"            # "r1\0"
    "1248f0"            # A=1 C=2 | G=4 T=8 | N=15 pad 0
    "000204" "0a14"     # qualities
    "6e704303"          # np C 3
    "727166" "0000003f"  # rq f 0.5
    "7a6d69" "90eefeff"  # zm i -70000
    "697042" "43" "03000000" "010203"
    "736e42" "66" "02000000" "0000c03f" "00000040"
    "52475a" "616200")
HEADER_TEXT = b"@HD\tVN:1.5\tSO:unknown\n"
HAND_BAM = b"BAM\x01" + struct.pack("<I", len(HEADER_TEXT)) + HEADER_TEXT + struct.pack("<I", 0) + \
    struct.pack("<I", len(HAND_RECORD)) + HAND_RECORD


def test_hand_assembled_file():
    raw = bgzf(HAND_BAM[:30]) + bgzf(HAND_BAM[30:], extra_first=b"XY" + struct.pack("<H", 3) + b"abc") + EOF_MARKER
    text, refs, recs = R.read_bam(raw)
    assert text == HEADER_TEXT and refs == [] and len(recs) == 1
    a = recs[0]
    assert R.sam_fields(a, refs) == ["r1", "4", "*", "0", "0", "*", "*", "0", "0", "ACGTN", "!#%+5"]
    assert a["bin"] == R.reg2bin(-1, 0) == 4680
    assert a["aux"] == [("np", "C", 3), ("rq", "f", 0.5), ("zm", "i", -70000), ("ip", "BC", [1, 2, 3]),
                        ("sn", "Bf", [1.5, 2.0]), ("RG", "Z", "ab")]
    assert R.NIBBLE == list("=ACMGRSVTWYHKDBN")
    assert [R.smallest_int_type(v) for v in (0, 255, 256, 65535, 65536, -1, -128, -129, -32768, -32769)] == \
        list("CCSSIccssi")


@pytest.mark.parametrize("damage", ["crc", "isize", "no_eof", "bsize", "no_bc", "flags", "trailing"])
def test_framing_mistakes_are_caught(damage):
    good = bytearray(bgzf(HAND_BAM))
    raw = bytes(good) + EOF_MARKER
    if damage == "crc":
        good[-8] ^= 1
        raw = bytes(good) + EOF_MARKER
    elif damage == "isize":
        good[-4] ^= 1
        raw = bytes(good) + EOF_MARKER
    elif damage == "no_eof":
        raw = bytes(good)
    elif damage == "bsize":
        good[16] ^= 1
        raw = bytes(good) + EOF_MARKER
    elif damage == "no_bc":
        good[12:14] = b"XX"
        raw = bytes(good) + EOF_MARKER
    elif damage == "flags":
        good[3] |= 8
        raw = bytes(good) + EOF_MARKER
    elif damage == "trailing":
        raw = bytes(good) + EOF_MARKER + b"\0"
    with pytest.raises((R.BamFormatError, zlib.error)):
        R.read_bam(raw)
