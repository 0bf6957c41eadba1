"""A second, independent reader for BAM files, written from the SAM/BAM specification alone (SAMv1 sections 4.1 "The BGZF
compression format" and 4.2 "The BAM format", RFC 1952 for the member framing): test infrastructure that stands in for
samtools / htslib, which this image does not have.  It shares no code and no constants with the product's BAM writer
(k_bam_finish, cli.cpp) or with tests/test_gpu_bam.py; where the specification fixes a value, the value is derived here from
the specification's own wording (see the comments), not copied from the writer.

    blocks(raw)            -> iterator of uncompressed BGZF block payloads (framing, CRC-32 and ISIZE verified)
    read_bam(raw)          -> (header text, references, [alignment dict, ...])
    sam_fields(alignment)  -> the eleven mandatory SAM columns as strings, as SAMv1 section 1.4 defines them from BAM fields
"""
import zlib


class BamFormatError(AssertionError):
    pass


def need(cond, what):
    if not cond:
        raise BamFormatError(what)


def le(buf, at, n, signed=False):
    return int.from_bytes(buf[at:at + n], "little", signed=signed)


def blocks(raw):
    """RFC 1952 member by member; SAMv1 4.1: each member has FLG.FEXTRA set and an extra subfield SI1=66 ('B'), SI2=67 ('C'),
    SLEN=2 whose value BSIZE is the total block size minus one; CM=8 (deflate); the data is a raw deflate stream followed by
    CRC32 and ISIZE (both of the uncompressed data); at most 64 KiB either side.  The file ends with an empty block."""
    view = memoryview(raw)
    at = 0
    last_isize = None
    n_blocks = 0
    while at < len(view):
        need(len(view) - at >= 18, "truncated member header at %d" % at)
        id1, id2, cm, flg = view[at], view[at + 1], view[at + 2], view[at + 3]
        need(id1 == 31 and id2 == 139, "not a gzip member at %d" % at)          # RFC 1952 2.3.1: ID1 = 31, ID2 = 139
        need(cm == 8, "compression method is not deflate")
        need(flg & 4, "FLG.FEXTRA is not set")                                   # bit 2 = FEXTRA
        need(flg & ~4 == 0, "BGZF sets no other flag")                           # no FNAME / FCOMMENT / FHCRC / FTEXT
        xlen = le(view, at + 10, 2)
        extra_at, extra_end = at + 12, at + 12 + xlen
        bsize = None
        p = extra_at
        while p + 4 <= extra_end:                                                # walk ALL subfields, BC need not be first
            si1, si2, slen = view[p], view[p + 1], le(view, p + 2, 2)
            if si1 == ord("B") and si2 == ord("C"):
                need(slen == 2, "BC subfield must be two bytes")
                bsize = le(view, p + 4, 2)
            p += 4 + slen
        need(p == extra_end, "extra field does not end on a subfield boundary")
        need(bsize is not None, "no BC subfield")
        total = bsize + 1
        need(total <= 65536 and at + total <= len(view), "bad BSIZE")
        data_at, trailer_at = extra_end, at + total - 8
        inflater = zlib.decompressobj(-15)
        data = inflater.decompress(bytes(view[data_at:trailer_at]))
        need(inflater.eof, "deflate stream does not end inside its block")
        need(inflater.unused_data == b"", "bytes between the deflate stream and the trailer")
        crc, isize = le(view, trailer_at, 4), le(view, trailer_at + 4, 4)
        need(isize == len(data) and isize <= 65536, "ISIZE mismatch")
        need(crc == zlib.crc32(data), "CRC-32 mismatch")
        last_isize = isize
        n_blocks += 1
        at += total
        yield data
    need(at == len(view), "trailing bytes")
    need(n_blocks > 0 and last_isize == 0, "no end-of-file marker (an empty BGZF block, SAMv1 4.1.2)")


# SAMv1 4.2.3: "=ACMGRSVTWYHKDBN" -> [0, 15]; built from the IUPAC table rather than copied: bit 0 = A, 1 = C, 2 = G, 3 = T,
# a code's letter is the IUPAC symbol of the set of its bits, and '=' is the empty set.
_IUPAC = {"": "=", "A": "A", "C": "C", "G": "G", "T": "T", "AC": "M", "AG": "R", "AT": "W", "CG": "S", "CT": "Y", "GT": "K",
          "ACG": "V", "ACT": "H", "AGT": "D", "CGT": "B", "ACGT": "N"}
NIBBLE = ["".join(b for k, b in enumerate("ACGT") if code >> k & 1) for code in range(16)]
NIBBLE = [_IUPAC[s] for s in NIBBLE]

_INT_TYPES = {"c": (1, True), "C": (1, False), "s": (2, True), "S": (2, False), "i": (4, True), "I": (4, False)}


def _float32(buf, at):
    import struct
    return struct.unpack_from("<f", bytes(buf[at:at + 4]))[0]


def _aux(rec, at):
    """one optional field (SAMv1 4.2.4): tag[2] val_type[1] value; returns (tag, type letter, value, next offset)"""
    tag = bytes(rec[at:at + 2]).decode("ascii")
    typ = chr(rec[at + 2])
    at += 3
    if typ == "A":
        return tag, typ, chr(rec[at]), at + 1
    if typ in _INT_TYPES:
        n, sg = _INT_TYPES[typ]
        return tag, typ, le(rec, at, n, sg), at + n
    if typ == "f":
        return tag, typ, _float32(rec, at), at + 4
    if typ in "ZH":
        end = at
        while rec[end] != 0:
            end += 1
        return tag, typ, bytes(rec[at:end]).decode("ascii"), end + 1
    if typ == "B":
        sub = chr(rec[at])
        count = le(rec, at + 1, 4)
        at += 5
        if sub == "f":
            vals = [_float32(rec, at + 4 * k) for k in range(count)]
            return tag, "B" + sub, vals, at + 4 * count
        need(sub in _INT_TYPES, "unknown array subtype %r" % sub)
        n, sg = _INT_TYPES[sub]
        return tag, "B" + sub, [le(rec, at + n * k, n, sg) for k in range(count)], at + n * count
    raise BamFormatError("unknown optional field type %r" % typ)


def read_bam(raw):
    stream = b"".join(blocks(raw))
    need(stream[:4] == b"BAM\x01", "magic")
    l_text = le(stream, 4, 4)
    text = stream[8:8 + l_text]
    at = 8 + l_text
    n_ref = le(stream, at, 4)
    at += 4
    refs = []
    for _ in range(n_ref):
        l_name = le(stream, at, 4)
        name = stream[at + 4:at + 4 + l_name - 1].decode("ascii")
        refs.append((name, le(stream, at + 4 + l_name, 4)))
        at += 8 + l_name
    out = []
    while at < len(stream):
        block_size = le(stream, at, 4)
        rec = memoryview(stream)[at + 4:at + 4 + block_size]
        need(len(rec) == block_size, "truncated alignment")
        at += 4 + block_size
        a = {"refID": le(rec, 0, 4, True), "pos": le(rec, 4, 4, True), "l_read_name": rec[8], "mapq": rec[9],
             "bin": le(rec, 10, 2), "n_cigar_op": le(rec, 12, 2), "flag": le(rec, 14, 2), "l_seq": le(rec, 16, 4),
             "next_refID": le(rec, 20, 4, True), "next_pos": le(rec, 24, 4, True), "tlen": le(rec, 28, 4, True)}
        p = 32
        need(rec[p + a["l_read_name"] - 1] == 0, "read_name is not NUL-terminated")
        a["read_name"] = bytes(rec[p:p + a["l_read_name"] - 1]).decode("ascii")
        p += a["l_read_name"]
        cigar = []
        for k in range(a["n_cigar_op"]):
            v = le(rec, p + 4 * k, 4)
            cigar.append((v >> 4, "MIDNSHP=X"[v & 15]))
        p += 4 * a["n_cigar_op"]
        a["cigar"] = cigar
        n_seq_bytes = (a["l_seq"] + 1) // 2
        seq = []
        for k in range(a["l_seq"]):
            byte = rec[p + k // 2]
            seq.append(NIBBLE[byte >> 4 if k % 2 == 0 else byte & 15])   # "the high nibble first"
        if a["l_seq"] % 2:
            need(rec[p + n_seq_bytes - 1] & 15 == 0, "padding nibble of an odd-length sequence is not zero")
        a["seq"] = "".join(seq)
        p += n_seq_bytes
        a["qual"] = bytes(rec[p:p + a["l_seq"]])          # phred values, 0xFF x l_seq when absent
        p += a["l_seq"]
        aux = []
        while p < block_size:
            tag, typ, val, p = _aux(rec, p)
            aux.append((tag, typ, val))
        need(p == block_size, "optional fields overrun the record")
        a["aux"] = aux
        out.append(a)
    return text, refs, out


def reg2bin(beg, end):
    """SAMv1 5.3, the specification's own C function"""
    end -= 1
    if beg >> 14 == end >> 14:
        return ((1 << 15) - 1) // 7 + (beg >> 14)
    if beg >> 17 == end >> 17:
        return ((1 << 12) - 1) // 7 + (beg >> 17)
    if beg >> 20 == end >> 20:
        return ((1 << 9) - 1) // 7 + (beg >> 20)
    if beg >> 23 == end >> 23:
        return ((1 << 6) - 1) // 7 + (beg >> 23)
    if beg >> 26 == end >> 26:
        return ((1 << 3) - 1) // 7 + (beg >> 26)
    return 0


def sam_fields(a, refs):
    """QNAME FLAG RNAME POS MAPQ CIGAR RNEXT PNEXT TLEN SEQ QUAL of SAMv1 1.4 from the BAM fields (pos is 0-based in BAM)"""
    rname = refs[a["refID"]][0] if a["refID"] >= 0 else "*"
    if a["next_refID"] < 0:
        rnext = "*"
    elif a["next_refID"] == a["refID"]:
        rnext = "="
    else:
        rnext = refs[a["next_refID"]][0]
    cigar = "".join("%d%s" % c for c in a["cigar"]) or "*"
    qual = "*" if a["l_seq"] and all(q == 255 for q in a["qual"]) else "".join(chr(q + 33) for q in a["qual"])
    return [a["read_name"], str(a["flag"]), rname, str(a["pos"] + 1), str(a["mapq"]), cigar, rnext, str(a["next_pos"] + 1),
            str(a["tlen"]), a["seq"] or "*", qual or "*"]


def smallest_int_type(v):
    """the integer type samtools / htslib choose when they convert SAM text `i` to BAM: the smallest that holds the value"""
    if v < 0:
        return "c" if v >= -(1 << 7) else "s" if v >= -(1 << 15) else "i"
    return "C" if v < (1 << 8) else "S" if v < (1 << 16) else "I"
