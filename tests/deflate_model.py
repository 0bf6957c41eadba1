"""Executable specification of the on-GPU DEFLATE encoder (pbsim3_amd/csrc/deflate.hip), test infrastructure only.

The product compresses its FASTQ/SAM/BAM/MAF text on the GPU into BGZF-framed gzip members (RFC 1952 + SAMv1 4.1):
one member per CHUNK input bytes, one dynamic-Huffman block (RFC 1951 3.2.7) per member.  This file restates that
encoder step by step in plain Python so that (a) the bit-level format decisions (fixed code-length code, run-length
header, length limiting, distance-1 matches) are checked against zlib on the CPU, and (b) the GPU output can be compared
byte for byte, not only through a decompress round trip.

    tokens   : each THREAD owns SEG consecutive bytes; a byte equal to its predecessor (distance-1 match candidate)
               extends a run, runs of >= 3 become one match (length <= SEG), shorter runs stay literals
    lit/len  : Huffman over the chunk's histogram (two-queue merge on the (freq, symbol)-sorted leaves), depths folded
               to <= 15 by moving leaves down from the shortest deeper level, lengths re-dealt in sorted order
    header   : HLIT = last used symbol + 1, HDIST = 1, HCLEN = 19; the code-length alphabet uses a FIXED complete code
               (lengths CL_LEN below); zeros are run-length coded with 17/18 only, 16 is never produced
    fallback : a stored block when the Huffman block would not be smaller
"""
import struct
import zlib

CHUNK = 32768
SEG = 128
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
# fixed code for the code-length alphabet: 3 symbols of 3 bits, 4 of 4 bits, 12 of 5 bits (Kraft sum exactly 1)
CL_LEN = [3, 5, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 4, 5, 5, 5, 5, 3, 3]   # index = symbol 0..18


def canonical(lengths):
    """RFC 1951 3.2.2 canonical codes, returned bit-reversed (DEFLATE packs Huffman codes MSB first)."""
    maxl = max(lengths) if lengths else 0
    cnt = [0] * (maxl + 2)
    for l in lengths:
        if l:
            cnt[l] += 1
    nxt = [0] * (maxl + 2)
    code = 0
    for b in range(1, maxl + 1):
        code = (code + cnt[b - 1]) << 1
        nxt[b] = code
    out = []
    for l in lengths:
        if l == 0:
            out.append(0)
            continue
        c = nxt[l]
        nxt[l] += 1
        out.append(int(format(c, "0%db" % l)[::-1], 2))
    return out


CL_CODE = canonical(CL_LEN)


def length_symbol(L):
    v = L - 3
    if v < 8:
        return 257 + v, 0, 0
    if L == 258:
        return 285, 0, 0
    e = v.bit_length() - 3
    return 261 + 4 * e + ((v >> e) & 3), e, v & ((1 << e) - 1)


def tokenize(data):
    """list of ('lit', byte) / ('match', length), thread segments of SEG bytes"""
    toks = []
    n = len(data)
    for s in range(0, n, SEG):
        run = 0
        for i in range(s, min(s + SEG, n)):
            b = data[i]
            if i > 0 and b == data[i - 1]:
                run += 1
                continue
            toks.extend(flush(run, data[i - 1] if i else 0))
            run = 0
            toks.append(("lit", b))
        toks.extend(flush(run, data[min(s + SEG, n) - 1]))
    return toks


def flush(run, byte):
    if run >= 3:
        return [("match", run)]
    return [("lit", byte)] * run


def huffman_lengths(freq, limit=15):
    used = sorted((f, s) for s, f in enumerate(freq) if f)
    n = len(used)
    lengths = [0] * len(freq)
    if n == 1:
        lengths[used[0][1]] = 1
        return lengths
    leaf_parent = [0] * n
    w = [0] * (n - 1)
    par = [0] * (n - 1)
    leaf = root = 0
    for nxt in range(n - 1):
        tot = 0
        for _ in range(2):
            take_leaf = leaf < n and (root >= nxt or used[leaf][0] <= w[root])
            if take_leaf:
                tot += used[leaf][0]
                leaf_parent[leaf] = nxt
                leaf += 1
            else:
                tot += w[root]
                par[root] = nxt
                root += 1
        w[nxt] = tot
    depth = [0] * (n - 1)
    for i in range(n - 3, -1, -1):
        depth[i] = depth[par[i]] + 1
    cnt = [0] * (limit + 1)
    for i in range(n):
        d = depth[leaf_parent[i]] + 1
        cnt[min(d, limit)] += 1
    total = sum(cnt[l] << (limit - l) for l in range(1, limit + 1))
    while total > (1 << limit):
        cnt[limit] -= 1
        for l in range(limit - 1, 0, -1):
            if cnt[l]:
                cnt[l] -= 1
                cnt[l + 1] += 2
                break
        total -= 1
    i = 0
    for l in range(limit, 0, -1):      # ascending frequency gets the longest codes
        for _ in range(cnt[l]):
            lengths[used[i][1]] = l
            i += 1
    return lengths


class Bits:
    def __init__(self):
        self.acc = 0
        self.n = 0

    def put(self, v, k):
        self.acc |= v << self.n
        self.n += k

    def bytes(self):
        return self.acc.to_bytes((self.n + 7) // 8, "little")


def header_symbols(seq):
    """code-length symbols (sym, extra_bits, extra_value) for the lengths sequence; zeros use 17/18 only"""
    out = []
    i = 0
    while i < len(seq):
        if seq[i]:
            out.append((seq[i], 0, 0))
            i += 1
            continue
        j = i
        while j < len(seq) and seq[j] == 0:
            j += 1
        r = j - i
        while r >= 11:
            t = min(r, 138)
            out.append((18, 7, t - 11))
            r -= t
        if r >= 3:
            out.append((17, 3, r - 3))
            r = 0
        out.extend([(0, 0, 0)] * r)
        i = j
    return out


def deflate_block(data):
    toks = tokenize(data)
    freq = [0] * 286
    any_match = False
    for kind, v in toks:
        if kind == "lit":
            freq[v] += 1
        else:
            freq[length_symbol(v)[0]] += 1
            any_match = True
    freq[256] = 1
    lens = huffman_lengths(freq)
    codes = canonical(lens)
    hlit = max(s for s in range(286) if lens[s]) + 1
    hlit = max(hlit, 257)
    b = Bits()
    b.put(1, 1)
    b.put(2, 2)
    b.put(hlit - 257, 5)
    b.put(0, 5)
    b.put(15, 4)
    for s in CL_ORDER:
        b.put(CL_LEN[s], 3)
    for sym, eb, ev in header_symbols(lens[:hlit] + [1 if any_match else 0]):
        b.put(CL_CODE[sym], CL_LEN[sym])
        b.put(ev, eb)
    for kind, v in toks:
        if kind == "lit":
            b.put(codes[v], lens[v])
        else:
            s, eb, ev = length_symbol(v)
            b.put(codes[s], lens[s])
            b.put(ev, eb)
            b.put(0, 1)        # the single distance code (distance 1), one bit
    b.put(codes[256], lens[256])
    out = b.bytes()
    if len(out) >= len(data) + 5:
        n = len(data)
        out = b"\x01" + struct.pack("<HH", n, n ^ 0xFFFF) + data
    return out


def member(data):
    body = deflate_block(data)
    total = 18 + len(body) + 8
    return (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", total - 1) + body +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


def compress(data):
    return b"".join(member(data[i:i + CHUNK]) for i in range(0, len(data), CHUNK))
