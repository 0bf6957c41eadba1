"""CPU: the executable specification of the GPU DEFLATE encoder (tests/deflate_model.py) produces streams zlib
accepts -- gzip members with correct CRC32/ISIZE, BGZF framing, every code path of the format (stored fallback,
distance-1 matches, the depth limiter, the per-call code table whose repeated lengths are coded with 16)."""
import gzip
import os
import random
import struct
import zlib

import pytest

import deflate_model as D


def dna(n, seed):
    r = random.Random(seed)
    return bytes(r.choice(b"ACGT") for _ in range(n))


def fib_skew():
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    s = list(b"".join(bytes([65 + i]) * f for i, f in enumerate(fib)))
    random.Random(5).shuffle(s)
    return bytes(s[:32768])


CASES = {
    "one": b"A", "run4": b"AAAA", "run129": b"A" * 129, "bang40000": b"!" * 40000, "rand": os.urandom(3000),
    "dna": dna(40000, 1), "allbytes": bytes(range(256)) * 10, "ramps": b"".join(bytes([i]) * (i + 1) for i in range(256)),
    "fib": fib_skew(), "chunk+1": dna(32769, 4),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_model_streams_inflate(name):
    data = CASES[name]
    z = D.compress(data)
    assert gzip.decompress(z) == data
    p, k = 0, 0
    while p < len(z):                      # BGZF walk: SAMv1 4.1
        assert z[p:p + 4] == b"\x1f\x8b\x08\x04" and z[p + 12:p + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", z, p + 16)[0] + 1
        part = zlib.decompress(z[p + 18:p + bsize - 8], -15)
        assert part == data[k:k + D.CHUNK]
        k += len(part)
        p += bsize
    assert p == len(z) and k == len(data)


def test_depth_limiter_engages():
    freq = [0] * 286
    a, b = 1, 1
    for i in range(30):
        freq[i] = a
        a, b = b, a + b
    lens = D.huffman_lengths(freq)
    used = [l for l in lens if l]
    assert max(used) == 15 and sum(2.0 ** -l for l in used) <= 1.0 + 1e-12


def test_fixed_code_length_code_is_complete():
    assert sum(2.0 ** -l for l in D.CL_LEN) == 1.0 and len(D.CL_LEN) == 19


def test_stored_fallback_for_incompressible_input():
    data = os.urandom(5000)
    z = D.member(data, D.build_table(data))
    assert z[18] == 0x01 and len(z) == 18 + 5 + len(data) + 8


def test_symbols_unseen_in_the_sample_stay_encodable():
    """the code is fitted to the call's first SAMPLE_CHUNKS chunks; every symbol keeps a code (floor of one occurrence),
    so text that changes character later (here: random bytes behind 2.1 MB of DNA) still round-trips"""
    data = dna(D.SAMPLE_CHUNKS * D.CHUNK + 5000, 3) + os.urandom(70000) + b"!" * 5000
    z = D.compress(data)
    assert gzip.decompress(z) == data
