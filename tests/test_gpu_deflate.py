"""The on-GPU DEFLATE encoder (pbsim3_amd/csrc/deflate.hip) against zlib and against its executable
specification tests/deflate_model.py: every member must inflate to the input (CRC32 and ISIZE checked by
gzip), obey the BGZF container rules, and equal the model's bytes."""
import gzip
import os
import random
import struct
import zlib

import pytest

import deflate_model
import harness
import product
import pbsim3_amd as P
from cases import CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    with P.Context(P.default_params(), 0) as c:
        yield c


def members(raw):
    p, out = 0, []
    while p < len(raw):
        assert raw[p:p + 4] == b"\x1f\x8b\x08\x04" and raw[p + 10:p + 16] == b"\x06\x00BC\x02\x00"
        bsize = struct.unpack_from("<H", raw, p + 16)[0] + 1
        data = zlib.decompress(raw[p + 18:p + bsize - 8], -15)
        crc, isize = struct.unpack_from("<II", raw, p + bsize - 8)
        assert isize == len(data) and crc == (zlib.crc32(data) & 0xffffffff)
        out.append(data)
        p += bsize
    assert p == len(raw)
    return out


def dna(n, seed):
    r = random.Random(seed)
    return bytes(r.choice(b"ACGT") for _ in range(n))


def fib_skew():
    fib = [1, 1]
    while len(fib) < 24:
        fib.append(fib[-1] + fib[-2])
    s = list(b"".join(bytes([65 + i]) * f for i, f in enumerate(fib)))
    random.Random(5).shuffle(s)
    return bytes(s[:32768])          # forces Huffman depths beyond 15 -> the length limiter


SMALL = {
    "one": b"A", "two": b"AB", "run4": b"AAAA", "run127": b"A" * 127, "run128": b"A" * 128, "run129": b"A" * 129,
    "run300": b"!" * 300, "bang40000": b"!" * 40000, "rand1000": os.urandom(1000), "rand40000": os.urandom(40000),
    "dna50000": dna(50000, 1), "allbytes": bytes(range(256)) * 10,
    "ramps": b"".join(bytes([i]) * (i + 1) for i in range(256)), "fib": fib_skew(),
    "chunk-1": dna(32767, 2), "chunk": dna(32768, 3), "chunk+1": dna(32769, 4),
    "fastq_like": b"".join(b"@S1_%d\n" % i + dna(900 + 37 * i, i) + b"\n+S1_%d\n" % i + b"!" * (900 + 37 * i) + b"\n"
                           for i in range(40)),
}


@pytest.mark.parametrize("name", sorted(SMALL))
def test_matches_model_and_zlib(ctx, name):
    data = SMALL[name]
    got = ctx.deflate_buffer(data)
    assert gzip.decompress(got) == data
    parts = members(got)
    assert len(parts) == (len(data) + 32767) // 32768 and all(len(p) <= 32768 for p in parts)
    assert got == deflate_model.compress(data)


def test_empty(ctx):
    assert ctx.deflate_buffer(b"") == b""


def test_large_multi_piece(ctx):
    """> one launch piece (8192 chunks = 256 MiB): piece seams, the double-buffered copy, a ragged tail"""
    block = dna(1 << 20, 9) + b"!" * 4096 + os.urandom(1 << 16)
    data = (block * 262)[: 300 * (1 << 20) + 12345]
    got = ctx.deflate_buffer(data)
    assert len(got) < 0.45 * len(data)
    assert gzip.decompress(got) == data


@pytest.mark.parametrize("case", ["wgs_errhmm_rsii_default", "wgs_qshmm_rsii_pass3", "wgs_errhmm_ont_hpbias5"])
def test_simulation_sinks_deflated(case):
    """pbsim_set_deflate: the sinks' members inflate to exactly the text of the plain run"""
    args = harness.resolve(CASES[case]["args"])
    plain, _ = product.run_wgs(args)
    packed, _ = product.run_wgs(args, deflate=True)
    assert set(plain) == set(packed)
    for k, v in plain.items():
        g = packed[k]
        assert gzip.decompress(g) == v, k
        assert len(g) < 0.5 * len(v)


def test_randomized_buffers(ctx):
    """60 random buffers: sizes around the chunk / segment seams, alphabets from 1 to 256 symbols, geometric run
    lengths, skewed symbol frequencies.  Every stream must inflate to its input; the small ones must equal the model."""
    r = random.Random(2026)
    seams = [1, 2, 3, 4, 5, 127, 128, 129, 255, 256, 257, 4095, 32767, 32768, 32769, 65535, 65536, 65537]
    for it in range(60):
        n = r.choice(seams) if it < len(seams) else r.randint(1, 150000)
        k = r.choice([1, 2, 4, 5, 16, 64, 256])
        alphabet = bytes(r.sample(range(256), k))
        weights = [r.random() ** r.choice([1, 4, 12]) + 1e-9 for _ in alphabet]
        run_p = r.choice([0.0, 0.3, 0.9, 0.99])
        out = bytearray()
        while len(out) < n:
            b = r.choices(alphabet, weights)[0]
            run = 1
            while r.random() < run_p and run < 2000:
                run += 1
            out.extend(bytes([b]) * run)
        data = bytes(out[:n])
        got = ctx.deflate_buffer(data)
        assert gzip.decompress(got) == data, (it, n, k, run_p)
        if n <= 70000:
            assert got == deflate_model.compress(data), (it, n, k, run_p)


def test_contexts_created_and_destroyed_in_a_loop():
    """the look-back of k_deflate_chunks trusts any status word with the launch's epoch and a flag, the status
    array is never cleared between launches, and a lane's epochs restart with every context -- while hipMalloc hands the
    freed array of the context before back with that context's words in it.  A new array now starts from zeros: contexts in
    a loop, each compressing buffers of several launches' worth of chunks (its epochs 1..n meet the words the one before
    left at the same epochs), must keep producing streams that inflate to their input."""
    import hashlib
    block = dna(1 << 18, 21) + b"!" * 5000
    for it in range(8):
        with P.Context(P.default_params(), 0) as c:
            for k in range(3):
                n = (5 + 7 * ((it + k) % 4)) * (1 << 20) + 1000 * it + k
                data = (block * (n // len(block) + 1))[:n]
                got = c.deflate_buffer(data)
                assert hashlib.sha1(gzip.decompress(got)).digest() == hashlib.sha1(data).digest(), (it, k, n)
