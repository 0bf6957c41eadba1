"""CPU-side checks (no GPU): the C-ABI library builds for gfx950, loads, and
exports every symbol include/pbsim3_amd.h declares; Philox known answers; the
host-built integer tables against known answers of the reference."""
import ctypes as C
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import harness
import pbsim3_amd as P

ROOT = harness.ROOT
HEADER = os.path.join(ROOT, "include", "pbsim3_amd.h")

# Random123 known-answer vectors for Philox4x32-10 (kat_vectors)
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pbsim_[a-z0-9_]+)\s*\(", src)) - {"pbsim_ctx", "pbsim_sink"})


def test_library_exports_every_declared_symbol():
    lib = P.load()
    names = declared_functions()
    assert len(names) >= 25
    bound = {n for n, _, _ in P.API}
    for n in names:
        assert hasattr(lib, n), n
        assert n in bound, f"{n} is declared in the header but not bound in pbsim3_amd.API"


def test_fat_binary_is_gfx950_only():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", P.lib_path()],
                         capture_output=True, text=True, cwd="/tmp").stdout
    archs = set(re.findall(r"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", out))
    d = os.path.dirname(P.lib_path())   # llvm-objdump drops the extracted bundles next to the input
    for f in os.listdir(d):
        if f.startswith(os.path.basename(P.lib_path()) + "."):
            os.remove(os.path.join(d, f))
    assert archs == {"gfx950"}, archs


def test_product_philox_known_answers():
    lib = P.load()
    for ctr, key, want in KAT:
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        lib.pbsim_philox4x32_10(c, k, o)
        assert tuple(o) == want


def test_oracle_philox_known_answers(tmp_path):
    src = tmp_path / "kat.c"
    lines = ['#include <stdio.h>', f'#include "{ROOT}/oracle/philox4x32.h"', "int main(){uint32_t o[4];"]
    for ctr, key, _ in KAT:
        lines.append("{uint32_t c[4]={%s},k[2]={%s}; orc_philox4x32_10(c,k,o); printf(\"%%08x %%08x %%08x %%08x\\n\",o[0],o[1],o[2],o[3]);}"
                     % (",".join("0x%xu" % x for x in ctr), ",".join("0x%xu" % x for x in key)))
    lines.append("return 0;}")
    src.write_text("\n".join(lines))
    exe = tmp_path / "kat"
    subprocess.run(["gcc", "-O1", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split("\n")
    for (_, _, want), line in zip(KAT, out):
        assert tuple(int(x, 16) for x in line.split()) == want


def test_no_device_context_refuses_compute():
    ctx = P.Context(P.default_params(), -1)
    with pytest.raises(P.PbsimError, match="no HIP device|no CPU fallback"):
        ctx.set_reference(b"ACGT" * 100, 1)
    with pytest.raises(P.PbsimError):
        ctx.batch_walk(1, 10)
    ctx.close()


def test_header_tables_known_answers():
    """SURVEY A.1 known answers measured on the compiled reference: len_rv 99911,
    acc_rv 100000, classes 63..89, sha1 of prob2len as int64[100001]."""
    ctx = P.Context(P.default_params(), -1)
    p2l = np.frombuffer(ctx.dump_table(0), dtype=np.int32)
    p2a = np.frombuffer(ctx.dump_table(1), dtype=np.uint8)
    assert len(p2l) - 1 == 99911 and len(p2a) - 1 == 100000
    full = np.zeros(100001, dtype=np.int64)
    full[:len(p2l)] = p2l
    assert hashlib.sha1(full.tobytes()).hexdigest() == "2c438b3380a6edb5223a735864aac8812a2d9d05"
    full = np.zeros(100001, dtype=np.int64)
    full[:len(p2a)] = p2a
    assert hashlib.sha1(full.tobytes()).hexdigest() == "08e84ce834ae5e251c1f9fda9b1a39ab3e4049ac"
    counts = np.bincount(p2a[1:], minlength=101)
    assert counts[89] == 19800 and counts[88] == 15890 and counts[63] == 65
    assert p2a[1:].min() == 63 and p2a[1:].max() == 89
    ctx.close()


@pytest.mark.parametrize("model", ["ERRHMM-RSII", "ERRHMM-SEQUEL", "ERRHMM-ONT", "ERRHMM-ONT-HQ"])
def test_errhmm_class_tables_are_well_formed(model):
    """Every transition modulus of the shipped models is 1000, emission
    moduli are 0..1000 (near-pure-deletion states), packed CDF rows are monotone."""
    ctx = P.Context(P.default_params(), -1)
    ctx.load_errhmm(harness.model_path(model + ".model"))
    blob = ctx.dump_table(2)
    ncls = 89 - 63 + 1
    assert len(blob) % ncls == 0
    stride = len(blob) // ncls
    for c in range(ncls):
        b = blob[c * stride:(c + 1) * stride]
        hdr = np.frombuffer(b[:64], dtype=np.uint32)
        smax, init_rv, mode, mag, acc = (int(x) for x in hdr[:5])
        assert acc == 63 + c and mode in (0, 1, 2)
        assert init_rv == 1000
        rows = np.frombuffer(b[64:64 + 32 * (smax + 1)], dtype=np.uint16).reshape(smax + 1, 16)
        # emission rows: 16 B per state {magic u32, shift u8 | (2^24 - d) << 8, E0' u16, E1' u16, del_thr[hp 1] u16, del_thr[hp 11] u16}:
        # z % d by multiply-high and a 24-bit multiply-add (the low 24 bits of z + quo * (2^24 - d)), and the two deletion
        # thresholds the default bias can reach
        emis_off = 64 + 32 * (smax + 1)
        em = np.frombuffer(b[emis_off:emis_off + 16 * (smax + 1)], dtype=np.uint32).reshape(smax + 1, 4)
        init_off = emis_off + 16 * (smax + 1)
        init = np.frombuffer(b[init_off:init_off + 1000], dtype=np.uint8)
        # the transition rows follow the initial-state table without a gap (the walk indexes it as row 0)
        assert stride == (init_off + 1000 + 1000 * smax + 15) // 16 * 16
        assert init.min() >= 1 and init.max() <= smax
        assert (np.diff(init.astype(int)) >= 0).all()
        for j in range(1, smax + 1):
            tran_rv, emis_rv, e0, e1 = (int(x) for x in rows[j][:4])
            if tran_rv == 0:
                continue
            assert tran_rv == 1000 and 0 <= emis_rv <= 1000 and e0 <= e1 <= emis_rv
            magic, shift, neg_d = int(em[j][0]), int(em[j][1]) & 0xff, int(em[j][1]) >> 8
            d = (1 << 24) - neg_d
            t0, t1 = int(em[j][2]) & 0xffff, int(em[j][2]) >> 16
            if emis_rv >= 2:
                assert (d, t0, t1) == (emis_rv, e0, e1)
            else:                                  # emis_rv 0: rand() % 3; emis_rv 1: constant class
                assert d == 3
            for z in (0, 1, d - 1, d, 999, 1000, 123456789, 2 ** 31 - 1, (2 ** 31 - 1) // d * d, (2 ** 31 - 1) // d * d - 1):
                quo = (z * magic >> 32) >> shift
                assert z - quo * d == z % d
                assert (z + (quo & 0xffffff) * neg_d) & 0xffffff == z % d      # what the walk kernels compute
                r = z % d
                want = z % 3 if emis_rv == 0 else ((z % emis_rv + 1 > e0) + (z % emis_rv + 1 > e1))
                assert (r >= t0) + (r >= t1) == want
            assert rows[j][4 + 11] == 0           # Q1: hp 11 -> bias 0.0 -> no HMM deletion
            assert int(em[j][3]) == int(rows[j][4 + 1]) | (int(rows[j][4 + 11]) << 16)
            assert rows[j][4 + 1] == rows[j][4 + 10]  # default --hp-del-bias 1
    ctx.close()


def test_qshmm_states_above_state_max_follow_the_reference_layout(tmp_path):
    """SURVEY Q7, pinned: a QSHMM entry with state > STATE_MAX lands 51 slots further on in the reference's flat
    ip|ep|tp block (pbsim.cpp:160-166, 5606-5626).  `80 IP 52 x` is ip[81][1]: invisible to class 80, the initial state of
    class 81.  Only a write past the end of tp[] (class 100, it would hit exist_hmm[]) is refused."""
    ep = " ".join(["0"] * 20 + ["1.0"] + ["0"] * 70)
    base = "81 EP 1 %s\n81 TP 1 1.0\n80 IP 1 1.0\n80 EP 1 %s\n80 TP 1 1.0\n" % (ep, ep)
    a, b = tmp_path / "a.model", tmp_path / "b.model"
    a.write_text(base + "81 IP 1 1.0\n")
    b.write_text(base + "80 IP 52 1.0\n")            # ip[80][52] == ip[81][1]
    blobs = []
    for f in (a, b):
        ctx = P.Context(P.default_params(method=P.METHOD_QS), -1)
        ctx.load_qshmm(str(f))
        blobs.append(ctx.dump_table(2))
        ctx.close()
    assert blobs[0] == blobs[1]
    bad = tmp_path / "bad.model"
    bad.write_text("100 TP 56 " + " ".join(["0.1"] * 56) + "\n")
    ctx = P.Context(P.default_params(method=P.METHOD_QS), -1)
    with pytest.raises(P.PbsimError, match="past qshmm.tp"):
        ctx.load_qshmm(str(bad))
    ctx.close()
