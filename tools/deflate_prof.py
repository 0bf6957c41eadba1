"""GPU box: phase profile of the deflate kernel on FASTQ-like and MAF-like text (PBSIM_DEFLATE_PROF)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "NOPROF" not in os.environ:
    os.environ["PBSIM_DEFLATE_PROF"] = "1"
import numpy as np
import pbsim3_amd as P
rng = np.random.default_rng(1)
acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
L = 10000
recs = []
for i in range(3000):
    s = acgt[rng.integers(0, 4, L)].tobytes()
    recs.append(b"@S1_%d\n" % i + s + b"\n+S1_%d\n" % i + b"!" * L + b"\n")
fq = b"".join(recs)
maf = b"".join(b"a\ns ref 12345 %d + 100000000 " % L + acgt[rng.integers(0, 4, L)].tobytes() + b"\ns S1_%d 0 %d + %d " % (i, L, L)
               + acgt[rng.integers(0, 4, L)].tobytes() + b"\n\n" for i in range(3000))
with P.Context(P.default_params(), 0) as ctx:
    for name, data in (("fastq", fq), ("maf", maf)):
        ctx.deflate_buffer(data[:1 << 20])
        t = time.perf_counter()
        z = ctx.deflate_buffer(data)
        dt = time.perf_counter() - t
        print(name, len(data), "->", len(z), "ratio %.3f" % (len(z) / len(data)), "host wall %.1f ms" % (dt * 1e3))
