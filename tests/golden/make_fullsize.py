#!/usr/bin/env python3
"""Generate tests/golden/fullsize.json by RUNNING THE REFERENCE ITSELF (oracle/_ref/pbsim_ref_philox: the unmodified
pbsim.cpp with oracle/ref_shim.h force-included) on the BASELINE-size cases of fullsize_cases.py.

The outputs are too large to keep (63 GB of text for one 750 Mbp record at depth 20), so `gzip` and `samtools` are replaced
on PATH by tests/golden/crcsum.c, which leaves "<crc32> <bytes>" of the text the reference piped into it in the output
file.  Those digests + the stderr report are what tests/test_gpu_fullsize.py compares the GPU job's folded member CRCs with.

  python tests/golden/make_fullsize.py [case ...]      (all cases: ~3 CPU-hours, run side by side: the 750 Mbp x depth 60 case alone takes 93 minutes)

Needs /root/reference (through oracle/_ref) and ~3 GB of memory per case.  Only digests are committed.
"""
import json
import os
import subprocess
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import harness  # noqa: E402
from fullsize_cases import FULLSIZE  # noqa: E402

OUT = os.path.join(HERE, "fullsize.json")


def write_fasta(path, length, seed):
    import numpy as np
    seq = harness.synth_bases(length, seed)
    with open(path, "wb") as f:
        f.write(b">synth_%d_%d\n" % (length, seed))
        width = 80
        full = length // width * width
        step = width * (1 << 16)
        for a in range(0, full, step):
            rows = seq[a:min(full, a + step)].reshape(-1, width)
            f.write(np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1).tobytes())
        if full < length:
            f.write(seq[full:].tobytes() + b"\n")


def run_case(name, stubs):
    case = FULLSIZE[name]
    length, seed = case["record"]
    args = harness.resolve(case["args"])
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir="/dev/shm") as td:
        fa = os.path.join(td, "g.fa")
        write_fasta(fa, length, seed)
        env = dict(os.environ, PATH=stubs + ":" + os.environ["PATH"], PBSHIM_SEED=args[args.index("--seed") + 1], PBSHIM_MODE="philox")
        p = subprocess.run([harness.REF_PHILOX] + args + ["--genome", fa, "--prefix", os.path.join(td, "out")], env=env,
                           capture_output=True, text=True, check=True, cwd=td)
        entry = {"record": [length, seed], "args": case["args"], "stderr": harness.strip_report(p.stderr)}
        for fn in sorted(os.listdir(td)):
            if not fn.startswith("out_0001") or fn.endswith(".ref"):
                continue
            key = fn[len("out_0001"):].replace(".fq.gz", ".fq").replace(".maf.gz", ".maf").replace(".bam", ".sam")
            with open(os.path.join(td, fn)) as f:
                crc, n = f.read().split()
            entry[key] = {"crc32": crc, "bytes": int(n)}
    entry["reference_seconds"] = round(time.time() - t0)
    print(name, {k: v for k, v in entry.items() if k != "stderr"}, flush=True)
    return name, entry


def main():
    names = sys.argv[1:] or list(FULLSIZE)
    subprocess.run(["make", "-s", "-C", os.path.join(harness.ROOT, "oracle")], check=True)
    stubs = tempfile.mkdtemp(prefix="crcsum_stubs_")
    exe = os.path.join(stubs, "crcsum")
    subprocess.run(["cc", "-O2", os.path.join(HERE, "crcsum.c"), "-lz", "-o", exe], check=True)
    for n in ("gzip", "samtools"):
        os.symlink(exe, os.path.join(stubs, n))
    done = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            done = json.load(f)
    with ThreadPoolExecutor(max_workers=len(names)) as ex:
        for name, entry in ex.map(lambda n: run_case(n, stubs), names):
            done[name] = entry
            with open(OUT, "w") as f:
                json.dump(done, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
