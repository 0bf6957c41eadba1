#!/usr/bin/env python3
"""Generate tests/golden/manifest.json (+ a few full outputs) by RUNNING THE
REFERENCE ITSELF in this container:

  glibc  mode -> oracle/_ref/pbsim_ref          (unmodified pbsim.cpp)
  philox mode -> oracle/_ref/pbsim_ref_philox   (same source + oracle/ref_shim.h)

`gzip` and `samtools` are replaced on PATH by `cat` stubs because the
reference reaches them through popen("gzip > f") (pbsim.cpp:709-730); the
decompressed bytes are what the goldens pin.  Needs /root/reference (models)
and `make -C oracle`.  The reference never travels: only hashes and small
output vectors are committed.
"""
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from cases import CASES, FULL, MODES  # noqa: E402

REF_DATA = "/root/reference/data"


def strip_report(err: str) -> str:
    """Drop run-dependent lines: file paths, prefix, wall/CPU time."""
    keep = []
    for line in err.splitlines():
        if line.startswith((":::: System utilization", "CPU time(s)", "Elapsed time(s)")):
            continue
        if line.split(" : ")[0] in ("prefix", "genome", "transcript", "errhmm", "qshmm", "file name", "template"):
            continue
        keep.append(line)
    return "\n".join(keep).rstrip("\n") + "\n"


def resolve(args, inputs_dir, model_dir):
    out = []
    for a in args:
        if a.startswith("MODEL:"):
            out.append(os.path.join(model_dir, a[6:]))
        elif a.startswith("INPUT:"):
            out.append(os.path.join(inputs_dir, a[6:]))
        else:
            out.append(a)
    return out


def make_stubs(d):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "gzip"), "w") as f:
        f.write("#!/bin/sh\nexec cat\n")
    with open(os.path.join(d, "samtools"), "w") as f:
        f.write('#!/bin/sh\nexec cat > "$4"\n')
    for n in ("gzip", "samtools"):
        os.chmod(os.path.join(d, n), 0o755)


def canonical_outputs(workdir, prefix):
    """name -> bytes, with reference file names mapped to plain-text names."""
    res = {}
    for fn in sorted(os.listdir(workdir)):
        if not fn.startswith(prefix):
            continue
        key = fn[len(prefix):]
        key = key.replace(".fq.gz", ".fq").replace(".maf.gz", ".maf").replace(".bam", ".sam")
        with open(os.path.join(workdir, fn), "rb") as f:
            res[key] = f.read()
    return res


def run_reference(case, mode, workdir):
    exe = os.path.join(ROOT, "oracle", "_ref", "pbsim_ref" if mode == "glibc" else "pbsim_ref_philox")
    args = resolve(CASES[case]["args"], os.path.join(HERE, "inputs"), REF_DATA)
    seed = args[args.index("--seed") + 1]
    stubs = os.path.join(workdir, "stubs")
    make_stubs(stubs)
    env = dict(os.environ, PATH=stubs + ":" + os.environ["PATH"], PBSHIM_SEED=seed, PBSHIM_MODE="philox")
    p = subprocess.run([exe] + args + ["--prefix", os.path.join(workdir, "out")], env=env,
                       capture_output=True, text=True, check=True)
    outs = canonical_outputs(workdir, "out")
    outs[".stderr"] = strip_report(p.stderr).encode()
    return outs


def main():
    manifest = {}
    full_dir = os.path.join(HERE, "full")
    shutil.rmtree(full_dir, ignore_errors=True)
    os.makedirs(full_dir)
    for case in CASES:
        for mode in MODES:
            with tempfile.TemporaryDirectory() as td:
                outs = run_reference(case, mode, td)
            entry = {k: {"sha256": hashlib.sha256(v).hexdigest(), "bytes": len(v)} for k, v in outs.items()}
            manifest[f"{case}/{mode}"] = entry
            if case in FULL:
                for k, v in outs.items():
                    if k.endswith(".ref"):
                        continue
                    with gzip.GzipFile(os.path.join(full_dir, f"{case}.{mode}{k}.gz"), "wb", mtime=0) as g:
                        g.write(v)
            print(case, mode, {k: e["bytes"] for k, e in entry.items()})
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
