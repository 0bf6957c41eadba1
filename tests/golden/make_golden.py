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


sys.path.insert(0, os.path.dirname(HERE))
import harness  # noqa: E402  (same canonicalisation as the tests)


def run_reference(case, mode, workdir):
    # the six model files of the cases are committed under tests/golden/models (copies of /root/reference/data)
    return harness.run_reference(CASES[case]["args"], mode, workdir, case=CASES[case])


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
