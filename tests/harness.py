"""Test harness: runs the CHECKER (oracle/pbsim_oracle, the compiled reference
under oracle/_ref when present) and canonicalises outputs for comparison.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it."""
import hashlib
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE = os.path.join(ROOT, "oracle", "pbsim_oracle")
REF_GLIBC = os.path.join(ROOT, "oracle", "_ref", "pbsim_ref")
REF_PHILOX = os.path.join(ROOT, "oracle", "_ref", "pbsim_ref_philox")
# model files are inputs (format contract); tests look for them here first and
# fall back to the read-only reference mount in this container
MODEL_GZ_DIR = os.path.join(ROOT, "tests", "golden", "models")
MODEL_CACHE = os.path.join(MODEL_GZ_DIR, "_unpacked")


def model_path(name):
    """FIC-HMM model files are INPUT DATA (format contract); the six
    used by the tests are committed gzip-compressed and unpacked on first use."""
    import gzip
    p = os.path.join(MODEL_CACHE, name)
    if not os.path.exists(p):
        src = os.path.join(MODEL_GZ_DIR, name + ".gz")
        if not os.path.exists(src):
            raise FileNotFoundError(name)
        os.makedirs(MODEL_CACHE, exist_ok=True)
        tmp = p + ".%d.tmp" % os.getpid()
        with gzip.open(src, "rb") as f, open(tmp, "wb") as g:
            g.write(f.read())
        os.replace(tmp, p)
    return p


INPUT_CACHE = os.path.join(GOLDEN, "inputs", "_unpacked")


def synth_bases_at(first, n, seed):
    """bytes first .. first + n - 1 of synth_bases(.., seed): a pure function of the position"""
    import numpy as np
    out = np.empty(n, dtype=np.uint8)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    step = 1 << 22
    with np.errstate(over="ignore"):
        for a in range(first, first + n, step):
            x = (np.arange(a, min(first + n, a + step), dtype=np.uint64) + np.uint64(seed * 0x632BE59BD9B4E019 & (2**64 - 1))) * np.uint64(0x9E3779B97F4A7C15)
            x ^= x >> np.uint64(30)
            x *= np.uint64(0xBF58476D1CE4E5B9)
            x ^= x >> np.uint64(27)
            x *= np.uint64(0x94D049BB133111EB)
            x ^= x >> np.uint64(31)
            out[a - first:a - first + len(x)] = lut[(x >> np.uint64(61)).astype(np.int64) & 3]
    return out


def synth_bases(n, seed):
    """n pseudo-random A/C/G/T bytes from plain 64-bit integer arithmetic (splitmix64 of the position), so the file is
    the same on every box and numpy version without being committed."""
    return synth_bases_at(0, n, seed)


def synth_bases_torch(n, seed, device="cuda"):
    """synth_bases on a torch device (the GPU box: 3 Gbp in a second instead of a minute): the same splitmix64 in int64
    arithmetic -- products wrap like uint64's, logical right shifts are arithmetic ones with the sign bits masked off.
    tests/test_fullsize_digests.py checks it against synth_bases on the CPU."""
    import torch

    def i64(v):                    # a uint64 constant as the int64 with the same bits
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    out = torch.empty(n, dtype=torch.uint8, device=device)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    step = 1 << 26
    add = i64(seed * 0x632BE59BD9B4E019)
    for a in range(0, n, step):
        x = (torch.arange(a, min(n, a + step), dtype=torch.int64, device=device) + add) * i64(0x9E3779B97F4A7C15)
        x = x ^ lsr(x, 30)
        x = x * i64(0xBF58476D1CE4E5B9)
        x = x ^ lsr(x, 27)
        x = x * i64(0x94D049BB133111EB)
        x = x ^ lsr(x, 31)
        out[a:a + x.numel()] = lut[lsr(x, 61) & 3]
    return out


def load_fullsize():
    """tests/golden/fullsize.json: CRC-32 + length of the streams the REFERENCE wrote for the BASELINE-size cases
    (tests/golden/make_fullsize.py), plus its stderr report"""
    with open(os.path.join(GOLDEN, "fullsize.json")) as f:
        return json.load(f)


def input_path(name):
    """tests/golden/inputs/<name>: committed as is, committed gzip-compressed (the reference's own sample files: data),
    or generated on first use -- `synth_<bases>_<seed>.fa`: one FASTA record of that many synth_bases, 80 per line."""
    import gzip
    import re
    p = os.path.join(GOLDEN, "inputs", name)
    if os.path.exists(p):
        return p
    q = os.path.join(INPUT_CACHE, name)
    if os.path.exists(q):
        return q
    os.makedirs(INPUT_CACHE, exist_ok=True)
    tmp = q + ".%d.tmp" % os.getpid()
    m = re.fullmatch(r"synth_(\d+)_(\d+)\.fa", name)
    if os.path.exists(p + ".gz"):
        with gzip.open(p + ".gz", "rb") as f, open(tmp, "wb") as g:
            g.write(f.read())
    elif m:
        import numpy as np
        n, seed = int(m.group(1)), int(m.group(2))
        b = synth_bases(n, seed)
        full = n // 80 * 80
        lines = np.empty((full // 80, 81), dtype=np.uint8)
        lines[:, :80] = b[:full].reshape(-1, 80)
        lines[:, 80] = 10
        with open(tmp, "wb") as g:
            g.write(b">synth_%d_%d\n" % (n, seed))
            g.write(lines.tobytes())
            if full < n:
                g.write(b[full:].tobytes() + b"\n")
    else:
        raise FileNotFoundError(name)
    os.replace(tmp, q)
    return q


def build_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    return ORACLE


def strip_report(err: str) -> str:
    keep = []
    for line in err.splitlines():
        if line.startswith((":::: System utilization", "CPU time(s)", "Elapsed time(s)", "oracle draws")):
            continue
        if line.split(" : ")[0] in ("prefix", "genome", "transcript", "errhmm", "qshmm", "file name", "template", "sample"):
            continue
        keep.append(line)
    return "\n".join(keep).rstrip("\n") + "\n"


def resolve(args):
    out = []
    for a in args:
        if a.startswith("MODEL:"):
            out.append(model_path(a[6:]))
        elif a.startswith("INPUT:"):
            out.append(input_path(a[6:]))
        else:
            out.append(a)
    return out


def collect(workdir, prefix="out"):
    res = {}
    for fn in sorted(os.listdir(workdir)):
        if fn.startswith("sample_profile_"):     # --sample-profile-id writes into the working directory (pbsim.cpp:1590)
            with open(os.path.join(workdir, fn), "rb") as f:
                res[".profile_" + fn.rsplit(".", 1)[1]] = f.read()
        if fn.startswith(prefix) and os.path.isfile(os.path.join(workdir, fn)):
            key = fn[len(prefix):]
            key = key.replace(".fq.gz", ".fq").replace(".maf.gz", ".maf").replace(".bam", ".sam")
            with open(os.path.join(workdir, fn), "rb") as f:
                res[key] = f.read()
    return res


def run_setup(exe_and_flags, case, workdir, env=None):
    """A case may name a command that must have run before it in the same directory (a stored sample profile)."""
    if case and case.get("setup"):
        subprocess.run(exe_and_flags[:1] + resolve(case["setup"]) + ["--prefix", os.path.join(workdir, "setup")] +
                       exe_and_flags[1:], capture_output=True, text=True, check=True, cwd=workdir, env=env)
        for fn in os.listdir(workdir):
            if fn.startswith("setup"):
                os.remove(os.path.join(workdir, fn))


def run_oracle(args, mode, workdir, extra=(), case=None):
    build_oracle()
    run_setup([ORACLE, "--rng", mode], case, workdir)
    p = subprocess.run([ORACLE] + resolve(args) + ["--prefix", os.path.join(workdir, "out"), "--rng", mode] + list(extra),
                       capture_output=True, text=True, cwd=workdir)
    if p.returncode != 0:
        raise RuntimeError(f"oracle failed ({p.returncode}): {p.stderr[-2000:]}")
    outs = collect(workdir)
    outs[".stderr"] = strip_report(p.stderr).encode()
    return outs


def make_stubs(d):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "gzip"), "w") as f:
        f.write("#!/bin/sh\nexec cat\n")
    with open(os.path.join(d, "samtools"), "w") as f:
        f.write('#!/bin/sh\nexec cat > "$4"\n')
    for n in ("gzip", "samtools"):
        os.chmod(os.path.join(d, n), 0o755)


def run_reference(args, mode, workdir, case=None):
    exe = REF_GLIBC if mode == "glibc" else REF_PHILOX
    args = resolve(args)
    seed = args[args.index("--seed") + 1]
    stubs = os.path.join(workdir, "stubs")
    make_stubs(stubs)
    env = dict(os.environ, PATH=stubs + ":" + os.environ["PATH"], PBSHIM_SEED=seed, PBSHIM_MODE="philox")
    run_setup([exe], case, workdir, env)
    p = subprocess.run([exe] + args + ["--prefix", os.path.join(workdir, "out")], env=env,
                       capture_output=True, text=True, check=True, cwd=workdir)
    outs = collect(workdir)
    outs[".stderr"] = strip_report(p.stderr).encode()
    return outs


def sha(b):
    return hashlib.sha256(b).hexdigest()


def load_manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)
