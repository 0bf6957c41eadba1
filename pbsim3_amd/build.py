"""Builds the HIP product in-tree: pbsim3_amd/lib/libpbsim3_amd.so (C ABI of
include/pbsim3_amd.h) and pbsim3_amd/bin/pbsim (the CLI shim), for gfx950 only.
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
BIN_DIR = os.path.join(HERE, "bin")
OBJ_DIR = os.path.join(LIB_DIR, "obj")
LIB = os.path.join(LIB_DIR, "libpbsim3_amd.so")
CLI = os.path.join(BIN_DIR, "pbsim")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

HIP_SOURCES = ["kernels.hip", "deflate.hip", "engine.cpp", "deflate_host.cpp", "units.cpp", "sample.cpp", "job.cpp", "rccl_capi.cpp"]
CXX_SOURCES = ["host_tables.cpp", "unit_io.cpp", "stats.cpp", "cli.cpp", "gzout.cpp", "numa_bind.cpp"]
CLI_SOURCES = ["main.cpp"]
COMMON = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"] + os.environ.get("PBSIM_EXTRA_CFLAGS", "").split()


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    d.append(os.path.join(os.path.dirname(HERE), "include", "pbsim3_amd.h"))
    return d


def _assert_gfx950(path):
    """The fat binary must carry a gfx950 code object (and nothing else)."""
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", path],
                         capture_output=True, text=True, cwd=os.path.dirname(path)).stdout
    for f in os.listdir(os.path.dirname(path)):   # llvm-objdump drops the extracted bundles next to the input
        if f.startswith(os.path.basename(path) + "."):
            os.remove(os.path.join(os.path.dirname(path), f))
    if f"{ARCH}" not in out:
        raise RuntimeError(f"{path}: no {ARCH} code object in the fat binary:\n{out[-2000:]}")


def build(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(BIN_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    deps = _deps()
    headers = [d for d in deps if d.endswith(".h")]
    if force or _newer(LIB, deps):
        # one object per source (kernels.hip alone takes most of a minute), relinked when any of them changed.
        # NB: no `-x` flags -- hipcc drops --offload-arch when it sees one and silently builds for its default target
        objs = []
        for s in HIP_SOURCES + CXX_SOURCES:
            src = os.path.join(CSRC, s)
            if not os.path.exists(src):
                continue
            obj = os.path.join(OBJ_DIR, s + ".o")
            objs.append(obj)
            if force or _newer(obj, [src] + headers):
                cmd = [HIPCC, f"--offload-arch={ARCH}", "-c", "-o", obj + ".tmp"] + COMMON + [src]
                if verbose:
                    print(" ".join(cmd))
                subprocess.run(cmd, check=True)
                os.replace(obj + ".tmp", obj)
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-o", LIB + ".tmp"] + objs + ["-lz", "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        _assert_gfx950(LIB + ".tmp")
        os.replace(LIB + ".tmp", LIB)
    cli_srcs = [os.path.join(CSRC, s) for s in CLI_SOURCES]
    if all(os.path.exists(s) for s in cli_srcs) and (force or _newer(CLI, deps + [LIB])):
        cmd = [HIPCC, "-o", CLI + ".tmp"] + COMMON + cli_srcs + \
              ["-L" + LIB_DIR, "-lpbsim3_amd", "-Wl,-rpath,$ORIGIN/../lib", "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
        os.replace(CLI + ".tmp", CLI)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
