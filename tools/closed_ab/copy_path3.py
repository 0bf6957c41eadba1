"""which host process keeps its pinned D2H copies on SDMA under rocprofv3 --kernel-trace: a delivered job (50 Mbp x depth 20) from a
python process with (a) the system HIP runtime only (PBSIM_TORCH_COMPAT=0), (b) torch's bundled runtime mapped first but torch
not imported, (c) torch imported.  usage: copy_path3.py nocompat|compat|torch"""
import ctypes as C, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
mode = sys.argv[1]
if mode == "torch":
    import torch  # noqa: F401
if mode == "nocompat":
    os.environ["PBSIM_TORCH_COMPAT"] = "0"
import numpy as np
import pbsim3_amd as P
import harness
rng = np.random.default_rng(1)
genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, 50_000_000)].tobytes()
p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR, seed=1, depth=20.0)
ctx = P.Context(p, 0)
ctx.load_errhmm(harness.model_path("ERRHMM-ONT.model"))
ctx.set_deflate(7)
ctx.job_add_record(genome)
n = [0]
cb = P.REC_TEXT_CB(lambda u, r, t, k, o: (n.__setitem__(0, n[0] + k), 1)[1])
sink = P.RecordSink(None, cb, cb, P.REC_DONE_CB())
for i in range(3):
    n[0] = 0
    t0 = time.perf_counter()
    P._check(ctx.lib.pbsim_job_run(ctx.h, None, C.byref(sink)))
    dt = time.perf_counter() - t0
print("%s: job %.1f ms, %.1f MB delivered = %.1f GB/s" % (mode, dt * 1e3, n[0] / 1e6, n[0] / dt / 1e9))
ctx.close()
