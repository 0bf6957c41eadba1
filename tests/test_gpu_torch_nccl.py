"""The bench's and run_multi's communicator on its real backend: pbsim3_amd.torch_comm over torch.distributed `nccl` (= RCCL on
ROCm), as a group of ONE rank -- all this box can offer (RCCL takes one rank per GPU).  The pbsim_comm callbacks are called
through their C function pointers exactly as job.cpp calls them: all-gather (rank-major), all-reduce SUM / MIN / MAX on int64,
large buffers (the statistics merge moves 100 001-bin histograms and 8 bytes per read), and the record broadcast bench.py does
with dist.broadcast.  What a group of one cannot show -- ranks waiting for each other -- is covered over gloo
(tests/test_gpu_run_multi.py) and by the watchdog / abort tests."""
import os
import subprocess
import sys

import pytest

import harness

pytestmark = pytest.mark.gpu

CODE = r'''
import os, sys, ctypes as C
import numpy as np
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import pbsim3_amd as P
P.bind_host_to_device(0)
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
comm = P.torch_comm(dist, torch.device("cuda", 0))
assert (comm.rank, comm.world) == (0, 1)
I64 = C.POINTER(C.c_int64)
# C3: all-gather
send = np.array([5, -7, 1 << 40], dtype=np.int64)
recv = np.zeros(3, dtype=np.int64)
assert comm.all_gather_i64(None, send.ctypes.data_as(I64), 3, recv.ctypes.data_as(I64)) == 1
assert recv.tolist() == send.tolist()
# C2: all-reduce, every op, a histogram-sized buffer
for op in (P.OP_SUM, P.OP_MIN, P.OP_MAX):
    buf = (np.arange(100001, dtype=np.int64) * 3 - 50000)
    want = buf.copy()
    assert comm.all_reduce_i64(None, buf.ctypes.data_as(I64), len(buf), op) == 1
    assert (buf == want).all()
big = np.arange(2_000_000, dtype=np.int64)
out = np.zeros_like(big)
assert comm.all_gather_i64(None, big.ctypes.data_as(I64), len(big), out.ctypes.data_as(I64)) == 1 and (out == big).all()
# C1 as bench.py does it
t = torch.arange(0, 1 << 24, dtype=torch.uint8, device="cuda")
dist.broadcast(t, src=0)
torch.cuda.synchronize()
# the statistics merge through the same communicator (a group of one returns at once, the call path is the product's)
p = P.default_params(strategy=P.STRATEGY_WGS, method=P.METHOD_ERR)
ctx = P.Context(p, -1)
ctx.stats_keep_values(True)
ctx.stats_add_tasks(0, [100, 200], [1, 2], [3, 4], [5, 6])
ctx.stats_merge(comm)
assert ctx.stats().res_num == 2
# the library's own communicator beside torch's, its id through torch's store (what bench.py --gpus N and run_multi do)
nat = P.RcclComm.from_torch(dist, 0)
assert nat.info()["ranks_seen"] == 1 and (nat.comm.rank, nat.comm.world) == (0, 1)
out3 = np.zeros(3, dtype=np.int64)
assert nat.comm.all_gather_i64(nat.comm.user, send.ctypes.data_as(I64), 3, out3.ctypes.data_as(I64)) == 1 and out3.tolist() == send.tolist()
hist = np.arange(100001, dtype=np.int64)
assert nat.comm.all_reduce_i64(nat.comm.user, hist.ctypes.data_as(I64), len(hist), P.OP_SUM) == 1 and hist[-1] == 100000
lat = P.comm_latency(nat.ref, 8, 200, 20)
tlat = P.comm_latency(C.byref(comm), 8, 200, 20)
print("latency us: native", lat["all_gather_us"], "torch callbacks", tlat["all_gather_us"])
nat.close()
dist.barrier()
dist.destroy_process_group()
print("NCCL-COMM-OK")
'''


def test_torch_comm_over_nccl_group_of_one():
    port = str(33500 + os.getpid() % 1000)
    env = dict(os.environ, PYTHONPATH=harness.ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", CODE, port], capture_output=True, text=True, cwd=harness.ROOT, env=env, timeout=280)
    assert p.returncode == 0 and "NCCL-COMM-OK" in p.stdout, (p.stdout + p.stderr)[-3000:]
