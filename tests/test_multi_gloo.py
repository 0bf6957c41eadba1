"""The N > 1 path on CPU (gloo, world size 2 and 3): the job's communicator (pbsim3_amd.torch_comm -> pbsim_comm
callbacks) and the statistics merge of include/pbsim3_amd.h (pbsim_stats_keep_values / pbsim_stats_merge), both
device-free.  Every rank accounts its blocks of a synthetic unit's tasks; after the merge EVERY rank must report exactly
what one context reports for all tasks in read order -- counters, min/max, both SDs (the two histograms) and the
order-dependent accuracy sum of pbsim.cpp:4003 bit for bit."""
import ctypes as C
import os
import struct
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def tasks(n, pass_num, seed):
    rng = np.random.default_rng(seed)
    n_tasks = n * pass_num
    out_len = rng.integers(50, 30000, n_tasks).astype(np.int32)
    nsub = (out_len * rng.uniform(0.0, 0.02, n_tasks)).astype(np.int32)
    nins = (out_len * rng.uniform(0.0, 0.08, n_tasks)).astype(np.int32)
    ndel = (out_len * rng.uniform(0.0, 0.06, n_tasks)).astype(np.int32)
    qsum = out_len * rng.uniform(0.01, 0.2, n_tasks)
    return out_len, nsub, nins, ndel, qsum


def blocks_of(n_reads, world, block):
    """round-robin blocks of `block` reads like the job's rounds: block k goes to rank k % world"""
    out = [[] for _ in range(world)]
    for k, first in enumerate(range(0, n_reads, block)):
        out[k % world].append((first, min(block, n_reads - first)))
    return out


def stats_tuple(s):
    return tuple(getattr(s, f) if not isinstance(getattr(s, f), float) else struct.pack("<d", getattr(s, f))
                 for f, _ in type(s)._fields_)


def single(method, pass_num, n_reads, seed):
    import pbsim3_amd as P
    ctx = P.Context(P.default_params(method=method, pass_num=pass_num), -1)
    t = tasks(n_reads, pass_num, seed)
    ctx.stats_add_tasks(0, *t[:4], qsum=t[4] if method == P.METHOD_QS else None)
    s = ctx.stats()
    ctx.close()
    return stats_tuple(s)


def _worker(rank, world, port, method, pass_num, n_reads, block, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pbsim3_amd as P
    comm = P.torch_comm(dist, "cpu")
    # the communicator itself: gather is rank-major, reduce is element-wise
    lib = P.load()
    send = (C.c_int64 * 3)(rank, 10 * rank, -rank)
    recv = (C.c_int64 * (3 * world))()
    assert comm.all_gather_i64(None, send, 3, recv) == 1
    assert list(recv) == [v for r in range(world) for v in (r, 10 * r, -r)]
    for op, want in ((P.OP_SUM, sum(range(world))), (P.OP_MIN, 0), (P.OP_MAX, world - 1)):
        buf = (C.c_int64 * 2)(rank, rank)
        assert comm.all_reduce_i64(None, buf, 2, op) == 1
        assert list(buf) == [want, want]
    # the merge
    ctx = P.Context(P.default_params(method=method, pass_num=pass_num), -1)
    ctx.stats_keep_values(True)
    t = tasks(n_reads, pass_num, seed)
    for first, n in blocks_of(n_reads, world, block)[rank]:
        sl = slice(first * pass_num, (first + n) * pass_num)
        ctx.stats_add_tasks(first * pass_num, *[x[sl] for x in t[:4]], qsum=t[4][sl] if method == P.METHOD_QS else None)
    ctx.stats_merge(comm)
    q.put((rank, stats_tuple(ctx.stats())))
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    del lib


@pytest.mark.parametrize("world,method,pass_num,n_reads,block", [(2, 2, 1, 5000, 700), (2, 1, 3, 1200, 97), (3, 2, 1, 4001, 512)])
def test_merged_statistics_equal_one_context(world, method, pass_num, n_reads, block):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + n_reads + block) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, method, pass_num, n_reads, block, 11, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = single(method, pass_num, n_reads, 11)
    for r in range(world):
        assert res[r] == want, (r, res[r], want)


def test_sum_order_matters_in_this_data():
    """the check above is only meaningful if a per-rank partial sum would have given different bits"""
    t = tasks(5000, 1, 11)
    v = 1.0 - (t[1].astype(np.int64) + t[2] + t[3]) / t[0]
    seq = 0.0
    for x in v:
        seq += x
    parts = [0.0, 0.0]
    for rank in range(2):
        for first, n in blocks_of(5000, 2, 700)[rank]:
            for x in v[first:first + n]:
                parts[rank] += x
    assert struct.pack("<d", seq) != struct.pack("<d", parts[0] + parts[1])


def _id_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pbsim3_amd as P
    got = []
    for n in (P.RCCL_ID_BYTES, 0, P.RCCL_ID_BYTES):     # a good id, "rank 0 could not make one", and a second communicator's id
        ident = bytes((7 * i + n + 1) & 255 for i in range(n)) if rank == 0 else None
        got.append(P.store_exchange(dist)(ident))
    q.put((rank, got))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_id_travels_through_the_store():
    """the side channel of pbsim3_amd.RcclComm.from_torch (bench.py --gpus N, run_multi): rank 0's 128 bytes reach every rank
    through torch.distributed's store, call after call under fresh keys; an empty id (rank 0 failed) arrives as empty"""
    world, port = 3, 29700 + os.getpid() % 200
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_id_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    import pbsim3_amd as P
    want = [bytes((7 * i + n + 1) & 255 for i in range(n)) for n in (P.RCCL_ID_BYTES, 0, P.RCCL_ID_BYTES)]
    assert all(res[r] == want for r in range(world))
