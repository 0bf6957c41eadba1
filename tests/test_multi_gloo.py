"""The N>1 sharding protocol (pbsim3_amd/multi.py) on CPU: two gloo ranks drive a
deterministic stand-in engine that implements the reference's quota rule
(pbsim.cpp:3792-3800, 3989-3991) on hash-derived read lengths.  The sharded run
must keep exactly the reads -- and the same truncated tail -- as a single rank."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pbsim3_amd import multi  # noqa: E402


class Info:
    pass


class FakeEngine:
    """raw length and output length of read r are pure functions of r (like the keyed stream)."""

    def __init__(self, quota, len_min=100):
        self.quota, self.len_min = quota, len_min

    @staticmethod
    def raw(r):
        return 200 + (r * 2654435761 % 4001)

    @staticmethod
    def out(r, L):
        return max(1, L + ((r * 40503) % 61) - 30)

    def unit_quota(self):
        return self.quota

    def batch_walk(self, first, n, trunc):
        self.first, self.n, self.trunc = first, n, trunc
        self.L = []
        for r in range(first, first + n):
            L = self.raw(r)
            if trunc >= 0 and L > trunc:
                L = max(trunc, self.len_min)
            self.L.append(L)
        self.o = [self.out(r, L) for r, L in zip(range(first, first + n), self.L)]
        return sum(self.o)

    def batch_finalize(self, before):
        i = Info()
        t, k = before, 0
        if self.trunc >= 0:
            k, t = 1, before + self.o[0]
            i.quota_reached = int(t >= self.quota)
            i.need_truncated_read = int(not i.quota_reached)
        else:
            while k < self.n and t < self.quota and t + self.raw(self.first + k) <= self.quota:
                t += self.o[k]
                k += 1
            i.quota_reached = int(k < self.n or t >= self.quota)
            i.need_truncated_read = int(k < self.n and t < self.quota)
        i.first_read, i.n_reads, i.n_final, i.len_total_after = self.first, self.n, k, t
        i.lens = self.o[:k]
        return i


def _worker(rank, world, port, quota, batch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kept = []
    reads, total = multi.simulate_record_sharded(FakeEngine(quota), multi.TorchComm(dist), batch,
                                                 on_batch=lambda i: kept.append((i.first_read, i.lens)))
    q.put((rank, reads, total, kept))
    dist.barrier()
    dist.destroy_process_group()


def _serial(quota, batch):
    kept = []
    reads, total = multi.simulate_record_sharded(FakeEngine(quota), multi.SoloComm(), batch,
                                                 on_batch=lambda i: kept.append((i.first_read, i.lens)))
    return reads, total, kept


def _flatten(kept):
    out = {}
    for first, lens in kept:
        for k, v in enumerate(lens):
            assert first + k not in out
            out[first + k] = v
    return out


@pytest.mark.parametrize("quota,batch", [(250_000, 16), (250_000, 37), (1_000, 8), (90_000, 64)])
def test_two_ranks_equal_one(quota, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + quota + batch) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, quota, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    s_reads, s_total, s_kept = _serial(quota, batch)
    merged = {}
    for rank, reads, total, kept in res:
        assert (reads, total) == (s_reads, s_total)
        merged.update(_flatten(kept))
    want = _flatten(s_kept)
    assert merged == want
    assert sorted(want) == list(range(1, s_reads + 1))
    assert sum(want.values()) == s_total >= quota
