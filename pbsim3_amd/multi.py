"""Read-block sharding of one WGS record over the GPUs of a node.

Reads are independent given the keyed Philox stream, so
the path shards with no data-path collective.  Per round every rank walks one
contiguous block of `batch_reads` reads; the only exchange is

  C3  all_gather of one int64 per rank: the block's pass-0 bases, from which each
      rank derives `len_total` in front of its block (the quota prefix), and
  C3' all_gather of (n_final, need_truncated, len_total_after) to agree on the cut.

The serial tail behind a truncated read (pbsim.cpp:3795-3800) runs on rank 0.
Concatenating the delivered text in (round, rank) order, then the tail, gives
byte for byte what one GPU produces.  The statistics counters are summed with
one all_reduce at the end (C2); the order-dependent double `accuracy_total`
(pbsim.cpp:4003) is summed per rank and then across ranks, so its last bits may
differ from the single-GPU sum.

`engine` is a pbsim3_amd.Context (or anything with batch_walk / batch_finalize /
unit_quota); `comm` needs rank, world and all_gather_i64(list[int]) -> list[list[int]].
"""


class TorchComm:
    """torch.distributed adapter (backend nccl = RCCL on GPUs, gloo on CPU)."""

    def __init__(self, dist, device=None):
        import torch
        self.dist, self.torch, self.device = dist, torch, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def all_gather_i64(self, values):
        t = self.torch.tensor(list(values), dtype=self.torch.int64, device=self.device)
        out = [self.torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [[int(x) for x in o.tolist()] for o in out]


class SoloComm:
    rank, world = 0, 1

    def all_gather_i64(self, values):
        return [list(values)]


def simulate_record_sharded(engine, comm, batch_reads, on_batch=None):
    """Quota loop of the current record, sharded by read block.

    on_batch(info) is called on the rank that owns a finalized batch with
    info.n_final > 0, in the order the reads must be concatenated per rank.
    Returns (reads, len_total) of the whole record (identical on every rank)."""
    quota = engine.unit_quota()
    rank, world = comm.rank, comm.world
    len_total, next_read = 0, 1
    while len_total < quota:
        first = next_read + rank * batch_reads
        pass0 = engine.batch_walk(first, batch_reads, -1)
        allp = [v[0] for v in comm.all_gather_i64([pass0])]
        before = len_total + sum(allp[:rank])
        # a block that starts at or beyond the quota is void: finalize reports n_final == 0
        info = engine.batch_finalize(before)
        res = comm.all_gather_i64([info.n_final, info.need_truncated_read, info.len_total_after])
        cut = next((r for r in range(world) if res[r][0] < batch_reads), None)
        mine_valid = cut is None or rank <= cut
        if mine_valid and info.n_final > 0 and on_batch:
            on_batch(info)
        if cut is None:
            next_read += world * batch_reads
            len_total = res[world - 1][2]
            continue
        next_read += cut * batch_reads + res[cut][0]
        len_total = res[cut][2]
        need_trunc = bool(res[cut][1])
        # serial tail on rank 0, one truncated read at a time; everyone follows its progress
        while need_trunc and len_total < quota:
            if rank == 0:
                engine.batch_walk(next_read, 1, quota - len_total)
                info = engine.batch_finalize(len_total)
                if on_batch:
                    on_batch(info)
                state = [info.n_final, info.len_total_after]
            else:
                state = [0, 0]
            state = comm.all_gather_i64(state)[0]
            next_read += state[0]
            len_total = state[1]
        break
    return next_read - 1, len_total
