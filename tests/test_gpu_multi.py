"""Multi-rank invariance on real hardware: two processes (gloo rendezvous, both on
the one GPU of the test box) shard a record by read block through
pbsim3_amd.multi; the concatenated FASTQ/MAF must equal the single-context run
byte for byte (N-GPU output == 1-GPU output)."""
import os
import sys

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import harness
import product
from cases import CASES

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, args, batch, q):
    sys.path.insert(0, harness.ROOT)
    import pbsim3_amd as P
    from pbsim3_amd import multi
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p, a = product.params_from_args(args)
    out = []
    with P.Context(p, 0) as ctx:
        ctx.set_scratch_bytes(256 << 20)
        ctx.load_errhmm(a["--errhmm"])
        recs = product.read_fasta(a["--genome"])
        for i, r in enumerate(recs, 1):
            ctx.set_reference(r, i)
            kept = []

            def on_batch(info):
                rt, mt = ctx.batch_fetch(info)
                kept.append((info.first_read, rt, mt))

            reads, total = multi.simulate_record_sharded(ctx, multi.TorchComm(dist), batch, on_batch)
            out.append((reads, total, kept))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch", [7, 40])
def test_two_ranks_reproduce_single_context(batch):
    case = "wgs_errhmm-ont_quirk"
    args = harness.resolve(CASES[case]["args"])
    want, stats = product.run_wgs(args)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + batch) % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, args, batch, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    for rec in range(len(stats)):
        pieces = sorted(res[0][rec][2] + res[1][rec][2])
        fq = b"".join(x[1] for x in pieces)
        maf = b"".join(x[2] for x in pieces)
        assert res[0][rec][0] == res[1][rec][0] == stats[rec].res_num
        assert fq == want["_%04d.fq" % (rec + 1)]
        assert maf == want["_%04d.maf" % (rec + 1)]


def test_unit_range_shards_concatenate_to_the_whole(tmp_path):
    """pbsim_simulate_units_range: any partition of 1 .. pbsim_unit_reads() into contiguous blocks gives, concatenated,
    the bytes of pbsim_simulate_trans; the shard statistics add up; bad ranges are refused"""
    import pbsim3_amd as P
    from pbsim3_amd import args as A
    argv = harness.resolve(CASES["trans_errhmm_sequel"]["args"])
    p, a = A.parse(argv)
    with P.Context(p, 0) as ctx:
        ctx.load_errhmm(a["--errhmm"])
        n_units, total = ctx.load_transcript_file(a["--transcript"])
        R = ctx.unit_reads()
        assert n_units > 0 and R == total
        whole = ctx.simulate_trans()
        st = ctx.stats()
        for cuts in ([0, R], [0, 1, R], [0, R // 3, R // 3, 2 * R // 3, R - 1, R]):
            rt = mt = b""
            n = bases = 0
            for lo, hi in zip(cuts, cuts[1:]):
                r, m = ctx.simulate_units_range(lo + 1, hi - lo)
                rt += r
                mt += m
                if hi > lo:
                    s = ctx.stats()
                    n += s.res_num
                    bases += s.res_len_total
            assert (rt, mt) == whole
            assert (n, bases) == (st.res_num, st.res_len_total)
        for first, cnt in ((0, 1), (1, R + 1), (R + 1, 1), (1, -1)):
            with pytest.raises(P.PbsimError):
                ctx.simulate_units_range(first, cnt)
