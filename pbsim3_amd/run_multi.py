"""Multi-GPU front-end (wgs: quota loop sharded by read block; trans / templ: read blocks of the unit set): one process per GPU under torchrun,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m pbsim3_amd.run_multi --strategy wgs --method errhmm --errhmm M --genome G.fa --prefix out [...]

Rank 0 reads the FASTA and broadcasts each record (C1); every rank walks its read blocks
(pbsim3_amd.multi: C3 all-gathers place the quota); text is written per rank and stitched by rank 0
in read order; counters are reduced at the end of each record (C2).  The FASTQ/MAF bytes equal the
single-GPU `pbsim --no-gzip` output; the report is the reference's stderr block.
Extra options: --batch-reads N (reads per rank per round), --scratch-mb M, --backend nccl|gloo,
--one-gpu (all ranks on device 0: plumbing check on a single-GPU box), --gzip (every rank compresses
its text on its GPU; gzip members are self-contained, so stitching them in read order gives valid
<prefix>_NNNN.fq.gz / .maf.gz, and .bam for --pass-num > 1, whose decompressed bytes are the same).
"""
import os
import sys

import numpy as np


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    flags = {"--one-gpu": False, "--gzip": False}
    for f in list(flags):
        if f in argv:
            argv.remove(f)
            flags[f] = True
    import torch
    import torch.distributed as dist

    import pbsim3_amd as P
    from pbsim3_amd import args as A
    from pbsim3_amd import multi

    p, a = A.parse(argv)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if flags["--one-gpu"] else int(os.environ.get("LOCAL_RANK", "0"))
    backend = a.get("--backend", "nccl")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if (world == 1 or backend == "nccl") else torch.device("cpu")
    comm = multi.TorchComm(dist, cdev) if world > 1 else multi.SoloComm()
    prefix = a.get("--prefix", "sd")

    def bcast_i64(vals):
        t = torch.tensor(vals, dtype=torch.int64, device=cdev)
        if world > 1:
            dist.broadcast(t, src=0)
        return [int(x) for x in t.tolist()]

    if p.strategy != P.STRATEGY_WGS:
        run_units(p, a, flags, P, torch, dist, comm, rank, world, local, cdev, prefix)
        if world > 1:
            dist.destroy_process_group()
        return

    recs = []
    if rank == 0:
        recs, _ = A.read_fasta(a["--genome"])
    n_rec = bcast_i64([len(recs)])[0]
    lens = bcast_i64([len(r) for r in recs] if rank == 0 else [0] * n_rec)

    ctx = P.Context(p, local)
    if "--scratch-mb" in a:
        ctx.set_scratch_bytes(int(a["--scratch-mb"]) << 20)
    (ctx.load_errhmm if p.method == P.METHOD_ERR else ctx.load_qshmm)(a["--errhmm" if p.method == P.METHOD_ERR else "--qshmm"])

    def record_tensor(i):
        if rank == 0:
            t = torch.frombuffer(bytearray(recs[i]), dtype=torch.uint8)
        else:
            t = torch.empty(lens[i], dtype=torch.uint8)
        t = t.to(cdev)
        if world > 1:
            dist.broadcast(t, src=0)                      # C1
        return t.to(dev)

    if p.hp_del_bias != 1:                                 # census over ALL records first (pbsim.cpp:677-696)
        for i in range(n_rec):
            ctx.add_hp_census(bytes(record_tensor(i).cpu().numpy()))
        ctx.finish_hp_census()

    for i in range(n_rec):
        g = record_tensor(i)
        ctx.set_reference_device(g.data_ptr(), lens[i], i + 1)
        del g
        ctx.reset_stats()
        batch = int(a.get("--batch-reads", 0)) or max(1, ctx.batch_capacity())
        part = "%s_%04d.rank%d" % (prefix, i + 1, rank)
        gz = flags["--gzip"]
        ext = "fq" if p.pass_num == 1 else "sam"
        if gz and p.pass_num > 1:
            ctx.set_bam_output(True)
        out_ext, maf_ext = ((ext + ".gz") if p.pass_num == 1 else "bam", "maf.gz") if gz else (ext, "maf")
        index = []                                         # (first_read, bytes of read text, bytes of maf text)
        with open(part + "." + ext, "wb") as fr, open(part + ".maf", "wb") as fm:
            def on_batch(info):
                rt, mt = ctx.batch_fetch_deflated(info) if gz else ctx.batch_fetch(info)
                fr.write(rt)
                fm.write(mt)
                ctx.batch_account()
                index.append((info.first_read, len(rt), len(mt)))

            reads, total = multi.simulate_record_sharded(ctx, comm, batch, on_batch)
        # ---- stitch in read order on rank 0
        mine = [x for t3 in index for x in t3]
        n_max = max(v[0] for v in comm.all_gather_i64([len(index)]))
        table = comm.all_gather_i64(mine + [0] * (3 * n_max - len(mine)))
        if world > 1:
            dist.barrier()
        if rank == 0:
            pieces = []
            for r in range(world):
                off_r = off_m = 0
                for k in range(0, 3 * n_max, 3):
                    first, nr, nm = table[r][k:k + 3]
                    if nr == 0 and nm == 0:
                        continue
                    pieces.append((first, r, off_r, nr, off_m, nm))
                    off_r += nr
                    off_m += nm
            pieces.sort()
            with open("%s_%04d.%s" % (prefix, i + 1, out_ext), "wb") as fr, \
                    open("%s_%04d.%s" % (prefix, i + 1, maf_ext), "wb") as fm:
                if p.pass_num > 1:
                    fr.write(ctx.deflate_buffer(ctx.bam_header()) if gz else ctx.sam_header())
                files = {r: (open("%s_%04d.rank%d.%s" % (prefix, i + 1, r, ext), "rb"),
                             open("%s_%04d.rank%d.maf" % (prefix, i + 1, r), "rb")) for r in range(world)}
                for first, r, off_r, nr, off_m, nm in pieces:
                    files[r][0].seek(off_r)
                    fr.write(files[r][0].read(nr))
                    files[r][1].seek(off_m)
                    fm.write(files[r][1].read(nm))
                for r in files:
                    files[r][0].close()
                    files[r][1].close()
                if gz:    # BAM: the BGZF end-of-file marker; .gz: an empty member keeps an empty output a valid gzip file
                    if p.pass_num > 1 or fr.tell() == 0:
                        fr.write(P.BGZF_EOF)
                    if fm.tell() == 0:
                        fm.write(P.BGZF_EOF)
        if world > 1:
            dist.barrier()
        os.remove(part + "." + ext)
        os.remove(part + ".maf")
        # ---- C2: counters of the record
        st = ctx.stats()
        ints = torch.tensor([st.res_num, st.res_len_total, st.res_sub_num, st.res_ins_num, st.res_del_num],
                            dtype=torch.int64, device=cdev)
        acc = torch.tensor([st.res_accuracy_mean * st.res_pass_num if st.res_num else 0.0], dtype=torch.float64, device=cdev)
        mn = torch.tensor([st.res_len_min if st.res_num else 2**62], dtype=torch.int64, device=cdev)
        mx = torch.tensor([st.res_len_max], dtype=torch.int64, device=cdev)
        if world > 1:
            dist.all_reduce(ints)
            dist.all_reduce(acc)
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        if rank == 0:
            n, tot, ns, ni, nd = (int(x) for x in ints.tolist())
            passes = n * p.pass_num
            sys.stderr.write(":::: Simulation stats (ref.%d) ::::\n\n" % (i + 1))
            sys.stderr.write("read num. : %d\n" % n)
            sys.stderr.write("depth : %f\n" % (tot / lens[i] / p.pass_num))
            sys.stderr.write("read length mean : %f\n" % (tot / passes))
            sys.stderr.write("read length min : %d\nread length max : %d\n" % (int(mn.item()), int(mx.item())))
            sys.stderr.write("read accuracy mean : %f\n" % (float(acc.item()) / passes))
            sys.stderr.write("substitution rate. : %f\ninsertion rate. : %f\ndeletion rate. : %f\n\n" % (ns / tot, ni / tot, nd / tot))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def run_units(p, a, flags, P, torch, dist, comm, rank, world, local, cdev, prefix):
    """trans / templ strategies: no quota, every read of the unit set is final (simulate_by_*_trans pbsim.cpp:4428-4770,
    simulate_by_*_templ :4807-5392), so rank r simply takes the r-th contiguous block of the global read numbering.
    Every rank parses the unit file itself (one node, one file system); text is stitched in rank order."""
    templ = p.strategy == P.STRATEGY_TEMPL
    ctx = P.Context(p, local)
    if "--scratch-mb" in a:
        ctx.set_scratch_bytes(int(a["--scratch-mb"]) << 20)
    (ctx.load_errhmm if p.method == P.METHOD_ERR else ctx.load_qshmm)(a["--errhmm" if p.method == P.METHOD_ERR else "--qshmm"])
    n_units, total = ctx.load_template_file(a["--template"]) if templ else ctx.load_transcript_file(a["--transcript"])
    if rank == 0:
        if templ:
            sys.stderr.write(":::: Template stats ::::\n\nfile name : %s\ntemplate num. : %d\ntemplate total length : %d\n\n"
                             % (a["--template"], n_units, total))
        else:
            sys.stderr.write(":::: transcript stats ::::\n\nfile name : %s\ntranscript num : %d\ntotal expression value : %d\n\n"
                             % (a["--transcript"], n_units, total))
    R = ctx.unit_reads()
    per = (R + world - 1) // world
    first = 1 + rank * per
    n = max(0, min(per, R - first + 1))
    gz = flags["--gzip"]
    ext = "fq" if p.pass_num == 1 else "sam"
    if gz:
        ctx.set_deflate(3)
        if p.pass_num > 1:
            ctx.set_bam_output(True)
    rt, mt = ctx.simulate_units_range(first, n) if n > 0 else (b"", b"")
    part = "%s.rank%d" % (prefix, rank)
    with open(part + "." + ext, "wb") as f:
        f.write(rt)
    with open(part + ".maf", "wb") as f:
        f.write(mt)
    if world > 1:
        dist.barrier()
    if rank == 0:
        out_ext, maf_ext = ((ext + ".gz") if p.pass_num == 1 else "bam", "maf.gz") if gz else (ext, "maf")
        with open(prefix + "." + out_ext, "wb") as fr, open(prefix + "." + maf_ext, "wb") as fm:
            if p.pass_num > 1:
                fr.write(ctx.deflate_buffer(ctx.bam_header()) if gz else ctx.sam_header())
            for r in range(world):
                with open("%s.rank%d.%s" % (prefix, r, ext), "rb") as f:
                    fr.write(f.read())
                with open("%s.rank%d.maf" % (prefix, r), "rb") as f:
                    fm.write(f.read())
            if gz:
                if p.pass_num > 1 or fr.tell() == 0:
                    fr.write(P.BGZF_EOF)
                if fm.tell() == 0:
                    fm.write(P.BGZF_EOF)
    if world > 1:
        dist.barrier()
    os.remove(part + "." + ext)
    os.remove(part + ".maf")
    # ---- C2: counters
    st = ctx.stats() if n > 0 else None
    ints = torch.tensor([st.res_num, st.res_len_total, st.res_sub_num, st.res_ins_num, st.res_del_num] if st else [0] * 5,
                        dtype=torch.int64, device=cdev)
    acc = torch.tensor([st.res_accuracy_mean * st.res_pass_num if st and st.res_num else 0.0], dtype=torch.float64, device=cdev)
    mn = torch.tensor([st.res_len_min if st and st.res_num else 2**62], dtype=torch.int64, device=cdev)
    mx = torch.tensor([st.res_len_max if st else 0], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(ints)
        dist.all_reduce(acc)
        dist.all_reduce(mn, op=dist.ReduceOp.MIN)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    if rank == 0:
        nr, tot, ns, ni, nd = (int(x) for x in ints.tolist())
        passes = max(1, nr * p.pass_num)
        sys.stderr.write(":::: Simulation stats ::::\n\nread num. : %d\n" % nr)
        sys.stderr.write("read length mean : %f\n" % (tot / passes))
        sys.stderr.write("read length min : %d\nread length max : %d\n" % (int(mn.item()) if nr else 0, int(mx.item())))
        sys.stderr.write("read accuracy mean : %f\n" % (float(acc.item()) / passes))
        sys.stderr.write("substitution rate. : %f\ninsertion rate. : %f\ndeletion rate. : %f\n\n"
                         % ((ns / tot, ni / tot, nd / tot) if tot else (0.0, 0.0, 0.0)))
    ctx.close()


if __name__ == "__main__":
    main()
