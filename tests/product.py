"""Drives the HIP product through its C ABI from a pbsim command line (test
helper; mirrors what pbsim3_amd/csrc/cli.cpp does in C++)."""
import os

import pbsim3_amd as P


def scratch_mb_for(case):
    """scratch pool (MB per slot) for a golden case through the job pipeline: a few MB force many rounds per record; a pool
    must still hold one wave of the case's longest reads (64 lanes x rows x (2 L + 64) columns)"""
    if "ultralong" in case:
        return 1024
    if "config0" in case:
        return 64
    return 32 if "default" in case else 4


def read_fasta(path):
    from pbsim3_amd import args
    return args.read_fasta(path)[0]


def params_from_args(argv):
    from pbsim3_amd import args
    return args.parse(argv)


def run_wgs(args, device=0, scratch_mb=None, deflate=False):
    """Returns {'_0001.fq': bytes, '_0001.maf': bytes, ...} plus per-record Stats."""
    p, a = params_from_args(args)
    outs, stats = {}, []
    with P.Context(p, device) as ctx:
        if scratch_mb:
            ctx.set_scratch_bytes(scratch_mb << 20)
        if deflate:     # the sinks then receive gzip members compressed on the GPU
            ctx.set_deflate(True)
        if p.method == P.METHOD_ERR:
            ctx.load_errhmm(a["--errhmm"])
        else:
            ctx.load_qshmm(a["--qshmm"])
        recs = read_fasta(a["--genome"])
        if p.hp_del_bias != 1:
            for r in recs:
                ctx.add_hp_census(r)
            ctx.finish_hp_census()
        for i, r in enumerate(recs, 1):
            ctx.set_reference(r, i)
            rt, mt = ctx.simulate_wgs()
            if p.pass_num > 1:
                h = ctx.sam_header()        # main() writes it when opening the pipe (pbsim.cpp:721-722)
                rt = (ctx.deflate_buffer(h) if deflate else h) + rt
            outs["_%04d.%s" % (i, "fq" if p.pass_num == 1 else "sam")] = rt
            outs["_%04d.maf" % i] = mt
            stats.append(ctx.stats())
    return outs, stats


def run_wgs_job(args, device=0, scratch_mb=None):
    """The same run through the job pipeline (pbsim_job_*: all records resident, one pipeline of rounds)."""
    p, a = params_from_args(args)
    outs, stats = {}, []
    with P.Context(p, device) as ctx:
        if scratch_mb:
            ctx.set_scratch_bytes(scratch_mb << 20)
        (ctx.load_errhmm if p.method == P.METHOD_ERR else ctx.load_qshmm)(a["--errhmm" if p.method == P.METHOD_ERR else "--qshmm"])
        for r in read_fasta(a["--genome"]):
            ctx.job_add_record(r)
        texts, done = ctx.job_run()
        for i in sorted(texts):
            rt, mt = bytes(texts[i][0]), bytes(texts[i][1])
            if p.pass_num > 1:
                rt = ctx.job_sam_header(i) + rt
            outs["_%04d.%s" % (i, "fq" if p.pass_num == 1 else "sam")] = rt
            outs["_%04d.maf" % i] = mt
            stats.append(done[i][0])
    return outs, stats
