"""Drives the HIP product through its C ABI from a pbsim command line (test
helper; mirrors what pbsim3_amd/csrc/cli.cpp does in C++)."""
import os

import pbsim3_amd as P


def read_fasta(path):
    """Records as get_genome_inf/get_genome_seq assemble them (pbsim.cpp:914-965,
    1014-1033): header = line starting with '>', sequence = concatenated lines."""
    recs, cur = [], None
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\n")
            if line.startswith(b">"):
                cur = []
                recs.append(cur)
            elif cur is not None:
                cur.append(line)
    return [b"".join(r) for r in recs]


def params_from_args(args):
    a = dict(zip(args[::2], args[1::2]))
    kw = {}
    kw["strategy"] = {"wgs": 1, "trans": 2, "templ": 3}[a["--strategy"]]
    kw["method"] = {"qshmm": 1, "errhmm": 2}[a["--method"]]
    if "--seed" in a:
        kw["seed"] = int(a["--seed"])
    if "--depth" in a:
        kw["depth"] = float(a["--depth"])
    if "--length-mean" in a:
        kw["len_mean"] = float(a["--length-mean"])
    if "--length-sd" in a:
        kw["len_sd"] = float(a["--length-sd"])
    if "--length-min" in a:
        kw["len_min"] = int(a["--length-min"])
    if "--length-max" in a:
        kw["len_max"] = int(a["--length-max"])
    if "--accuracy-mean" in a:
        kw["accuracy_mean"] = int(float(a["--accuracy-mean"]) * 100) * 0.01  # pbsim.cpp:1660
    if "--pass-num" in a:
        kw["pass_num"] = int(a["--pass-num"])
    if "--hp-del-bias" in a:
        kw["hp_del_bias"] = float(a["--hp-del-bias"])
    if "--difference-ratio" in a:
        s, i, d = (int(x) for x in a["--difference-ratio"].split(":"))
        kw.update(sub_ratio=s, ins_ratio=i, del_ratio=d)
    if "--id-prefix" in a:
        kw["id_prefix"] = a["--id-prefix"]
    return P.default_params(**kw), a


def run_wgs(args, device=0, scratch_mb=None):
    """Returns {'_0001.fq': bytes, '_0001.maf': bytes, ...} plus per-record Stats."""
    p, a = params_from_args(args)
    outs, stats = {}, []
    with P.Context(p, device) as ctx:
        if scratch_mb:
            ctx.set_scratch_bytes(scratch_mb << 20)
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
                rt = ctx.sam_header() + rt  # main() writes it when opening the pipe (pbsim.cpp:721-722)
            outs["_%04d.%s" % (i, "fq" if p.pass_num == 1 else "sam")] = rt
            outs["_%04d.maf" % i] = mt
            stats.append(ctx.stats())
    return outs, stats
