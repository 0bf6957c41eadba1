"""pbsim command line -> pbsim_params (the Python mirror of set_sim_param, pbsim.cpp:1451-1688,
used by the multi-GPU front-end and by the tests; the single-GPU front-end is csrc/cli.cpp)."""
from . import default_params


def parse(argv):
    """argv: ['--strategy', 'wgs', ...] -> (Params, dict of raw options)."""
    a = dict(zip(argv[::2], argv[1::2]))
    kw = {}
    kw["strategy"] = {"wgs": 1, "trans": 2, "templ": 3}[a["--strategy"][:5] if a["--strategy"].startswith("t") else "wgs"]
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
    return default_params(**kw), a


def read_fasta(path):
    """Records as get_genome_inf/get_genome_seq assemble them (pbsim.cpp:914-965, 1014-1033):
    header = line starting with '>', sequence = the concatenated lines up to the next header."""
    recs, ids, cur = [], [], None
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\n")
            if line.startswith(b">"):
                cur = []
                recs.append(cur)
                ids.append(line[1:129].decode(errors="replace"))
            elif cur is not None:
                cur.append(line)
    return [b"".join(r) for r in recs], ids
