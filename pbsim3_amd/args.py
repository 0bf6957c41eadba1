"""pbsim command line -> pbsim_params: a TEST-SIDE mirror of the option table (tests/product.py drives the C ABI from a
command line with it).  It maps options to fields and nothing more -- the validation of set_sim_param (pbsim.cpp:1451-1688)
lives in csrc/cli.cpp (pbsim_cli_main), which every front-end of the product goes through (the `pbsim` binary,
pbsim3_amd.run_multi)."""
from . import default_params


def parse(argv):
    """argv: ['--strategy', 'wgs', ...] -> (Params, dict of raw options)."""
    a = dict(zip(argv[::2], argv[1::2]))
    kw = {}
    kw["strategy"] = {"wgs": 1, "trans": 2, "templ": 3}[a["--strategy"][:5] if a["--strategy"].startswith("t") else "wgs"]
    kw["method"] = {"qshmm": 1, "errhmm": 2, "sample": 3}[a["--method"]]
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


def read_sample_fastq(path, len_min=100, len_max=1000000, acc_min=0.75, acc_max=1.0):
    """The filtered quality strings of get_sample_inf (pbsim.cpp:1216-1283) for a well-formed 4-line FASTQ:
    length within [len_min, len_max], accuracy 1 - mean(10^(-Q/10)) within [acc_min, acc_max], file order."""
    out = []
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    for i in range(3, len(lines), 4):
        q = lines[i]
        if not (len_min <= len(q) <= len_max):
            continue
        prob = 0.0
        for ch in q:                      # same summation order as the reference
            prob += 10 ** ((ch - 33) / -10)
        acc = 1.0 - prob / len(q)
        if acc_min <= acc <= acc_max:
            out.append(q)
    return out
