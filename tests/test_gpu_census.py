"""Draw census: how often the reference reaches each of its rand() call sites -- measured by running
the reference itself under oracle/ref_shim.h (tests/golden/make_census.py -> census.json) -- against what the PRODUCT's
kernels counted per task on the GPU (columns walked, bases emitted, substitutions, insertions, deletions: pbsim_stats).

With the keyed stream a draw is a word of the block of its MAF column, so "draws consumed" is not a position in a stream
but a count per call-site group; the walk's control flow fixes those counts:

  ERRHMM  state draw and deletion test once per column; an emission draw per column whose deletion test does not fire; a nucleotide draw
          per substitution and insertion (+ one more per substitution of a non-ACGT base); three header draws per read
  QSHMM   quality and error-class draw once per emitted base; a state draw per emitted base of a class that has a model;
          a deletion test in front of every column but a task's first; a nucleotide draw per substitution and insertion

A walk that reached a site more or less often than the reference -- a re-initialisation too many (Q2), a deletion test at
the end of a read, an out-of-range class that re-draws when it should not (Q3) -- shows here even if some other test
missed the bytes it changes."""
import json
import os
import re

import pytest

import harness
import product
from cases import CASES

pytestmark = pytest.mark.gpu
CENSUS = {k: {int(a): b for a, b in v.items()} for k, v in json.load(open(os.path.join(harness.GOLDEN, "census.json"))).items()}


def site_roles():
    """line -> (kind, sub, slot) from the shim's own table (oracle/ref_shim.cpp SITES)"""
    text = open(os.path.join(harness.ROOT, "oracle", "ref_shim.cpp")).read()
    return {int(m.group(1)): (m.group(2), int(m.group(3)), int(m.group(4))) for m in re.finditer(r"\{(\d+),K_(\w+),(\d+),(\d+)\}", text)}


def product_stats(case):
    import pbsim3_amd as P
    args = harness.resolve(CASES[case]["args"])
    p, a = product.params_from_args(args)
    if p.strategy == P.STRATEGY_WGS:
        _, stats = product.run_wgs_job(args, scratch_mb=64)
        return p, stats
    with P.Context(p, 0) as ctx:
        (ctx.load_errhmm if p.method == P.METHOD_ERR else ctx.load_qshmm)(a["--errhmm" if p.method == P.METHOD_ERR else "--qshmm"])
        if p.strategy == P.STRATEGY_TRANS:
            ctx.load_transcript_file(a["--transcript"])
            ctx.simulate_trans(collect=False)
        else:
            ctx.load_template_file(a["--template"])
            ctx._simulate(ctx.lib.pbsim_simulate_templ, False)
        return p, [ctx.stats()]


@pytest.mark.parametrize("case", sorted(CENSUS))
def test_call_site_counts_follow_from_the_kernels_counters(case):
    import pbsim3_amd as P
    roles = site_roles()
    groups = {}
    for line, n in CENSUS[case].items():
        groups[roles[line]] = groups.get(roles[line], 0) + n
    p, stats = product_stats(case)
    reads = sum(s.res_num for s in stats)
    tasks = sum(s.res_pass_num for s in stats)
    bases = sum(s.res_len_total for s in stats)
    nsub, nins, ndel = (sum(getattr(s, f) for s in stats) for f in ("res_sub_num", "res_ins_num", "res_del_num"))
    columns = bases + ndel
    assert tasks == reads * p.pass_num
    templ = p.strategy == P.STRATEGY_TEMPL
    assert groups.get(("HDR", 0, 1), 0) == reads                                   # the accuracy draw: every read
    assert groups.get(("HDR", 0, 0), 0) == (0 if templ else reads)                   # the length draw
    assert groups.get(("HDR", 0, 2), 0) <= reads                                   # offset / start bucket (none when the read is the record)
    nuc = groups.get(("WALK", 0, 3), 0)
    assert nuc == nsub + nins, (nuc, nsub, nins)
    if p.method == P.METHOD_ERR:
        state, deltest, emis = (groups.get(("WALK", 0, k), 0) for k in (0, 1, 2))
        assert state == deltest
        # an emission draw per column whose deletion test did not fire.  Classes outside the model's range re-draw (Q3): below
        # it a match may become a deletion after its emission draw (%3 + 1, pbsim.cpp:3893-3895), above it a fired deletion
        # may become a match without one (:3921) -- so the count is exact without such classes and bounded by their re-draws
        redraw_test, redraw_type = groups.get(("WALK", 1, 0), 0), groups.get(("WALK", 1, 1), 0)
        assert -redraw_test <= emis - (state - ndel) <= redraw_type, (emis, state, ndel, redraw_test, redraw_type)
        # accuracy class 100 is copied verbatim (pbsim.cpp:3837-3845): those tasks draw nothing
        verbatim = "acc98" in case or "acc99" in case
        assert (state <= columns) if verbatim else (state == columns), (state, columns)
        assert groups.get(("WALK", 1, 2), 0) <= nsub                              # non-ACGT substitutions
        assert groups.get(("WALK", 1, 1), 0) <= groups.get(("WALK", 1, 0), 0) <= columns   # the out-of-range classes' re-draws (Q3)
    else:
        quality, errclass, state = (groups.get(("WALK", 0, k), 0) for k in (1, 2, 0))
        assert quality == errclass == bases, (quality, errclass, bases)
        assert state <= bases                                                      # classes without a model draw no state
        assert groups.get(("WALK", 2, 0), 0) == columns - tasks, (groups.get(("WALK", 2, 0), 0), columns, tasks)
        assert groups.get(("WALK", 1, 0), 0) <= nsub
