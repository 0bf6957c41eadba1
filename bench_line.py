"""bench_line.py -- the ONE JSON line bench.py prints, built from the detailed record it measures.

The driver parses the last stdout line of `python bench.py ...` and keeps only a tail of stdout: a line that outgrows that tail
is an unmeasured round (round 5's was 24 KB and was lost).  So the line carries numbers only -- the contract fields, `roofline`,
`cpu_baseline`, one figure per sub-measurement -- under a hard size limit, and everything else (per-rank segments, phase
tables, notes, the replay's rows) goes into a sidecar file whose path the line names.  No GPU, no torch: tests build the line
from a recorded detail fixture (tests/test_bench_line.py).
"""
import json
import os

LINE_LIMIT = 6000          # bytes; VERDICT r5 item 1
LINE_TARGET = 4096         # what compact() aims for before the optional blocks are dropped


def _num(x, digits=6):
    """floats rounded to `digits` significant figures (a line of 17-digit doubles is a third longer for nothing)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (digits, x))
    if isinstance(x, str) and len(x) > 160:
        return x[:159] + "~"
    return x


def _pick(d, keys, digits=6):
    return {k: _num(d[k], digits) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "~"


def compact(out, detail_path=None):
    """the line's object from bench.py's detailed record `out` (a dict); pure function"""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data", "reads_per_sec"), 9)
    cfg = out.get("config", {})
    line["config"] = {"workload": _short(cfg.get("workload", ""), 200)}
    line["config"].update(_pick(cfg, ("bases_per_step", "reads_per_step", "rounds_per_step", "parallelism", "comm",
                                      "rccl_ranks_seen", "collective_us", "one_gpu", "delivered", "host_runtime")))
    rf = out.get("roofline")
    if isinstance(rf, dict):
        r = _pick(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "avg_launch_ms",
                       "launches", "own_bytes_frac", "frac_rocprof", "rocprof_avg_launch_ms", "rocprof_source", "kernel_limiter",
                       "job_limiter", "valu_busy_frac", "pcie_frac", "lds_hit_rate"))
        line["roofline"] = r
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        c = _pick(cb, ("value", "unit", "cores", "kind", "error"))
        if "sample" in cb:
            c["sample"] = _short(cb["sample"], 120)
        if isinstance(cb.get("all_cores"), dict):
            c["all_cores"] = _pick(cb["all_cores"], ("value", "cores"))
        if isinstance(cb.get("philox_mode"), dict):
            c["philox_mode"] = _pick(cb["philox_mode"], ("value", "cores"))
        line["cpu_baseline"] = c
    dl = out.get("delivery")
    if isinstance(dl, dict):
        line["delivery"] = _pick(dl, ("host_bytes_per_step", "pcie_frac"))
    for k in ("whole_job_hbm", "steady_state_hbm"):
        if isinstance(out.get(k), dict) and "value" in out[k]:
            line[k] = _num(out[k]["value"])
    st = out.get("setup")
    if isinstance(st, dict):
        line["setup"] = _pick(st, ("k0_prepare_s", "value_from_fresh_records"))
    cp = out.get("critical_path")
    if isinstance(cp, dict):
        line["critical_path"] = _pick(cp, ("serial_share", "slowest_rank"))
        rows = cp.get("per_rank") or []
        if rows:   # one row, the slowest rank's: where its round loop waited (ms per step)
            w = rows[cp.get("slowest_rank", 0)] if cp.get("slowest_rank", 0) < len(rows) else rows[0]
            line["critical_path"]["slowest"] = _pick(w, ("wall", "walk", "deflate_link", "collectives", "exposed_tail"), 5)
    rp = out.get("replay")
    if isinstance(rp, dict):
        line["replay"] = {n: _pick(r, ("speedup", "speedup_if_ranks_never_wait", "max_rank_wall_ms", "collective_us"), 5)
                          for n, r in (rp.get("by_world") or {}).items()}
    cl = out.get("comm_latency")
    if isinstance(cl, dict):
        line["comm_latency"] = {k: _pick(v, ("all_gather_us", "all_reduce_us", "world")) if isinstance(v, dict) else _num(v)
                                for k, v in cl.items()}
    oc = out.get("other_configs")
    if isinstance(oc, dict):
        line["other_configs"] = {}
        for name, row in oc.items():
            if not isinstance(row, dict):
                continue
            if "error" in row:
                line["other_configs"][name] = {"error": _short(row["error"], 80)}
            else:
                line["other_configs"][name] = _pick(row, ("value", "unit", "ms_per_step", "steps", "delivered", "pcie_frac",
                                                          "walk_frac", "bases_per_sec"))
    pr = out.get("per_rank")
    if isinstance(pr, dict):
        line["per_rank"] = {k: ([_num(x) for x in v] if isinstance(v, list) else _num(v)) for k, v in pr.items()}
    if isinstance(out.get("sub_errors"), dict) and out["sub_errors"]:
        line["sub_errors"] = sorted(out["sub_errors"])      # names only; the messages are in the detail file
    if detail_path:
        line["detail"] = detail_path
    # the size guard: optional blocks go, in this order, until the line fits (the contract fields, roofline and
    # cpu_baseline never do)
    for k in ("per_rank", "critical_path", "setup", "comm_latency", "replay", "other_configs", "delivery"):
        if len(json.dumps(line)) <= LINE_TARGET:
            break
        line.pop(k, None)
    return line


def emit(out, detail_path, stream):
    """writes the detailed record to `detail_path` (best effort) and the compact line to `stream`; returns the line's text"""
    if detail_path:
        try:
            tmp = detail_path + ".tmp"
            with open(tmp, "w") as f:
                json.dump(out, f, indent=1)
            os.replace(tmp, detail_path)
        except OSError:
            detail_path = None
    text = json.dumps(compact(out, os.path.relpath(detail_path) if detail_path else None))
    if len(text) >= LINE_LIMIT:       # cannot happen with the fields above; never print an unparseable tail
        raise RuntimeError("bench line of %d bytes exceeds the %d-byte limit" % (len(text), LINE_LIMIT))
    try:        # whatever native libraries still hold in C stdio buffers (RCCL's banner) goes out BEFORE the line, not at exit
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    stream.write(text + "\n")
    stream.flush()
    return text
