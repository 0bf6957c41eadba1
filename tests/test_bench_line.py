"""bench.py's printed line (bench_line.py): the driver parses the LAST stdout line and keeps only a tail of stdout, so the line
has a hard size limit and everything else goes to a sidecar file.  Built here from recorded detail records: round 5's
(tests/golden/bench/r05_detail_n1.json -- the very 24 KB record whose line the driver lost) and a synthetic eight-rank one."""
import io
import json
import os
import sys

import harness

sys.path.insert(0, harness.ROOT)
import bench_line  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def detail():
    return json.load(open(os.path.join(harness.ROOT, "tests", "golden", "bench", "r05_detail_n1.json")))


def check(text, want_n):
    assert "\n" not in text
    assert len(text) < bench_line.LINE_LIMIT, len(text)
    line = json.loads(text)
    for k in CONTRACT:
        assert k in line, k
    assert line["n_gpus"] == want_n
    rf = line["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "alg_bytes_per_launch", "avg_launch_ms", "launches"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = line["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind"):
        assert k in cb, k
    assert "workload" in line["config"] and "bases_per_step" in line["config"]
    return line


def test_round5_record_fits_and_round_trips(tmp_path):
    d = detail()
    assert len(json.dumps(d)) > 20000          # the record that was lost as a line
    out = io.StringIO()
    side = str(tmp_path / "bench_detail.json")
    text = bench_line.emit(d, side, out)
    assert out.getvalue() == text + "\n"
    line = check(text, 1)
    assert len(text) <= bench_line.LINE_TARGET + 200
    assert abs(line["value"] - d["value"]) / d["value"] < 1e-8 and line["steps"] == 20 and line["warmup"] == 5
    assert abs(line["ms_per_step"] - d["ms_per_step"]) < 1e-3
    assert line["whole_job_hbm"] > 1e11 and line["steady_state_hbm"] > 1e11
    assert set(line["other_configs"]) == set(d["other_configs"])
    assert line["cpu_baseline"]["all_cores"]["value"] > 0
    assert "8" in line["replay"]
    assert json.load(open(side)) == d           # nothing is lost: the sidecar holds the full record
    assert line["detail"].endswith("bench_detail.json")


def test_eight_rank_record_fits():
    d = detail()
    d["n_gpus"] = 8
    row = d["critical_path"]["per_rank"][0]
    d["critical_path"]["per_rank"] = [dict(row, rank=r) for r in range(8)]
    d["critical_path"]["slowest_rank"] = 7
    d["per_rank"] = {"reads_delivered": [847786 + r for r in range(8)], "host_bytes": [6954766179 + r for r in range(8)]}
    d["comm_latency"] = {"job_comm": {"world": 8, "words": 8, "iters": 300, "all_gather_us": 61.234567, "all_reduce_us": 58.7654321,
                                      "kind": "rccl-native (ncclCommInitRank)"}}
    d["config"].update({"comm": "rccl-native (ncclCommInitRank)", "rccl_ranks_seen": 8})
    d["roofline"].update({"frac_rocprof": 0.0413, "own_bytes_frac": 0.0556, "kernel_limiter": "valu-issue", "job_limiter": "pcie",
                          "valu_busy_frac": 0.44, "pcie_frac": 0.82, "rocprof_source": "profiles/r06z_bench_prof_kernel_stats.csv"})
    text = json.dumps(bench_line.compact(d, "bench_detail.json"))
    line = check(text, 8)
    assert line["per_rank"]["reads_delivered"][7] == 847793
    assert line["config"]["rccl_ranks_seen"] == 8
    assert line["roofline"]["frac_rocprof"] == 0.0413 and line["roofline"]["job_limiter"] == "pcie"
    assert line["comm_latency"]["job_comm"]["all_gather_us"] > 60


def test_bloated_record_drops_optional_blocks_not_the_contract():
    d = detail()
    d["other_configs"] = {"configs[%d] x" % i: {"value": 1.0 * i, "unit": "bases/s", "ms_per_step": 3.0} for i in range(200)}
    d["config"]["workload"] = "w" * 5000
    text = json.dumps(bench_line.compact(d, None))
    check(text, 1)


def test_errors_in_sub_measurements_still_give_a_line():
    d = detail()
    d["cpu_baseline"] = {"error": "x" * 3000}
    d["other_configs"]["configs[3] trans"] = {"error": "boom " * 1000, "command": "..."}
    d["roofline"]["traffic"] = None
    d["value"] = float("nan")
    line = json.loads(json.dumps(bench_line.compact(d, None)))
    assert line["value"] is None and line["roofline"]["traffic"] is None
    assert len(json.dumps(line)) < bench_line.LINE_LIMIT
