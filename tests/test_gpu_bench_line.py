"""bench.py on the GPU box as the driver runs it: the LAST stdout line must parse, stay under the size limit and carry
`roofline` + `cpu_baseline`; and the exact command shape of the scaling run -- `bench.py --gpus 2` starting
its ranks itself -- must give one parseable line with n_gpus == 2 (item 8; --one-gpu: the two ranks are two contexts on this
box's one GPU over gloo, RCCL takes one rank per GPU)."""
import json
import os
import subprocess
import sys

import pytest

import harness

sys.path.insert(0, harness.ROOT)
import bench_line  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def run_bench(args, tmp_path):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    side = str(tmp_path / "bench_detail.json")
    p = subprocess.run([sys.executable, os.path.join(harness.ROOT, "bench.py")] + args + ["--detail", side],
                       capture_output=True, text=True, env=env, cwd=harness.ROOT, timeout=850)
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    lines = p.stdout.splitlines()
    assert lines and lines[-1].startswith("{"), p.stdout[-2000:]
    assert len(lines[-1]) < bench_line.LINE_LIMIT
    return json.loads(lines[-1]), json.load(open(side)), p


def test_default_shape_small_records(tmp_path):
    line, detail, _ = run_bench(["--gpus", "1", "--steps", "1", "--warmup", "1", "--record-len", "4000000", "--records", "2"], tmp_path)
    assert line["n_gpus"] == 1 and line["steps"] == 1 and line["warmup"] == 1 and line["unit"] == "bases/s"
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert abs(line["value"] - line["config"]["bases_per_step"] / (line["ms_per_step"] / 1e3)) / line["value"] < 1e-3
    rf = line["roofline"]
    assert rf["bound"] == "hbm" and rf["kernel"] == "k_walk_errhmm" and rf["launches"] > 0
    assert abs(rf["frac"] - rf["achieved"] / 8000.0) < 1e-6 and rf["kernel_limiter"] == "valu-issue" and rf["job_limiter"] == "pcie"
    assert 0 < rf["own_bytes_frac"] < rf["frac"]
    cb = line["cpu_baseline"]
    assert cb["value"] > 1e6 and cb["cores"] == 1 and cb["kind"] in ("reference", "port") and cb["all_cores"]["value"] > cb["value"]
    # the per-collective latency of both communicators, on groups of one (item 2)
    cl = line["comm_latency"]
    assert 0 < cl["rccl_native"]["all_gather_us"] < 5000 and 0 < cl["torch_callbacks"]["all_gather_us"] < 50000
    assert detail["comm_latency"]["rccl_native"]["ranks_seen"] == 1
    assert line["whole_job_hbm"] > 0 and line["steady_state_hbm"] > 0
    # the sidecar holds what the line left out
    assert "secondary" in detail["roofline"] and "per_rank" in detail["critical_path"]


def test_scale_command_dry_run_two_ranks_one_gpu(tmp_path):
    line, detail, p = run_bench(["--gpus", "2", "--one-gpu", "--steps", "1", "--warmup", "1", "--record-len", "4000000",
                                 "--records", "2", "--no-cpu-baseline"], tmp_path)
    assert "launching 2 ranks" in p.stderr
    assert line["n_gpus"] == 2 and line["steps"] == 1 and line["value"] > 0
    assert len(line["per_rank"]["reads_delivered"]) == 2 and all(x > 0 for x in line["per_rank"]["reads_delivered"])
    assert all(x > 0 for x in line["per_rank"]["host_bytes"])
    assert sum(line["per_rank"]["reads_delivered"]) == line["config"]["reads_per_step"]
    assert line["config"]["comm"].startswith("torch.distributed/gloo")       # one GPU: RCCL takes one rank per GPU
    assert line["comm_latency"]["job_comm"]["world"] == 2 and line["comm_latency"]["job_comm"]["all_gather_us"] > 0
    assert len(detail["critical_path"]["per_rank"]) == 2
    assert "roofline" in line and line["roofline"]["launches"] > 0


def test_no_torch_mode_is_the_same_job(tmp_path):
    """bench.py --no-torch: the job from a process that maps the system HIP runtime only (the configuration rocprofv3 traces
    without turning SDMA off, profiles/r06_trace_sdma_ab.txt) -- same records (harness.synth_bases on the host), same reads and
    bases as the default run of the same size"""
    args = ["--gpus", "1", "--steps", "1", "--warmup", "1", "--record-len", "4000000", "--records", "2", "--no-cpu-baseline"]
    a, _, _ = run_bench(args + ["--no-torch"], tmp_path)
    b, _, _ = run_bench(args + ["--no-extras"], tmp_path)
    assert a["config"]["host_runtime"].startswith("system HIP runtime") and b["config"]["host_runtime"].startswith("PyTorch")
    for k in ("bases_per_step", "reads_per_step", "rounds_per_step"):
        assert a["config"][k] == b["config"][k], k
    assert a["delivery"]["host_bytes_per_step"] == b["delivery"]["host_bytes_per_step"]
    assert a["roofline"]["launches"] == b["roofline"]["launches"] and a["value"] > 0
