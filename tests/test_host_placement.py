"""Host-side pieces around the job that need no GPU: NUMA placement of a rank (csrc/numa_bind.cpp, dry run on a fake
sysfs tree) and bench.py's launch contract (`--gpus N` starts N ranks itself or refuses)."""
import os
import subprocess
import sys

import pbsim3_amd as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_tree(root, gpus):
    """gpus: list of (domain, bus, dev, fn, numa_node, cpulist); KFD nodes 0..1 are CPUs (simd_count 0), then the GPUs"""
    nodes = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    for i in range(2):
        os.makedirs(os.path.join(nodes, str(i)))
        open(os.path.join(nodes, str(i), "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for k, (dom, bus, dev, fn, node, cpus) in enumerate(gpus):
        d = os.path.join(nodes, str(2 + k))
        os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write(
            "cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain %d\ngfx_target_version 90500\n" % ((bus << 8) | (dev << 3) | fn, dom))
        p = os.path.join(root, "sys/bus/pci/devices/%04x:%02x:%02x.%x" % (dom, bus, dev, fn))
        os.makedirs(p)
        open(os.path.join(p, "numa_node"), "w").write("%d\n" % node)
        open(os.path.join(p, "local_cpulist"), "w").write(cpus + "\n")


def bind(device, root, **env):
    code = ("import pbsim3_amd as P, sys; sys.stdout.write(P.bind_host_to_device(%d))" % device)
    e = dict(os.environ, PBSIM_SYSFS_ROOT=root, **env)
    for k in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL", "ROCR_VISIBLE_DEVICES", "PBSIM_NUMA_BIND"):
        if k not in env:
            e.pop(k, None)
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    return p.stdout


def test_numa_binding_follows_the_gpu(tmp_path):
    ncpu = os.cpu_count() or 1
    allc = "0-%d" % (ncpu - 1)
    root = str(tmp_path)
    make_tree(root, [(0, 0x05, 0, 0, 0, allc), (0, 0x15, 0, 0, 0, allc), (1, 0x85, 0, 0, 1, "0"), (1, 0x95, 0, 0, -1, allc)])
    assert bind(0, root) == "gpu 0 (0000:05:00.0): numa node 0, %d cpus (dry run)" % ncpu
    assert bind(2, root) == "gpu 2 (0001:85:00.0): numa node 1, 1 cpus (dry run)"
    assert bind(3, root) == ""                       # the platform does not name a node: unbound
    assert bind(7, root) == ""                       # no such GPU
    # visible-device lists remap the index (HIP indexes into what ROCr left)
    assert bind(0, root, HIP_VISIBLE_DEVICES="2,0").startswith("gpu 0 (0001:85:00.0)")
    assert bind(1, root, ROCR_VISIBLE_DEVICES="1,2", HIP_VISIBLE_DEVICES="1,0").startswith("gpu 1 (0000:15:00.0)")
    assert bind(0, root, ROCR_VISIBLE_DEVICES="GPU-abcdef") == ""   # UUIDs: left alone
    # HIP honours CUDA_VISIBLE_DEVICES (what torch launchers set) and GPU_DEVICE_ORDINAL like its own variable
    assert bind(0, root, CUDA_VISIBLE_DEVICES="2").startswith("gpu 0 (0001:85:00.0)")
    assert bind(1, root, GPU_DEVICE_ORDINAL="2,1").startswith("gpu 1 (0000:15:00.0)")
    assert bind(0, root, CUDA_VISIBLE_DEVICES="2", HIP_VISIBLE_DEVICES="2").startswith("gpu 0 (0001:85:00.0)")
    assert bind(0, root, CUDA_VISIBLE_DEVICES="2", HIP_VISIBLE_DEVICES="1") == ""   # they disagree: unbound, not mis-bound
    assert bind(0, root, PBSIM_NUMA_BIND="0") == ""


def test_numa_binding_without_topology_is_a_noop(tmp_path):
    assert bind(0, str(tmp_path)) == ""
    assert P.bind_host_to_device(-1) == ""


def run_bench(args, **env):
    e = dict(os.environ, **env)
    if "WORLD_SIZE" not in env:
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            e.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e,
                          cwd=ROOT, timeout=600)


def test_bench_refuses_a_world_that_is_not_gpus():
    p = run_bench(["--gpus", "2"], WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and "--gpus 2" in p.stderr and "3 rank" in p.stderr
    p = run_bench(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert p.returncode != 0 and "--gpus 1" in p.stderr


def test_bench_launches_its_ranks_itself():
    """bare `bench.py --gpus 2`: the parent starts two ranks under torch.distributed.run before touching a GPU and leaves
    with their status (here they fail: this container has no GPU -- each rank says so, nobody hangs)."""
    p = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert "launching 2 ranks" in p.stderr and "torch.distributed.run" in p.stderr
    assert "--nproc-per-node=2" in p.stderr
    import torch
    if not torch.cuda.is_available():
        assert p.returncode != 0
        assert "--gpus 2 but this node shows 0 GPU(s)" in p.stderr


def test_cli_launch_modes_refuse_contradictions():
    """the `pbsim` binary's one-process-per-GPU front-ends check their arguments before any HIP call (no GPU needed)"""
    import pbsim3_amd.build as b
    b.build()
    cli = os.path.join(ROOT, "pbsim3_amd", "bin", "pbsim")
    p = subprocess.run([cli, "--processes", "2", "--devices", "0,0"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 255 and "--processes N starts the ranks itself" in p.stderr
    p = subprocess.run([cli, "--rank", "2", "--world", "2", "--rendezvous", "x"], capture_output=True, text=True, timeout=60)
    assert p.returncode == 255 and "--rank R --world N --rendezvous FILE" in p.stderr
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "OMPI_COMM_WORLD_RANK",
                                                            "OMPI_COMM_WORLD_SIZE", "PMI_RANK", "PMI_SIZE", "SLURM_PROCID", "SLURM_NTASKS")}
    p = subprocess.run([cli, "--rendezvous", "x"], capture_output=True, text=True, timeout=60, env=env)   # no launcher around it
    assert p.returncode == 255 and "--rank R --world N --rendezvous FILE" in p.stderr
