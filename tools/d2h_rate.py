"""GPU box: what the PCIe link moves device -> pinned host memory with the copy engines: one stream and two streams at once,
pieces of 83 MB (a compressed MAF piece) and 256 MB; idle GPU, with a GEMM and with an HBM-streaming kernel beside the copies.
usage: python tools/d2h_rate.py [numa node]"""
import os
import sys
import time


def cpus_of(node):
    out = set()
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        out.update(range(int(a), int(b or a) + 1))
    return out


if len(sys.argv) > 1:      # run on (and allocate from) the CPUs of one NUMA node
    os.sched_setaffinity(0, cpus_of(int(sys.argv[1])))
import torch

dev = torch.device("cuda", 0)


def run(piece_mb, streams, busy):
    n = piece_mb << 20
    src = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(streams)]
    dst = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(streams)]
    st = [torch.cuda.Stream() for _ in range(streams)]
    side = torch.cuda.Stream()
    a = torch.randn(8192, 8192, device=dev)
    global big1, big2
    reps = max(4, 2048 // piece_mb)
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if busy == "gemm":
            with torch.cuda.stream(side):
                for _ in range(6):
                    a = (a @ a).clamp_(-1, 1)
        elif busy == "hbm":     # a kernel that streams HBM at full rate beside the copies
            with torch.cuda.stream(side):
                for _ in range(40):
                    big2.copy_(big1)
        for r in range(reps):
            for i in range(streams):
                with torch.cuda.stream(st[i]):
                    dst[i].copy_(src[i], non_blocking=True)
        for s in st:
            s.synchronize()
        dt = time.perf_counter() - t0
        torch.cuda.synchronize()
    return reps * streams * n / dt / 1e9


big1 = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
big2 = torch.empty(4 << 30, dtype=torch.uint8, device=dev)
def run_cycling(piece_mb, n_dst, streams=2):
    """the same copies into n_dst different pinned buffers in turn (a working set the host's caches cannot hold)"""
    n = piece_mb << 20
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    dst = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(n_dst)]
    st = [torch.cuda.Stream() for _ in range(streams)]
    reps = max(2 * n_dst, 4096 // piece_mb)
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            with torch.cuda.stream(st[r % streams]):
                dst[r % n_dst].copy_(src, non_blocking=True)
        for s in st:
            s.synchronize()
        dt = time.perf_counter() - t0
    return reps * n / dt / 1e9


for n_dst in (1, 2, 8, 24):
    print("two streams, 83 MB pieces into %2d pinned buffers in turn (%4d MB): %.1f GB/s" % (n_dst, 83 * n_dst, run_cycling(83, n_dst)))
for busy in (None, "gemm", "hbm"):
    for piece in (83, 256):
        for streams in (1, 2):
            print("GPU %s, %3d MB pieces, %d stream(s): %.1f GB/s" % (busy or "idle", piece, streams, run(piece, streams, busy)))
