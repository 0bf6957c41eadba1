"""GPU box: the host-side rates the end-to-end CLI depends on.
  1. pwrite() of pinned-sized pieces into a /dev/shm file from 1..32 threads (the sink of the compressed members)
  2. host -> device copies of a 760 MB record: from a mmap'ed /dev/shm file (page cache), from malloc'ed memory, from pinned memory
usage: python tools/closed_ab/host_io_rates.py"""
import mmap
import os
import threading
import time

import numpy as np
import torch

d = "/dev/shm/pbsim_io_test"
os.makedirs(d, exist_ok=True)
piece = 64 << 20
src = np.random.default_rng(1).integers(0, 255, piece, dtype=np.uint8).tobytes()
total = 16 << 30
for nt in (1, 2, 4, 8, 16, 32):
    path = os.path.join(d, "w%d" % nt)
    fd = os.open(path, os.O_CREAT | os.O_WRONLY | os.O_TRUNC, 0o644)
    n_pieces = total // piece

    def work(t):
        for k in range(t, n_pieces, nt):
            os.pwrite(fd, src, k * piece)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(nt)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.perf_counter() - t0
    os.close(fd)
    os.unlink(path)
    print("pwrite into /dev/shm: %2d threads  %.1f GB/s" % (nt, total / dt / 1e9), flush=True)

n = 760_000_000
path = os.path.join(d, "rec")
with open(path, "wb") as f:
    f.write(np.random.default_rng(2).integers(65, 90, n, dtype=np.uint8).tobytes())
dev = torch.device("cuda", 0)
dst = torch.empty(n, dtype=torch.uint8, device=dev)
f = open(path, "rb")
mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
t_map = torch.frombuffer(mm, dtype=torch.uint8)
t_mal = torch.from_numpy(np.frombuffer(open(path, "rb").read(), dtype=np.uint8).copy())
t_pin = t_mal.pin_memory()
for name, t in (("mmap'ed /dev/shm file", t_map), ("malloc'ed memory", t_mal), ("pinned memory", t_pin)):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        dst.copy_(t, non_blocking=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("H2D 760 MB from %-24s %.1f GB/s" % (name, n / dt / 1e9), flush=True)
t0 = time.perf_counter()
x = t_mal.pin_memory()
print("pinning 760 MB (copy into fresh page-locked memory): %.2f s" % (time.perf_counter() - t0))
os.unlink(path)
os.rmdir(d)
