#!/bin/bash
R=$GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/copy_path $R/tools/closed_ab/copy_path.hip || exit 1
cd /tmp && export TMPDIR=/tmp
TL=/usr/local/lib/python3.10/dist-packages/torch/lib
echo "== torch runtime, plain"; LD_LIBRARY_PATH=$TL /tmp/copy_path | tail -1
export LD_LIBRARY_PATH=$TL
echo "== torch runtime, traced"; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_t -- /tmp/copy_path | tail -1; cat /tmp/cp_t/*/*kernel_stats.csv | cut -c1-100
echo "== torch runtime, traced, log"; AMD_LOG_LEVEL=4 rocprofv3 --kernel-trace --output-format csv -d /tmp/cp_t2 -- /tmp/copy_path 2>&1 | grep -i -E "HSA Copy|copyBuffer|ShaderName|sdma" | cut -c1-200 | head -6
unset LD_LIBRARY_PATH
echo "== python: torch imported first, then a pinned D2H through torch"
cat > /tmp/tcopy.py <<'PY'
import torch, time
d = torch.empty(256 << 20, dtype=torch.uint8, device="cuda"); d.fill_(3)
h = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(d, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("torch D2H 256 MB: %.2f ms = %.1f GB/s" % (dt * 1e3, (256 << 20) / dt / 1e9))
PY
python3 /tmp/tcopy.py | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_t4 -- python3 /tmp/tcopy.py | tail -1; cat /tmp/cp_t4/*/*kernel_stats.csv | cut -c1-100 | head -5
