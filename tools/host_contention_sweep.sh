#!/bin/bash
# usage (GPU box, repo root): tools/host_contention_sweep.sh TAG
# Eight ranks' load on the host's memory, emulated on the one-GPU box: the one-GPU configs[1] job (bench.py,
# two steps, no sub-measurements) and ranks 0 / 7 of the replayed eight-rank job, beside tools/host_load.c standing in for the
# seven other ranks' DMA writes (non-temporal stores into their NUMA nodes) and their sinks (reads).  The pool's boxes give a
# container 16 CPUs' worth of time (cpu.max 1600000 100000): a load that needs more is throttled TOGETHER WITH the job's own
# threads, so every line carries the cgroup's throttle counter around it -- a line with throttling says nothing about memory.
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_host_contention.txt
cc -O2 -pthread -o /tmp/host_load $R/tools/host_load.c || exit 1
node=$(python3 -c "
import re, pbsim3_amd as P
m = re.search(r'numa node (\d+)', str(P.bind_host_to_device(0) or ''))
print(m.group(1) if m else 0)")
echo "real rank's GPU on NUMA node $node; $(ls -d /sys/devices/system/node/node* | wc -l) node(s), $(nproc) cpus visible, cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)" > $out
thr() { grep nr_throttled /sys/fs/cgroup/cpu.stat 2>/dev/null | awk '{print $2}'; }
job() { python3 bench.py --no-extras --no-cpu-baseline --steps 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f ms per step, %.1f Gbases/s, pcie_frac %.3f' % (d['ms_per_step'], d['value']/1e9, d['delivery']['pcie_frac']))"; }
ranks() { PBSIM_REPLAY_ONLY=0,7 python3 bench.py --replay-ranks 8 --c1-gbs 0 --no-extras --no-cpu-baseline --steps 1 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(', '.join('rank %d of 8: %.1f ms (delivery thread busy %.1f)' % (x['rank'], x['wall_ms'], x['breakdown_ms']['worker_busy']) for x in d['replay']['by_world']['8']['per_rank']))"; }
line() {  # line LABEL -- host_load is running (or not)
  t0=$(thr); j=$(job); t1=$(thr); r=$(ranks); t2=$(thr)
  echo "$1" >> $out
  echo "    one GPU: $j   [throttled periods +$((t1 - t0))]" >> $out
  echo "    $r   [throttled periods +$((t2 - t1))]" >> $out
}
with_load() {  # with_load LABEL ENV... -- RATE
  label=$1; shift
  env "$@" > /tmp/hl.log 2>&1 &
  hl=$!
  sleep 1
  line "$label"
  kill -INT $hl; wait $hl 2>/dev/null
  tail -1 /tmp/hl.log | sed 's/^/    /' >> $out
}
line "quiet box"
with_load "7 x 47 GB/s of DMA-like writes (the other ranks' members arriving), no reader" HOST_LOAD_NO_READERS=1 HOST_LOAD_WRITERS=2 /tmp/host_load 8 47 900 0 $node
with_load "7 sinks reading at full speed (no writes)" HOST_LOAD_READ_ONLY=1 HOST_LOAD_READERS=1 /tmp/host_load 8 47 900 0 $node
with_load "7 x 16 GB/s written + read back" HOST_LOAD_WRITERS=1 HOST_LOAD_READERS=1 /tmp/host_load 8 16 900 0 $node
with_load "7 x 24 GB/s written + read back" HOST_LOAD_WRITERS=1 HOST_LOAD_READERS=1 /tmp/host_load 8 24 900 0 $node
with_load "7 x 47 GB/s written + read back (needs more CPU time than the container has)" HOST_LOAD_WRITERS=2 HOST_LOAD_READERS=2 /tmp/host_load 8 47 900 0 $node
cat $out
