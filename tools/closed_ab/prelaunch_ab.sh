#!/bin/bash
# GPU box, EXPERIMENTAL build: the first pieces of a round's compressions launched ahead (PBSIM_DEFLATE_PRELAUNCH=1, the default)
# against launched by the delivery itself (=0): the one-GPU job and ranks 0 / 7 of eight
cd "$(dirname "$0")/../.."
one() { python3 bench.py --no-cpu-baseline --no-extras --steps 3 "$@" 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f ms per step, %.2f Gbases/s, pcie_frac %.3f' % (j['ms_per_step'], j['value']/1e9, j['delivery']['pcie_frac']))"; }
ranks() { PBSIM_REPLAY_ONLY=0,3,7 python3 bench.py --replay-ranks 8 --c1-gbs 0 --no-extras --no-cpu-baseline --steps 1 "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(', '.join('rank %d of 8: %.1f ms (busy %.1f)' % (x['rank'], x['wall_ms'], x['breakdown_ms']['worker_busy']) for x in d['replay']['by_world']['8']['per_rank']))"; }
for i in 1; do
for v in 1 0; do
  export PBSIM_DEFLATE_PRELAUNCH=$v
  echo "prelaunch=$v: configs[1] $(one)"
  echo "prelaunch=$v: configs[1] $(ranks)"
done
done
for v in 1 0; do export PBSIM_DEFLATE_PRELAUNCH=$v; echo "prelaunch=$v: configs[4] $(one --workload onthq60 --steps 1)"; done
