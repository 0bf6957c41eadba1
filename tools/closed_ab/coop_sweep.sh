#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/coop_sweep.sh [lengths...]  -- the whole job with the text left in HBM, by the share of
# reads the wave walker takes (PBSIM_COOP_LEN: -1 none, n = reads of at least n bases; default = the library's choice)
L="$@"; [ -z "$L" ] && L="-1 default 45056 36096 27136 18176 9216 0"
for cl in $L; do
  if [ $cl = default ]; then unset PBSIM_COOP_LEN; else export PBSIM_COOP_LEN=$cl; fi
  python bench.py --hbm-only --no-extras --steps 3 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('coop $cl', round(d['value']/1e9,1), 'G/s', round(d['ms_per_step'],1), 'ms  walk frac', round(d['roofline']['frac'],4))"
done
