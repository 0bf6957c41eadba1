#!/bin/bash
cd $GRAFT_REPO_ROOT

run() { echo -n "$* : "; env "$@" python3 bench.py --workload trans --detail "" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms/step %.1f  %.2f Gbases/s  pcie %.3f  hbm %.1f G' % (j['ms_per_step'], j['value']/1e9, j['delivery']['pcie_frac'], j['whole_job_hbm']/1e9))"; }
run PBSIM_UNITS_PARTS=3 PBSIM_UNITS_PRELAUNCH=0
run PBSIM_UNITS_PARTS=3 PBSIM_UNITS_PRELAUNCH=1
run PBSIM_UNITS_PARTS=6 PBSIM_UNITS_PRELAUNCH=0
run PBSIM_UNITS_PARTS=6 PBSIM_UNITS_PRELAUNCH=1
run PBSIM_UNITS_PARTS=12 PBSIM_UNITS_PRELAUNCH=1
run PBSIM_UNITS_PARTS=12 PBSIM_UNITS_PRELAUNCH=1 PBSIM_DEFLATE_PRE_PIECES=3
run PBSIM_UNITS_PARTS=24 PBSIM_UNITS_PRELAUNCH=1 PBSIM_DEFLATE_PRE_PIECES=3
run PBSIM_UNITS_PARTS=8 PBSIM_UNITS_PRELAUNCH=1 PBSIM_DEFLATE_PRE_PIECES=4 PBSIM_PIPELINE_DEPTH=4
