#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python3 bench.py --workload trans --detail "" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms/step %.1f  %.2f Gbases/s  pcie %.3f  hbm %.1f G' % (j['ms_per_step'], j['value']/1e9, j['delivery']['pcie_frac'], j['whole_job_hbm']/1e9))"; }
run PBSIM_UNITS_IN_ORDER=0 PBSIM_UNITS_RAMP=0
run PBSIM_UNITS_IN_ORDER=1 PBSIM_UNITS_RAMP=0
run PBSIM_UNITS_IN_ORDER=1 PBSIM_UNITS_RAMP=1
run PBSIM_UNITS_IN_ORDER=1 PBSIM_UNITS_RAMP=1 PBSIM_UNITS_PARTS=4 PBSIM_PIPELINE_DEPTH=4
run PBSIM_UNITS_IN_ORDER=0 PBSIM_UNITS_RAMP=1
run PBSIM_UNITS_IN_ORDER=1 PBSIM_UNITS_RAMP=1
