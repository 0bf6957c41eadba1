#!/bin/bash
# usage (GPU box, repo root): tools/pmc_round.sh TAG   -> gpurun_out/TAG_walk_pmc.json (+ the raw per-pass summaries)
# The round's counter evidence for the two ERRHMM walk kernels, each counter set in its own rocprofv3 run
# with --kernel-trace only (the pool refuses --pmc beside the other trace domains):
#   k_walk_errhmm        lane walker, one launch at a time (tools/walk_solo.py), at the delivered job's occupancy (ONE workgroup
#                        per CU, PBSIM_WALK_LDS_KB=81, since the second half of round 4; three, =41, before) and at the batch
#                        primitives' / HBM-only job's (five)
#   k_walk_errhmm_coop   wave walker with every read of a 100 000-read batch (PBSIM_COOP_LEN=0), three eight-wave workgroups per CU
#   fetch_calib          FETCH_SIZE / WRITE_SIZE against byte counts known by construction (tools/fetch_calib.hip)
tag=$1
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/${tag}_pmc
mkdir -p $out
hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib $R/tools/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
SQ2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_BRANCH"
pass() {  # pass NAME "COUNTERS" -- program args...
  local name=$1 ctrs=$2; shift; shift
  mkdir -p $out/$name
  timeout 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1
}
for c in FETCH_SIZE WRITE_SIZE; do pass calib_$c "$c" /tmp/fetch_calib; done
run_set() {  # run_set NAME -- env settings are exported by the caller
  local name=$1; shift
  timeout 600 python3 $R/tools/walk_solo.py "$@" > $out/$name.plain.log 2>&1     # the same launches without counters
  pass ${name}_sq1 "$SQ1" python3 $R/tools/walk_solo.py "$@"
  pass ${name}_sq2 "$SQ2" python3 $R/tools/walk_solo.py "$@"
  pass ${name}_fetch "FETCH_SIZE" python3 $R/tools/walk_solo.py "$@"
  pass ${name}_write "WRITE_SIZE" python3 $R/tools/walk_solo.py "$@"
}
export PBSIM_COOP_LEN=-1
PBSIM_WALK_LDS_KB=81 run_set lane1 errhmm 2
PBSIM_WALK_LDS_KB=41 run_set lane3 errhmm 2
run_set lane5 errhmm 2
export PBSIM_COOP_LEN=0
run_set coop errhmm 3 100000
unset PBSIM_COOP_LEN
cd $R
python3 tools/pmc_to_json.py $out $tag > gpurun_out/${tag}_walk_pmc.json
find $out -name "*.csv" -size +2M -delete
cat gpurun_out/${tag}_walk_pmc.json | head -c 3000
