#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/nt_experiment.sh OUT   -- builds the library with PBSIM_NT = 0..3 (bit 0: nontemporal scratch
# stores in the walks, bit 1: nontemporal scratch loads in k_text_rows) and runs the bench + the walk's fabric traffic for each
out=gpurun_out/$1; mkdir -p $out
for nt in 0 1 3; do
  PBSIM_EXTRA_CFLAGS="-DPBSIM_NT=$nt" python -c "import pbsim3_amd.build as b; b.build(force=True)" > $out/build_$nt.log 2>&1
  python bench.py --no-cpu-baseline > $out/bench_$nt.json 2> $out/bench_$nt.err
  echo "NT=$nt $(cut -c1-120 $out/bench_$nt.json)"
  bash tools/stats_slots1.sh $1/s1_$nt > $out/slots1_$nt.log 2>&1; sed -n 1,4p $out/slots1_$nt.log
  bash tools/pmc_traffic.sh $out/traffic_$nt > $out/traffic_$nt.log 2>&1; tail -5 $out/traffic_$nt.log
done
