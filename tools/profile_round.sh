#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh TAG  -- the round's judged artefacts into gpurun_out/TAG:
# the default bench line; rocprofv3 --kernel-trace --stats of the same command WITH that run's own bench line beside the CSV
# (tools/roofline_check.py re-derives the line's roofline from the CSV); the other workloads' lines; solo walk times
tag=$1
out=gpurun_out/$tag
mkdir -p $out
R=$GRAFT_REPO_ROOT
T="timeout 600"
$T python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python3 $R/bench.py --no-cpu-baseline --no-extras > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
# the same job with its text left in HBM: there the lane walk runs at five workgroups per CU (the delivered job throttles it to
# one on purpose) -- the kernel's own profile
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_hbm -- python3 $R/bench.py --no-cpu-baseline --no-extras --hbm-only > $R/$out/bench_hbm_prof.json 2> $R/$out/bench_hbm_prof.err
cd $R
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
cp $(find $out/prof_hbm -name "*kernel_stats.csv" | head -1) $out/kernel_stats_hbm.csv
$T python3 bench.py --hbm-only --no-cpu-baseline --no-extras > $out/bench_hbm_only.json 2>/dev/null
$T python3 bench.py --workload onthq60 --no-cpu-baseline --steps 1 > $out/bench_onthq60.json 2>/dev/null
$T python3 bench.py --workload qshmm10 --no-cpu-baseline --steps 1 > $out/bench_qshmm10.json 2>/dev/null
$T python3 bench.py --workload trans > $out/bench_trans.json 2>/dev/null
$T python3 bench.py --workload sample > $out/bench_sample.json 2>/dev/null
for k in errhmm onthq qshmm10; do $T python3 tools/walk_solo.py $k 3 2>/dev/null | tail -1; done > $out/walk_solo.txt
rm -rf $out/prof $out/prof_hbm
for f in $out/bench_*.json; do grep '^{' $f | tail -1 > $f.tmp && mv $f.tmp $f; done
python3 tools/roofline_check.py $out/kernel_stats.csv $out/bench_prof.json > $out/roofline_check.txt 2>&1
echo "--- the job with its text left in HBM (--hbm-only) under the tracer ---" >> $out/roofline_check.txt
python3 tools/roofline_check.py $out/kernel_stats_hbm.csv $out/bench_hbm_prof.json >> $out/roofline_check.txt 2>&1
tail -c 600 $out/bench_n1.json; echo; head -8 $out/kernel_stats.csv | cut -c1-200; cat $out/walk_solo.txt; cat $out/roofline_check.txt
