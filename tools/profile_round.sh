#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh TAG  -- the round's judged artefacts into gpurun_out/TAG:
#   bench_n1.json (+ bench_n1_detail.json)   the default bench line as the driver runs it (PyTorch's bundled HIP runtime)
#   bench_notorch.json                        the same job from a process with the system HIP runtime only (bench.py --no-torch)
#   bench_prof.json + kernel_stats.csv        rocprofv3 --kernel-trace --stats around THAT command, its own line beside the CSV:
#                                             tools/roofline_check.py re-derives the line's roofline from the CSV.  (Around a
#                                             process that holds PyTorch's runtime the tracer turns SDMA off and every D2H copy
#                                             becomes a kernel -- another job; tools/trace_ab.sh, profiles/r06_trace_sdma_ab.txt)
#   bench_hbm_prof.json + kernel_stats_hbm.csv  the job with its text left in HBM under the tracer (five walk workgroups per CU)
#   the other workloads' lines; solo walk times
tag=$1
out=gpurun_out/$tag
mkdir -p $out
R=$GRAFT_REPO_ROOT
T="timeout 900"
$T python3 bench.py --detail $out/bench_n1_detail.json > $out/bench_n1.json 2> $out/bench_n1.err
$T python3 bench.py --no-torch --no-cpu-baseline --detail $out/bench_notorch_detail.json > $out/bench_notorch.json 2> $out/bench_notorch.err
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python3 $R/bench.py --no-torch --no-cpu-baseline --detail $R/$out/bench_prof_detail.json > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
$T rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof_hbm -- python3 $R/bench.py --no-torch --no-cpu-baseline --hbm-only --detail $R/$out/bench_hbm_prof_detail.json > $R/$out/bench_hbm_prof.json 2> $R/$out/bench_hbm_prof.err
cd $R
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
cp $(find $out/prof_hbm -name "*kernel_stats.csv" | head -1) $out/kernel_stats_hbm.csv
$T python3 bench.py --hbm-only --no-cpu-baseline --no-extras --detail "" > $out/bench_hbm_only.json 2>/dev/null
$T python3 bench.py --workload onthq60 --no-cpu-baseline --steps 1 --detail "" > $out/bench_onthq60.json 2>/dev/null
$T python3 bench.py --workload qshmm10 --no-cpu-baseline --steps 1 --detail "" > $out/bench_qshmm10.json 2>/dev/null
$T python3 bench.py --workload trans --detail "" > $out/bench_trans.json 2>/dev/null
$T python3 bench.py --workload sample --detail "" > $out/bench_sample.json 2>/dev/null
for k in errhmm onthq qshmm10; do $T python3 tools/walk_solo.py $k 3 2>/dev/null | tail -1; done > $out/walk_solo.txt
rm -rf $out/prof $out/prof_hbm
for f in $out/bench_*.json; do case $f in *_detail.json) continue;; esac; grep '^{' $f | tail -1 > $f.tmp && mv $f.tmp $f; done
python3 tools/roofline_check.py $out/kernel_stats.csv $out/bench_prof.json $out/bench_prof_detail.json > $out/roofline_check.txt 2>&1
echo "--- the untraced runs of the same job: default (PyTorch runtime) | --no-torch (system runtime) ---" >> $out/roofline_check.txt
python3 - $out >> $out/roofline_check.txt <<'PY'
import json, sys
for n in ("bench_n1", "bench_notorch", "bench_prof"):
    j = json.load(open("%s/%s.json" % (sys.argv[1], n)))
    rf = j["roofline"]
    print("%-14s ms/step %8.1f  value %.2f Gbases/s  walk avg %.3f ms x %d  frac %.4f  frac_rocprof %s" %
          (n, j["ms_per_step"], j["value"] / 1e9, rf["avg_launch_ms"], rf["launches"], rf["frac"], rf.get("frac_rocprof")))
PY
echo "--- the job with its text left in HBM (--hbm-only) under the tracer ---" >> $out/roofline_check.txt
python3 tools/roofline_check.py $out/kernel_stats_hbm.csv $out/bench_hbm_prof.json $out/bench_hbm_prof_detail.json >> $out/roofline_check.txt 2>&1
tail -c 600 $out/bench_n1.json; echo; head -8 $out/kernel_stats.csv | cut -c1-200; cat $out/walk_solo.txt; cat $out/roofline_check.txt
