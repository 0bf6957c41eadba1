#!/bin/bash
# usage (GPU box, repo root): tools/profile_round.sh TAG  -- the round's judged artefacts into gpurun_out/TAG:
# default bench line, rocprofv3 kernel stats of the same command, qshmm10 / trans / sample lines
tag=$1
out=gpurun_out/$tag
mkdir -p $out
R=$GRAFT_REPO_ROOT
python3 bench.py > $out/bench_n1.json 2> $out/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -- python3 $R/bench.py --no-cpu-baseline > $R/$out/bench_prof.json 2> $R/$out/bench_prof.err
cd $R
cp $(find $out/prof -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 bench.py --workload qshmm10 --no-cpu-baseline > $out/bench_qshmm10.json 2>/dev/null
python3 bench.py --workload trans > $out/bench_trans.json 2>/dev/null
python3 bench.py --workload sample > $out/bench_sample.json 2>/dev/null
python3 bench.py --whole-job --no-cpu-baseline > $out/bench_whole_job.json 2>/dev/null
python3 bench.py --whole-job --deflate --no-cpu-baseline > $out/bench_whole_job_deflate.json 2>/dev/null
rm -rf $out/prof
tail -c 1500 $out/bench_n1.json; echo; head -5 $out/kernel_stats.csv | cut -c1-200
