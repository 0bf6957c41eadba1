#!/bin/bash
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/r06c_trace
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --detail ''"
HSA_ENABLE_SDMA=0 timeout 600 python3 $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --detail "" > $out/plain_nosdma.json 2> $out/plain_nosdma.err
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/t_mc -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 3 --warmup 1 --detail "" > $out/traced_mc.json 2> $out/traced_mc.err
cp $(find $out/t_mc -name "*kernel_stats.csv" | head -1) $out/traced_mc.kernel_stats.csv
cp $(find $out/t_mc -name "*memory_copy_stats.csv" | head -1) $out/traced_mc.memory_copy_stats.csv
ls $out/t_mc/*/ | head; rm -rf $out/t_mc
ROCPROFILER_LOG_LEVEL=info timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t_k -- python3 $R/bench.py --no-extras --no-cpu-baseline --steps 1 --warmup 1 --detail "" > $out/traced_info.json 2> $out/traced_info.err
rm -rf $out/t_k
for n in plain_nosdma traced_mc; do grep '^{' $out/$n.json | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); rf=j['roofline']; print('$n', 'ms/step %.1f walk avg %.3f ms' % (j['ms_per_step'], rf['avg_launch_ms']))"; done
head -5 $out/traced_mc.kernel_stats.csv | cut -c1-150
cat $out/traced_mc.memory_copy_stats.csv | head
grep -i -E "sdma|blit|copy" $out/traced_info.err | head -10
env | grep -i -E "^HSA|^ROC|^HIP|^GPU_" | head -20
