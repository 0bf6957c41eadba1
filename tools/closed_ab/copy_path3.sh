#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in nocompat compat torch; do
  python3 $R/tools/closed_ab/copy_path3.py $m 2>/dev/null | tail -1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_$m -- python3 $R/tools/closed_ab/copy_path3.py $m 2>/dev/null | tail -1
  echo "   traced: $(grep -c . /tmp/cp_$m/*/*kernel_stats.csv) kernels; copyBuffer: $(grep copyBuffer /tmp/cp_$m/*/*kernel_stats.csv | cut -d, -f1-4)"
done
