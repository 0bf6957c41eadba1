#!/bin/bash
# GPU box: which engine carries pinned D2H copies, plain vs under rocprofv3 --kernel-trace (AMD_LOG_LEVEL=4 names it)
R=$GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O2 -w -o /tmp/copy_path $R/tools/closed_ab/copy_path.hip || exit 1
cd /tmp && export TMPDIR=/tmp
echo "== plain"; /tmp/copy_path
echo "== plain, log"; AMD_LOG_LEVEL=4 /tmp/copy_path 2>&1 | grep -i -E "HSA Copy|blit|sdma|copyBuffer|ShaderName" | sort | uniq -c | sort -rn | head -8
echo "== HSA_ENABLE_SDMA=0"; HSA_ENABLE_SDMA=0 /tmp/copy_path
echo "== traced"; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp_t -- /tmp/copy_path; cat /tmp/cp_t/*/*kernel_stats.csv | cut -c1-100
echo "== traced, log"; AMD_LOG_LEVEL=4 rocprofv3 --kernel-trace --output-format csv -d /tmp/cp_t2 -- /tmp/copy_path 2>&1 | grep -i -E "HSA Copy|blit|sdma|copyBuffer|ShaderName|profil" | cut -c1-220 | sort | uniq -c | sort -rn | head -12
for v in "GPU_BLIT_ENGINE_TYPE=0" "HSA_ENABLE_SDMA=1" "ROCPROFILER_DISABLE_SDMA=0"; do echo "== traced, $v"; env $v true; export $v; rocprofv3 --kernel-trace --output-format csv -d /tmp/cp_t3 -- /tmp/copy_path | tail -1; unset ${v%%=*}; done
