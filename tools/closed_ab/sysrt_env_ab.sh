#!/bin/bash
# GPU box: the delivered job on the SYSTEM HIP runtime (bench.py --no-torch: the `pbsim` binary's runtime) under runtime knobs
cd $GRAFT_REPO_ROOT
run() { echo -n "$* : "; env "$@" python3 bench.py --no-torch --no-cpu-baseline --steps 5 --warmup 1 --detail "" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print('ms/step %.1f  %.2f Gbases/s  pcie %.3f  walk %.2f ms' % (j['ms_per_step'], j['value']/1e9, j['delivery']['pcie_frac'], j['roofline']['avg_launch_ms']))"; }
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HSA_ENABLE_INTERRUPT=0
run GPU_MAX_HW_QUEUES=8
run GPU_MAX_HW_QUEUES=2
run HSA_ENABLE_SDMA_GANG=0
run DEBUG_CLR_BLIT_KERNARG_OPT=1
run X=2
