#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/prio_experiment.sh TAG -- wave priorities (s_setprio) of the text kernels vs the walk
out=gpurun_out/$1; mkdir -p $out
for f in "" "-DPBSIM_TEXT_PRIO=1" "-DPBSIM_TEXT_PRIO=3" "-DPBSIM_WALK_PRIO=0" "-DPBSIM_WALK_PRIO=0 -DPBSIM_TEXT_PRIO=2"; do
  PBSIM_EXTRA_CFLAGS="$f" python -c "import pbsim3_amd.build as b; b.build(force=True)" > $out/build.log 2>&1
  for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null > $out/b.json; echo "flags='$f' $(cut -c38-80 $out/b.json)"; done
done
