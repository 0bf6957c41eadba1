#!/bin/bash
# usage (GPU box, repo root): tools/closed_ab/r03_multi_onegpu.sh TAG -- the full-size N = 8 round structure on the ONE GPU of the box
# (8 contexts, gloo collectives): plumbing and the per-rank critical path, not a scaling measurement
tag=$1
out=gpurun_out/$tag
mkdir -p $out
for w in errhmm onthq60; do
  timeout 900 python3 bench.py --gpus 8 --one-gpu --workload $w --steps 2 --warmup 1 --no-extras --no-cpu-baseline \
    > $out/bench_onegpu_n8_$w.json 2> $out/bench_onegpu_n8_$w.err
  grep '^{' $out/bench_onegpu_n8_$w.json | tail -1 > $out/tmp.json && mv $out/tmp.json $out/bench_onegpu_n8_$w.json
  tail -3 $out/bench_onegpu_n8_$w.err
done
timeout 600 python3 bench.py --gpus 2 --one-gpu --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $out/bench_onegpu_n2.json 2> $out/bench_onegpu_n2.err
grep '^{' $out/bench_onegpu_n2.json | tail -1 > $out/tmp.json && mv $out/tmp.json $out/bench_onegpu_n2.json
