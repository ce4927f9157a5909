#!/bin/bash
# The plane GEMM alone per knob setting: tools/r4_planes_knobs.sh TAG "ENV=.. ENV=.." ...   (each argument one run of bench_planes.py)
tag=$1; shift
out=gpurun_out/r4_planes_$tag; mkdir -p $out
for cfg in "$@"; do
  name=$(echo "$cfg" | tr ' =' '__')
  echo "== $cfg" | tee -a $out/summary.txt
  env $cfg python tools/bench_planes.py 2>&1 | grep "^planes" | tee -a $out/summary.txt
done
