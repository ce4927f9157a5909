#!/bin/bash
# usage (GPU box, repo root): tools/repeat.sh TAG N "ENV=.." -- N identical bench runs with per-interval step times (run-to-run and in-run spread)
TAG=$1; N=$2; CFG=$3
OUT=gpurun_out/repeat_$TAG.txt
: > $OUT
for i in $(seq 1 $N); do
  env $CFG ARCVAE_BENCH_INTERVALS=50 timeout -k 10 150 python bench.py --steps 400 --warmup 20 --cpu-steps 0 --no-roofline 2> gpurun_out/repeat_$TAG.err > gpurun_out/repeat_$TAG.json || { echo "FAILED" >> $OUT; tail -5 gpurun_out/repeat_$TAG.err >> $OUT; exit 1; }
  grep "timed\|per interval" gpurun_out/repeat_$TAG.err | tee -a $OUT
done
