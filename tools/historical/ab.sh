#!/bin/bash
# usage (on the GPU box, from the repo root): tools/ab.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...
# one bench run per environment setting (first run = baseline with no setting); one line per run in gpurun_out/ab_TAG.txt
TAG=$1; shift
OUT=gpurun_out/ab_$TAG.txt
: > $OUT
for cfg in "" "$@"; do
  line=$(env $cfg timeout -k 10 150 python bench.py --steps 200 --warmup 20 --cpu-steps 0 2> gpurun_out/ab_$TAG.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms/step  bwd-step %.2f us' % (d['ms_per_step'], d['roofline']['us_per_launch']))") || { echo "FAILED: $cfg" >> $OUT; tail -5 gpurun_out/ab_$TAG.err >> $OUT; exit 1; }
  echo "[${cfg:-baseline}] $line" | tee -a $OUT
done
