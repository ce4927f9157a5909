#!/bin/bash
# usage: tools/ab_shapes.sh "bench args" "ENV=.." ...  -> ms/step and BPTT launch time for each environment setting
ARGS=$1; shift
for cfg in "" "$@"; do
  line=$(env $cfg timeout -k 10 200 python bench.py --cpu-steps 0 $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.3f ms/step  %.0f seq/s  bwd-step %.2f us  step-frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['us_per_launch'], d['step_frac_of_f32_mfma_peak']))") || line=FAILED
  echo "[$ARGS] [${cfg:-auto}] $line"
done
