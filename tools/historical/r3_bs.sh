#!/bin/bash
# usage: tools/r3_bs.sh BATCH "ENV=.." ...   -> bench line (ms/step, BPTT tick isolated / in-step) per environment setting at that batch
BS=$1; shift
for cfg in "" "$@"; do
  env $cfg timeout -k 10 240 python bench.py --cpu-steps 0 --batch-per-gpu $BS --steps 100 --warmup 15 2>/dev/null | python3 -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); r = d.get('roofline', {})
    print('[bs $BS] [${cfg:-auto}] %.3f ms/step  %.0f seq/s  %s  %.2f us isolated  %s in-step' % (d['ms_per_step'], d['value'], r.get('kernel'), r.get('us_per_launch', float('nan')), '%.2f' % r['in_step_us_per_launch'] if 'in_step_us_per_launch' in r else 'n/a'))
except Exception as e:
    print('[bs $BS] [${cfg:-auto}] no line', e)
"
done
