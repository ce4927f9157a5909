#!/bin/bash
# usage (GPU box, repo root): tools/r4_shard.sh TAG "ENV=.." ...
# Round 4: the 256-row shard (BASELINE.json configs[3], one GPU of eight) per environment setting -- bench line, then per-segment
# phase times of every setting (R4_PHASES=1).
TAG=$1; shift
OUT=gpurun_out/r4_shard_$TAG
mkdir -p $OUT
for cfg in "$@"; do
  name=$(echo "${cfg:-auto}" | tr ' =' '__')
  env $cfg timeout -k 10 240 python bench.py --cpu-steps 0 --batch-per-gpu ${R4_BATCH:-256} --steps 60 --warmup 10 \
      > $OUT/bench_$name.json 2> $OUT/bench_$name.log || echo "FAILED $cfg"
  python3 - "$OUT/bench_$name.json" "${cfg:-auto}" <<'EOF' | tee -a $OUT/summary.txt
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    r = d.get("roofline", {})
    print("[%s] %.3f ms/step  %.0f seq/s  kernel %s  %.2f us/launch isolated  %s in-step" % (
        sys.argv[2], d["ms_per_step"], d["value"], r.get("kernel"), r.get("us_per_launch", float("nan")),
        "%.2f" % r["in_step_us_per_launch"] if "in_step_us_per_launch" in r else "n/a"))
except Exception as e:
    print("[%s] no line (%s)" % (sys.argv[2], e))
EOF
  if [ -n "$R4_PHASES" ]; then
    echo "== phases [$cfg]" >> $OUT/phase_times.txt
    env $cfg timeout -k 10 200 python tools/phase_times.py ${R4_BATCH:-256} >> $OUT/phase_times.txt 2>&1 || echo "phase_times failed"
  fi
done
[ -n "$R4_PHASES" ] && cat $OUT/phase_times.txt
true
