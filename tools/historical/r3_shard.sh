#!/bin/bash
# usage (GPU box, repo root): tools/r3_shard.sh TAG ["ENV=.." ...]
# The 256-row shard of BASELINE.json configs[3] on one GPU: bench line per environment setting (ms/step, the isolated
# BPTT launch / tick, the in-step cadence), then the per-segment phase times and a kernel trace of the default setting.
TAG=$1; shift
OUT=gpurun_out/r3_shard_$TAG
mkdir -p $OUT
for cfg in "" "$@"; do
  name=$(echo "${cfg:-auto}" | tr ' =' '__')
  env $cfg timeout -k 10 240 python bench.py --cpu-steps 0 --batch-per-gpu 256 --steps 60 --warmup 10 \
      > $OUT/bench_$name.json 2> $OUT/bench_$name.log || echo "FAILED $cfg"
  python3 - "$OUT/bench_$name.json" "${cfg:-auto}" <<'EOF'
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
done
timeout -k 10 200 python tools/phase_times.py 256 > $OUT/phase_times.txt 2>&1 || echo "phase_times failed"
tail -20 $OUT/phase_times.txt
if [ -z "$R3_NO_TRACE" ]; then
  tools/prof2.sh r3shard_$TAG --batch-per-gpu 256 --steps 30 --warmup 10
  python3 tools/stats.py r3shard_$TAG 24
fi
