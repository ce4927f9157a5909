#!/bin/bash
# usage (GPU box, repo root): tools/r3_big.sh TAG ["ENV=.." ...]
# BASELINE.json configs[2] (H512 L4 bs 512) and the 2048-row strong leg under environment settings: ms/step, the BPTT tile
# launch alone and in the step.  R3_ONLY="--config big" (or "--batch-per-gpu 2048"): that shape only.
TAG=$1; shift
OUT=gpurun_out/r3_big_$TAG
mkdir -p $OUT
for cfg in "" "$@"; do
  name=$(echo "${cfg:-auto}" | tr ' =/' '___')
  for shape in "--config big" "--batch-per-gpu 2048"; do
    if [ -n "$R3_ONLY" ] && [ "$shape" != "$R3_ONLY" ]; then continue; fi
    sn=$(echo "$shape" | tr ' -' '__')
    env $cfg timeout -k 10 300 python bench.py --cpu-steps 0 $shape --steps 8 --warmup 3 --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 --shard-steps 0 --sampler-reps 0 \
        > $OUT/bench_${name}${sn}.json 2> $OUT/bench_${name}${sn}.log || echo "FAILED $cfg $shape"
    python3 - "$OUT/bench_${name}${sn}.json" "${cfg:-auto} $shape" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    r = d.get("roofline", {})
    print("[%s] %.3f ms/step  kernel %s  %.2f us/launch isolated  %s in-step" % (
        sys.argv[2], d["ms_per_step"], r.get("kernel"), r.get("us_per_launch", float("nan")),
        "%.2f" % r["in_step_us_per_launch"] if "in_step_us_per_launch" in r else "n/a"))
except Exception as e:
    print("[%s] no line (%s)" % (sys.argv[2], e))
PY
  done
done
