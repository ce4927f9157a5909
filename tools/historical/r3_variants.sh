#!/bin/bash
# usage (GPU box, repo root): tools/r3_variants.sh TAG "shape args" LIB1 LIB2 ... -- "ENV=.." ...
# One bench line per (library variant from ab_libs/, environment setting) at the given shape.
TAG=$1; SHAPE=$2; shift 2
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; shift
OUT=gpurun_out/r3_var_$TAG; mkdir -p $OUT
for lib in "${LIBS[@]}"; do
  for cfg in "" "$@"; do
    name=${lib}_$(echo "${cfg:-auto}" | tr ' =' '__')
    env ARCVAE_HIP_LIB=$PWD/ab_libs/libarcvae_$lib.so $cfg timeout -k 10 300 python bench.py --cpu-steps 0 $SHAPE --steps 8 --warmup 3 \
        --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 --shard-steps 0 --sampler-reps 0 > $OUT/$name.json 2> $OUT/$name.log || echo "FAILED $lib $cfg"
    python3 - "$OUT/$name.json" "$lib ${cfg:-auto}" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    r = d.get("roofline", {})
    print("[%s] %.3f ms/step  %.2f us/launch isolated  %s in-step" % (
        sys.argv[2], d["ms_per_step"], r.get("us_per_launch", float("nan")),
        "%.2f" % r["in_step_us_per_launch"] if "in_step_us_per_launch" in r else "n/a"))
except Exception as e:
    print("[%s] no line (%s)" % (sys.argv[2], e))
PY
  done
done
