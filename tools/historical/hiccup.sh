#!/bin/bash
# usage: tools/hiccup.sh -- where does the one-time stall of the first ~100 steps fall? (10-step intervals; GC off / on)
for gcmode in 0 0 0 0 1 1 1; do
  echo "ARCVAE_BENCH_GC=$gcmode"
  ARCVAE_BENCH_GC=$gcmode ARCVAE_BENCH_INTERVALS=10 timeout -k 10 150 python bench.py --steps 200 --warmup 20 --cpu-steps 0 --no-roofline 2>&1 >/dev/null | grep "timed\|per interval"
done
