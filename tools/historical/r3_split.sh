#!/bin/bash
# usage (GPU box, repo root): tools/r3_split.sh   -- isolated BPTT tile launch at configs[2] with the contraction or the epilogue
# switched off (ARCVAE_TILE_DEBUG 1 / 2: timing experiments), for the 64 x 32 wave tile and the K-split 64 x 64 form
for ks in 0 1; do for dbg in 0 1 2; do
  out=$(ARCVAE_BWD_KSPLIT3=$ks ARCVAE_TILE_DEBUG=$dbg timeout -k 10 200 python bench.py --config big --roofline-only 2>/dev/null | python3 -c "import sys,json; r=json.loads(sys.stdin.read())['roofline']; print('%.2f us/launch' % r['us_per_launch'])")
  echo "KSPLIT3=$ks TILE_DEBUG=$dbg: $out"
done; done
