#!/bin/bash
# usage (GPU box): tools/tile_debug.sh -- where a bf16 tiled sweep launch spends its time at BASELINE configs[2] (results are WRONG with the knob set)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2; do for ks in 1 0; do
  ARCVAE_TILE_DEBUG=$d ARCVAE_BWD_KSPLIT=$ks timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_td_${d}_$ks -- python3 $R/bench.py --config big --precision bf16 --steps 4 --warmup 2 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 > $R/gpurun_out/prof_td.log 2>&1
  f=$(ls $R/gpurun_out/prof_td_${d}_$ks/*/*kernel_stats.csv | head -1)
  echo "== TILE_DEBUG=$d KSPLIT=$ks"; grep -E "lstm_(fwd|bwd)_tile" $f | cut -d, -f1-4 | cut -c1-160
done; done
