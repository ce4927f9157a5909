#!/bin/bash
# usage (GPU box): tools/tile_debug2.sh -- cost of the bf16 operand-copy stores in the K-split BPTT epilogue (results WRONG with the knob set)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for d in 0 4 8 12; do
  ARCVAE_TILE_DEBUG=$d timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_te_${d} -- python3 $R/bench.py --config big --precision bf16 --steps 4 --warmup 2 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 > $R/gpurun_out/prof_te.log 2>&1
  f=$(ls $R/gpurun_out/prof_te_${d}/*/*kernel_stats.csv | head -1)
  echo "== TILE_DEBUG=$d"; grep -E "lstm_bwd_tile_ks" $f | cut -d, -f1-4 | cut -c1-160
done
