#!/bin/bash
# usage (GPU box): tools/tail_timeline.sh TAG -- the step's exposed ends with the heads' parameter gradients behind chunk 0 (round-2 order
# until item 11 of profiles/r02_wgrad_experiments.txt) and at the sweep's start (now)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in 0 1; do
  ARCVAE_HEADS_EARLY=$e timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_tail_${1}_$e -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 > $R/gpurun_out/prof_tail.log 2>&1
  f=$(ls $R/gpurun_out/prof_tail_${1}_$e/*/*kernel_trace.csv | head -1)
  echo "==== ARCVAE_HEADS_EARLY=$e"; python3 $R/tools/tail_timeline.py $f 12
done
