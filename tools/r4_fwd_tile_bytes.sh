#!/bin/bash
# Is the forward tile kernel bound by the bytes its waves pull through the CU's vector memory path?  The same step with a variant
# library whose waves 1-3 do not load the (shared) A operand (wrong results, a price tag): kernel averages from a kernel trace.
# GPU box, repo root: tools/r4_fwd_tile_bytes.sh
tools/build_variant.sh skipa "-DARCVAE_DBG_SKIP_A=1" > /dev/null
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in default skipa; do
  if [ $v = skipa ]; then export ARCVAE_HIP_LIB=$R/ab_libs/libarcvae_skipa.so; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4_fwdbytes_$v -- python3 $R/bench.py --config big --steps 6 --warmup 3 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 --shard-steps 0 --sampler-reps 0 > $R/gpurun_out/r4_fwdbytes_$v.log 2>&1
  f=$(find $R/gpurun_out/r4_fwdbytes_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; grep "ms_per_step" $R/gpurun_out/r4_fwdbytes_$v.log | sed 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/'; grep "lstm_fwd_tile_kernel\|lstm_bwd_tile_ks3\|wgrad_planes" $f | cut -d, -f1-4 | cut -c1-160
done
