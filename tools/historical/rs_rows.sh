#!/bin/bash
# usage (GPU box): tools/rs_rows.sh -- the reduce-scatter BPTT sweep at 64 / 128 / 256 rows per GPU: isolated tick and whole step, f32 and bf16,
# and (if mlx-vae_amd/arcvae_hip/libarcvae_hip_alt.so exists) the same for the alternative build
R=$GRAFT_REPO_ROOT
pr='import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline"); print(sys.argv[1], ("tick us %.3f" % r["us_per_launch"]) if r else ("step ms %.4f  seq/s %d  loss %.6f" % (d["ms_per_step"], d["value"], d["elbo"]["total"])))'
for lib in libarcvae_hip.so libarcvae_hip_alt.so; do
  [ -f $R/mlx-vae_amd/arcvae_hip/$lib ] || continue
  export ARCVAE_HIP_LIB=$R/mlx-vae_amd/arcvae_hip/$lib
  for b in 64 128 256; do for p in fp32 bf16; do
    ARCVAE_RS_MAX_B=256 timeout -k 10 200 python bench.py --roofline-only --precision $p --batch-per-gpu $b 2>/dev/null | python -c "$pr" "$lib B=$b $p RS_MAX_B=256"
  done; done
  for e in 128 256; do for p in fp32 bf16; do
    ARCVAE_RS_MAX_B=$e timeout -k 10 200 python bench.py --precision $p --batch-per-gpu 256 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 2>/dev/null | python -c "$pr" "$lib B=256 $p RS_MAX_B=$e"
  done; done
  for p in fp32 bf16; do
    timeout -k 10 200 python bench.py --precision $p --batch-per-gpu 128 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 2>/dev/null | python -c "$pr" "$lib B=128 $p"
  done
done
