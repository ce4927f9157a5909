#!/bin/bash
# Repeats one training step per configuration and counts steps whose result differs from the first (tools/race_hunt.py):
# every sweep form the engine can pick (rounds 3 and 4: + the fused seam kernel, the split finish, plane weight gradients beside the persistent 32-row BPTT), at the batch sizes that select it (the last three: the MFMA-bound regime's
# tile kernels, operand-plane weight gradients and dense decoder stack, forced on at small shapes).  Run on the GPU box; the output is
# committed as profiles/rNN_race_hunt.txt together with the commit it ran on.
N=${1:-1500}
for cfg in "RH_B=64" "RH_B=64 ARCVAE_RS_MFMA=0" "RH_B=64 ARCVAE_PERSIST_BWD=0" "RH_B=64 ARCVAE_GATES=0" "RH_B=64 ARCVAE_WX_ON_SIDE=0" \
           "RH_B=37 RH_T=19" "RH_B=128" "RH_B=128 ARCVAE_RS_R16=0" "RH_B=128 ARCVAE_PERSIST2=0" "RH_B=100 RH_T=17" \
           "RH_B=256" "RH_B=256 ARCVAE_PERSIST2=0" "RH_B=256 ARCVAE_RS_MAX_B=256 ARCVAE_PERSIST2=0" \
           "RH_B=256 ARCVAE_RS_MAX_B=256 ARCVAE_PERSIST2=0 ARCVAE_RS_R16=0" "RH_B=200 RH_T=9 RH_L=1" \
           "RH_B=512 RH_T=8 ARCVAE_LSTM_SPLIT3=0" "RH_B=512 RH_T=8" \
           "RH_B=512 RH_T=8 ARCVAE_STEP_TILE=4 ARCVAE_BWD_KSPLIT3=2 ARCVAE_DENSE_TILED=2" \
           "RH_B=288 RH_T=20 ARCVAE_STEP_TILE=4 ARCVAE_BWD_KSPLIT3=0 ARCVAE_DENSE_TILED=2" \
           "RH_B=512 RH_T=8 ARCVAE_STEP_TILE=4 ARCVAE_WGRAD_PLANES=0" \
           "RH_B=64 ARCVAE_SEAM_FUSED=1" "RH_B=37 RH_T=19 ARCVAE_SEAM_FUSED=1" "RH_B=64 ARCVAE_MERGE_FINISH=0" \
           "RH_B=256 ARCVAE_RS_MAX_B=256 ARCVAE_WGRAD_CONVERT=1" "RH_B=256 ARCVAE_RS_HALVES=1" "RH_B=200 RH_T=17 ARCVAE_RS_HALVES=1"; do
  echo "== $cfg"
  env $cfg timeout -k 10 150 python tools/race_hunt.py $N 2>&1 | grep -v amdgpu.ids | tail -8
done
