#!/bin/bash
for cfg in "A=1" "ARCVAE_RS_MFMA=0" "ARCVAE_BPTT_CHUNKS=0.3,0.6,0.85,1.0" "ARCVAE_WX_ON_SIDE=0" "ARCVAE_INPLACE_DG=0" "ARCVAE_PERSIST_BWD=0" "ARCVAE_GATES=0" "ARCVAE_RS_WREG=0"; do
  echo "== $cfg"
  env $cfg timeout -k 10 100 python tools/race_hunt.py 1500 2>&1 | grep -v amdgpu.ids | tail -8
done
