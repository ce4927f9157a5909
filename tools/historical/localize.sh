#!/bin/bash
# usage: tools/localize.sh -- one failing parity case under different knob settings
for cfg in "A=1" "ARCVAE_WX_ON_SIDE=0" "ARCVAE_BPTT_CHUNKS=0.3,0.6,0.85,1.0" "ARCVAE_TABLE_ON_SIDE=0" "ARCVAE_RS_MFMA=0" "ARCVAE_FWD_MFMA=0" "ARCVAE_BPTT_CHUNKS=0.58,1.0" "ARCVAE_GATES=0" "A=2"; do
  r=$(env $cfg timeout -k 10 120 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k "persistent_sweeps and 256-2-64-12-1-env3" 2>&1 | grep -E "AssertionError: |passed|failed" | head -2 | cut -c1-300 | tr '\n' ' ')
  echo "[$cfg] $r"
done
