#!/bin/bash
# usage: tools/r3_alone.sh BATCH "ENV=.." ...   -> the sweeps alone under each environment setting
BS=$1; shift
for cfg in "" "$@"; do
  echo "[${cfg:-auto}] $(env $cfg timeout -k 10 120 python tools/sweep_alone.py $BS 2>&1 | tail -1)"
done
