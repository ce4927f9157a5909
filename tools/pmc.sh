#!/bin/bash
# usage (GPU box, repo root): tools/pmc.sh TAG [bench args]  -> gpurun_out/pmc_TAG_<COUNTERS>/ , one rocprofv3 --pmc pass per counter set
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass).  Counter collection serialises kernel
# dispatches, so the device-side gates are switched off (a gate would spin until its producer is let through) and the
# step runs eagerly (every kernel is its own dispatch).
TAG=$1; shift     # further arguments go to bench.py (e.g. --config big: the MFMA-bound regime's kernels)
cd /tmp && export TMPDIR=/tmp
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  D=$GRAFT_REPO_ROOT/gpurun_out/pmc_${TAG}_$(echo $C | tr ' ' '+')
  ARCVAE_GATES=0 timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 --shard-steps 0 --sampler-reps 0 --mode eager "$@" > $D.log 2>&1
  echo "$C rc=$?"
done
