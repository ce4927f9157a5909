#!/bin/bash
# usage (GPU box, repo root): tools/prof2.sh TAG [bench args...]  -> gpurun_out/prof_TAG/ (rocprofv3 --kernel-trace --stats of bench.py)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 --no-roofline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; echo rc=$?
