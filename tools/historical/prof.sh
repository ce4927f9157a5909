#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh TAG
# runs the gpu tests, the bench, and a rocprofv3 kernel trace of the bench; outputs under gpurun_out/
TAG=$1
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; tail -2 gpurun_out/bench_$TAG.err
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-roofline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1; echo rc=$?
