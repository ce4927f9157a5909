#!/bin/bash
# usage (GPU box, repo root): tools/bf16_measure.sh TAG -- throughput mode next to the fp32 path: tests, bench lines, per-kernel profile
TAG=$1
R=$GRAFT_REPO_ROOT
python -m pytest tests/test_bf16_mode_gpu.py -x -q -s > gpurun_out/bf16_tests_$TAG.log 2>&1; tail -8 gpurun_out/bf16_tests_$TAG.log
show='import sys,json; d=json.loads(sys.stdin.read()); print(d["dtype"], d["config"]["workload"][:40], round(d["ms_per_step"],3), round(d["value"]), d["elbo"])'
rm -f gpurun_out/bf16_bench_$TAG.jsonl
for a in "--config big --precision fp32 --steps 10 --warmup 3" "--config big --precision bf16 --steps 10 --warmup 3" "--precision bf16" "--precision bf16 --batch-per-gpu 256" "--precision bf16 --batch-per-gpu 2048 --steps 30 --warmup 5"; do
  timeout -k 10 200 python bench.py $a --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 2>/dev/null | tee -a gpurun_out/bf16_bench_$TAG.jsonl | python -c "$show"
done
for m in 1 2 4; do
  ARCVAE_BF16_PARTS=$m timeout -k 10 200 python bench.py --config big --precision bf16 --steps 10 --warmup 3 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 2>/dev/null | python -c "$show" | sed "s/^/parts=$m /"
done
ARCVAE_BF16_PARTS=2 timeout -k 10 200 python bench.py --precision bf16 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 2>/dev/null | python -c "$show" | sed "s/^/parts=2 /"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bigb_$TAG -- python3 $R/bench.py --config big --precision bf16 --steps 5 --warmup 2 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 > $R/gpurun_out/prof_bigb_$TAG.log 2>&1
f=$(ls $R/gpurun_out/prof_bigb_$TAG/*/*kernel_stats.csv | head -1); head -24 $f | cut -c1-220
