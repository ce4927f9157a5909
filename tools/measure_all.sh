#!/bin/bash
# usage (GPU box, repo root): tools/measure_all.sh TAG -- every number DESIGN.md section 9 quotes, into gpurun_out/
TAG=$1
R=$GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
echo "bench done"
timeout -k 10 120 python tools/step_trace.py > gpurun_out/trace_$TAG.txt 2>&1; echo "trace rc=$?"
timeout -k 10 120 python tools/phase_times.py > gpurun_out/phase_$TAG.txt 2>&1; echo "phase rc=$?"
timeout -k 10 200 python tools/bench_sampler.py > gpurun_out/sampler_$TAG.json 2> gpurun_out/sampler_$TAG.err; echo "sampler rc=$?"
timeout -k 10 200 python tools/epoch_time.py > gpurun_out/epoch_$TAG.txt 2>&1; echo "epoch rc=$?"
rm -f gpurun_out/bench_extra_$TAG.jsonl
for a in "--batch-per-gpu 128" "--batch-per-gpu 256" "--batch-per-gpu 512 --steps 60 --warmup 10" "--batch-per-gpu 1024 --steps 40 --warmup 8" "--batch-per-gpu 2048 --steps 30 --warmup 5" "--config big --steps 30 --warmup 5" "--force-dp"; do
  timeout -k 10 200 python bench.py --cpu-steps 0 --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 --sampler-reps 0 $a 2>/dev/null >> gpurun_out/bench_extra_$TAG.jsonl; echo "extra [$a] rc=$?"
done
timeout -k 10 120 python tools/bench_planes.py > gpurun_out/planes_$TAG.txt 2>&1; echo "planes rc=$?"
timeout -k 10 200 python tools/step_trace.py --batch 512 --H 512 --L 4 --Z 256 --only-step --steps 8 > gpurun_out/trace_big_$TAG.txt 2>&1; echo "trace big rc=$?"
timeout -k 10 200 python tools/phase_times.py big > gpurun_out/phase_big_$TAG.txt 2>&1; echo "phase big rc=$?"
timeout -k 10 120 python tools/tick_stamps.py 64 > gpurun_out/tick_stamps_$TAG.txt 2>&1 < /dev/null; echo "stamps rc=$? (needs ab_libs/libarcvae_stamps.so: ARCVAE_HIP_LIB is set inside)"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 --sampler-reps 0 > $R/gpurun_out/prof_$TAG.log 2>&1; echo "prof rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_sweep -- python3 $R/bench.py --roofline-only > $R/gpurun_out/prof_${TAG}_sweep.log 2>&1; echo "sweep prof rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_big -- python3 $R/bench.py --config big --steps 6 --warmup 3 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 --shard-steps 0 --sampler-reps 0 > $R/gpurun_out/prof_${TAG}_big.log 2>&1; echo "big prof rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_sampler -- python3 $R/tools/bench_sampler.py > $R/gpurun_out/prof_${TAG}_sampler.log 2>&1; echo "sampler prof rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_shard -- python3 $R/bench.py --batch-per-gpu 256 --steps 20 --warmup 5 --cpu-steps 0 --no-roofline > $R/gpurun_out/prof_${TAG}_shard.log 2>&1; echo "shard prof rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_strong -- python3 $R/bench.py --batch-per-gpu 2048 --steps 6 --warmup 3 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 --shard-steps 0 --sampler-reps 0 > $R/gpurun_out/prof_${TAG}_strong.log 2>&1; echo "strong prof rc=$?"
