#!/bin/bash
# usage (GPU box, repo root): tools/bf16_measure.sh TAG -- the throughput mode (precision bf16) next to the fp32 path: its tests, bench
# lines at the BASELINE shapes, which part buys what (ARCVAE_BF16_PARTS), the GEMM micro-benchmark, per-kernel profiles and the
# segment timeline of configs[2].  Everything lands in gpurun_out/bf16_*_TAG.* (tools/collect_profiles.py copies the summaries).
TAG=$1
R=$GRAFT_REPO_ROOT
O=gpurun_out/bf16_mode_$TAG.txt
python -m pytest tests/test_bf16_mode_gpu.py -x -q -s > gpurun_out/bf16_tests_$TAG.log 2>&1
{ echo "== tests/test_bf16_mode_gpu.py (tolerance: loss 2e-2, mu/logvar 5e-2 of max, gradients 8e-2 rel-L2 and cosine > 0.995 vs the fp64 oracle)"; grep -E "^\.?bf16 mode|passed|failed" gpurun_out/bf16_tests_$TAG.log; } > $O
show='import sys,json; d=json.loads(sys.stdin.read()); print("%-5s %-62s %8.3f ms/step %8d seq/s  loss %.6f" % (d["dtype"], d["config"]["workload"][:62], d["ms_per_step"], d["value"], d["elbo"]["total"]))'
rm -f gpurun_out/bf16_bench_$TAG.jsonl
echo "== bench.py --precision {fp32,bf16} (300 / 10-30 timed steps), same box" >> $O
for a in "--precision fp32 --steps 300" "--precision bf16 --steps 300" "--precision fp32 --batch-per-gpu 128" "--precision bf16 --batch-per-gpu 128" \
         "--precision fp32 --batch-per-gpu 256" "--precision bf16 --batch-per-gpu 256" \
         "--precision fp32 --batch-per-gpu 2048 --steps 30 --warmup 5" "--precision bf16 --batch-per-gpu 2048 --steps 30 --warmup 5" \
         "--config big --precision fp32 --steps 10 --warmup 3" "--config big --precision bf16 --steps 10 --warmup 3"; do
  timeout -k 10 200 python bench.py $a --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 2>/dev/null | tee -a gpurun_out/bf16_bench_$TAG.jsonl | python -c "$show" >> $O
done
echo "== which part buys what at configs[2] (ARCVAE_BF16_PARTS: 1 sweeps, 2 decoder GEMMs, 4 weight-gradient GEMMs; 7 = all)" >> $O
for m in 1 2 4 3 5; do
  ARCVAE_BF16_PARTS=$m timeout -k 10 200 python bench.py --config big --precision bf16 --steps 10 --warmup 3 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 2>/dev/null | python -c "$show" | sed "s/^/parts=$m /" >> $O
done
echo "== and at the default shape (1 = the persistent sweeps' 4x4x4 bf16 blocks, 2 = decoder GEMMs)" >> $O
for m in 1 2; do
  ARCVAE_BF16_PARTS=$m timeout -k 10 200 python bench.py --precision bf16 --steps 300 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 2>/dev/null | python -c "$show" | sed "s/^/parts=$m /" >> $O
done
echo "== tools/bench_bf16_gemm.py: bf16-operand tile GEMM vs the exact-f32 MFMA tile GEMM (err = max |c - c_fp64| / sum|a||b|)" >> $O
timeout -k 10 300 python tools/bench_bf16_gemm.py 2>/dev/null >> $O
echo "== tools/phase_times.py big bf16 / big (segment timeline of configs[2], us)" >> $O
timeout -k 10 250 python tools/phase_times.py big bf16 2>/dev/null >> $O
timeout -k 10 250 python tools/phase_times.py big 2>/dev/null >> $O
echo "== isolated BPTT sweep tick (bench.py --roofline-only), fp32 / bf16 blocks" >> $O
for p in fp32 bf16; do
  timeout -k 10 200 python bench.py --roofline-only --precision $p 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); r=d.get("roofline", d); print(sys.argv[1], "tick us %.3f" % r["us_per_launch"])' $p >> $O
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bigb_$TAG -- python3 $R/bench.py --config big --precision bf16 --steps 5 --warmup 2 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 > $R/gpurun_out/prof_bigb_$TAG.log 2>&1; echo "prof big bf16 rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_defb_$TAG -- python3 $R/bench.py --precision bf16 --steps 20 --warmup 5 --cpu-steps 0 --no-roofline --strong-global-batch 0 --bf16-steps 0 --configs2-steps 0 > $R/gpurun_out/prof_defb_$TAG.log 2>&1; echo "prof default bf16 rc=$?"
cat $R/$O
