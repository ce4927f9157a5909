"""Copy the summaries of a tools/measure_all.sh + tools/pmc.sh run from gpurun_out/ (scratch) into profiles/ (tracked):
python tools/collect_profiles.py TAG PMCTAG ROUND   e.g.  r2a r2 r02"""
import glob, json, os, shutil, sys
tag, pmctag, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
G, P = "gpurun_out", "profiles"
def cp(src, dst):
    if src and os.path.exists(src):
        shutil.copy(src, os.path.join(P, dst)); print("ok  ", dst)
    else:
        print("MISSING", src, "->", dst)
def first(pat):
    f = sorted(glob.glob(pat)); return f[0] if f else None
cp(f"{G}/bench_{tag}.json", f"{rnd}_bench.json")
cp(first(f"{G}/prof_{tag}/*/*_kernel_stats.csv"), f"{rnd}_bench_kernel_stats.csv")
cp(first(f"{G}/prof_{tag}_sweep/*/*_kernel_stats.csv"), f"{rnd}_bwd_sweep_isolated_kernel_stats.csv")
cp(f"{G}/trace_{tag}.txt", f"{rnd}_step_trace.txt")
cp(f"{G}/phase_{tag}.txt", f"{rnd}_phase_times.txt")
cp(f"{G}/bench_extra_{tag}.jsonl", f"{rnd}_bench_other_shapes.jsonl")
cp(f"{G}/sampler_{tag}.json", f"{rnd}_sampler.json")
cp(f"{G}/epoch_{tag}.txt", f"{rnd}_epoch_time.txt")
cp(f"{G}/pmc_summary_{pmctag}.json", f"{rnd}_pmc_summary.json")
# throughput mode (tools/bf16_measure.sh TAG)
cp(f"{G}/bf16_mode_{tag}.txt", f"{rnd}_bf16_mode.txt")
cp(f"{G}/bf16_bench_{tag}.jsonl", f"{rnd}_bf16_bench.jsonl")
cp(first(f"{G}/prof_bigb_{tag}/*/*_kernel_stats.csv"), f"{rnd}_bf16_big_kernel_stats.csv")
cp(first(f"{G}/prof_defb_{tag}/*/*_kernel_stats.csv"), f"{rnd}_bf16_default_kernel_stats.csv")
# round 3
cp(first(f"{G}/prof_{tag}_big/*/*_kernel_stats.csv"), f"{rnd}_big_fp32_kernel_stats.csv")
cp(first(f"{G}/prof_{tag}_sampler/*/*_kernel_stats.csv"), f"{rnd}_sampler_kernel_stats.csv")
cp(first(f"{G}/prof_{tag}_shard/*/*_kernel_stats.csv"), f"{rnd}_shard256_kernel_stats.csv")
cp(f"{G}/tick_stamps_{tag}.txt", f"{rnd}_tick_stamps.txt")
cp(first(f"{G}/prof_{tag}_strong/*/*_kernel_stats.csv"), f"{rnd}_strong2048_kernel_stats.csv")
cp(f"{G}/planes_{tag}.txt", f"{rnd}_planes_gemm.txt")
cp(f"{G}/trace_big_{tag}.txt", f"{rnd}_step_trace_configs2.txt")
cp(f"{G}/phase_big_{tag}.txt", f"{rnd}_phase_times_configs2.txt")
