"""The bench line the driver parses (bench.py prints exactly this object on stdout): schema check on the newest
committed run profiles/r*_bench.json, so that a change to bench.py that drops or renames a field fails on CPU; and the
self-launch path of `python bench.py --gpus N` (no GPU needed: --dry-launch prints the command of the child)."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _newest_bench():
    return json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))[-1]))


def test_gpus_n_without_a_launcher_starts_its_own_ranks():
    """`python bench.py --gpus 8` (the command shape the driver uses at N = 1) must not die on a missing WORLD_SIZE:
    it launches torch.distributed.run as a child.  --dry-launch shows that command without touching a GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "3",
                        "--dry-launch"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "8", "--steps", "7", "--warmup", "3"]        # same arguments, minus --dry-launch
    # one GPU: nothing to launch
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-launch"], capture_output=True, text=True,
                        env=env, timeout=120)
    assert r1.returncode == 0 and json.loads(r1.stdout.strip())["launch"] is None


def test_committed_bench_line_has_every_contract_field():
    d = _newest_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "sequences/s" and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1
    # the legs reported NEXT TO the fp32 headline (never instead of it): strong scaling, throughput mode, configs[2]
    s = d["strong"]
    assert s["global_batch"] == 2048 and s["scaling"] == "strong" and s["value"] > 0
    b = d["bf16_mode"]
    assert b["dtype"] == "bf16" and b["ms_per_step"] > 0 and "tolerance" in b and "NOT" in b["tolerance"]
    o = d["other_configs"]["configs[2]"]
    assert o["fp32"]["dtype"] == "f32" and o["bf16"]["dtype"] == "bf16"
    # (SURVEY 8(d)'s ratio for configs[2] is algorithmic f32 FLOP over the f32 MFMA peak; the three-piece kernels issue bf16 products,
    # so since round 4 it can pass 1 -- then the line must say so, and the fraction of the pipe in use must be a fraction)
    assert 0.0 < o["fp32"]["frac_of_f32_mfma_peak"] < 1.5 and 0.0 < o["bf16"]["frac_of_bf16_mfma_peak"] < 1.0
    assert 0.0 < o["fp32"]["executed_frac_of_bf16_pipe"] < 1.0
    if o["fp32"]["frac_of_f32_mfma_peak"] >= 1.0:
        assert "frac_note" in o["fp32"]


def test_bench_source_names_the_contract_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ('"metric"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"',
              '"roofline"', '"cpu_baseline"', '"workload"', '"strong"', '"n_ranks_seen"', '"step_tflops_executed"',
              '"tick_model"', '"limiter"', '"bf16_mode"', '"other_configs"'):
        assert k in src, k
