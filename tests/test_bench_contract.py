"""The bench line the driver parses (bench.py prints exactly this object on stdout): schema check on the committed
run profiles/r01_bench.json, so that a change to bench.py that drops or renames a field fails on CPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_every_contract_field():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench.json")))
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


def test_bench_source_names_the_contract_fields():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for k in ('"metric"', '"n_gpus"', '"ms_per_step"', '"higher_is_better"', '"scaling"', '"vs_baseline"', '"dtype"',
              '"roofline"', '"cpu_baseline"', '"workload"'):
        assert k in src, k
