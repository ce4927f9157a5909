"""Per-kernel averages of the rocprofv3 --pmc passes of tools/pmc.sh: python tools/pmc_summary.py TAG > profiles/..json
bytes_per_launch = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes; the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md,
HBM section: it tallies 128-B requests at 64 B).  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's
1024 SIMDs) / (1024 * dispatch duration * 2.4 GHz), the dispatch duration being the End - Start stamps of the same
row (GRBM_GUI_ACTIVE is not usable as the denominator in counter mode: it reads ~19x the dispatch's cycles)."""
import csv, glob, json, sys, collections
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(f"gpurun_out/pmc_{tag}_*/"):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        per = collections.defaultdict(lambda: collections.defaultdict(float))  # (dispatch, kernel) -> counter -> value
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
            per[(r["Dispatch_Id"], name)][r["Counter_Name"]] += float(r["Counter_Value"])
            per[(r["Dispatch_Id"], name)]["_dur_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        for (_, name), cs in per.items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs["_dur_ns"] > 0:
                cs["_mfma_frac"] = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cs["_dur_ns"] * 2.4)
            for c, v in cs.items():
                acc[name][c].append(v)
out = {}
for name, cs in sorted(acc.items()):
    e = {"launches": min(len(v) for c, v in cs.items() if not c.startswith("_"))}
    avg = {c: sum(v) / len(v) for c, v in cs.items()}
    if "FETCH_SIZE" in avg: e["FETCH_SIZE_KB_avg"] = avg["FETCH_SIZE"]
    if "WRITE_SIZE" in avg: e["WRITE_SIZE_KB_avg"] = avg["WRITE_SIZE"]
    if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg:
        e["bytes_per_launch"] = 1024.0 * (2.0 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"])
        if name == "lstm_bwd_persist_rs_kernel":   # one launch per BPTT chunk: 130 ticks in CHUNKS launches at the default shape
            chunks = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
            e["bytes_per_tick"] = e["bytes_per_launch"] * chunks / 130.0
    if "_mfma_frac" in avg:
        e["mfma_busy_frac"] = avg["_mfma_frac"]
        e["mfma_busy_cycles_avg"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"]
    out[name] = e
json.dump(out, sys.stdout, indent=1)
