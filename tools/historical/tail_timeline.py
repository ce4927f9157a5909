"""The exposed ends of a step from a rocprofv3 kernel trace: every kernel between the end of the BPTT sweep of one step and
the start of the next step's forward sweep, with queue ids and times relative to the sweep's end (us).
python tools/tail_timeline.py <kernel_trace.csv> [step index, default 10]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
fw = [i for i, r in enumerate(rows) if "lstm_fwd_persist" in r["Kernel_Name"]]
i0, i1 = fw[k], fw[k + 1]
t0, t1 = int(rows[i0]["Start_Timestamp"]), int(rows[i1]["Start_Timestamp"])
bw = [i for i, r in enumerate(rows) if "lstm_bwd_persist_rs" in r["Kernel_Name"] and t0 < int(r["Start_Timestamp"]) < t1]
tb = int(rows[bw[-1]]["End_Timestamp"])
short = lambda n: n.replace("(anonymous namespace)::", "").replace("void ", "")[:64]
print(f"step {k}: forward sweep {(int(rows[i0]['End_Timestamp']) - t0) / 1e3:.1f} us, BPTT ends {(tb - t0) / 1e3:.1f} us after the forward's start, "
      f"next forward starts {(t1 - tb) / 1e3:.1f} us after the BPTT's end (profiler attached: host-side gaps are inflated)")
print("   start      end      dur  queue  kernel   (us relative to the end of the BPTT sweep)")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if int(rows[bw[-1]]["Start_Timestamp"]) - 1000 <= s <= t1 + 1000:
        print(f"{(s - tb) / 1e3:8.1f} {(e - tb) / 1e3:8.1f} {(e - s) / 1e3:8.1f}  q{r['Queue_Id']}  {short(r['Kernel_Name'])}")
