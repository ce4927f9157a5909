"""Main-queue timeline of one training step from a rocprofv3 kernel trace: python tools/mainline.py TAG [all]"""
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'transpose_tokens' in r['Kernel_Name']]
i0, i1 = starts[-3], starts[-2]
t0 = int(rows[i0]['Start_Timestamp'])
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:40]
print("step span", (int(rows[i1]['Start_Timestamp'])-t0)/1e3)
qmain = [r['Queue_Id'] for r in rows[i0:i1] if 'lstm_fwd' in r['Kernel_Name']][0]
show_all = len(sys.argv) > 2
prev=None
for r in rows[i0-6:i1+1]:
    k = short(r['Kernel_Name'])
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    if r['Queue_Id'] != qmain and not show_all: continue
    if 'lstm_' in k:
        if prev and prev[2]==k: prev=(prev[0],e,k,prev[3]+1); continue
        if prev: print(f"{prev[0]:8.1f} {prev[1]:8.1f} x{prev[3]} {prev[2]}")
        prev=(s,e,k,1); continue
    if prev: print(f"{prev[0]:8.1f} {prev[1]:8.1f} x{prev[3]} {prev[2]}"); prev=None
    print(f"{s:8.1f} {e:8.1f} {e-s:7.1f}  q={r['Queue_Id']} {k}")
if prev: print(f"{prev[0]:8.1f} {prev[1]:8.1f} x{prev[3]} {prev[2]}")
