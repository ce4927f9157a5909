"""All kernels of ONE steady-state training step from a rocprofv3 kernel trace (tools/prof2.sh TAG ...), every stream:
python tools/r4_timeline.py TAG [step_index_from_end=3]  ->  start, end, duration (us), stream, kernel; step = enc_prologue .. next enc_prologue"""
import csv, glob, sys
tag = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = sorted(glob.glob(f'gpurun_out/prof_{tag}/**/*kernel_trace.csv', recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'enc_prologue' in r['Kernel_Name'] or 'transpose_tokens' in r['Kernel_Name']]
i0, i1 = starts[-back - 1], starts[-back]
t0 = int(rows[i0]['Start_Timestamp'])
def short(n):
    return n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:46]
print("step span %.1f us (prologue to prologue)" % ((int(rows[i1]['Start_Timestamp']) - t0) / 1e3))
sid = {}
for r in rows[max(0, i0 - 8):i1 + 2]:
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    st = sid.setdefault(r['Stream_Id'], len(sid))
    print(f"{s:9.1f} {e:9.1f} {e - s:8.1f}  s{st} q{r['Queue_Id']}  {'  ' * st}{short(r['Kernel_Name'])}")
