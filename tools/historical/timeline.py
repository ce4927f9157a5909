"""Per-step timeline summary from a rocprofv3 kernel trace: python tools/timeline.py TAG"""
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a step starts at transpose_tokens_kernel
starts = [i for i, r in enumerate(rows) if 'transpose_tokens' in r['Kernel_Name']]
i0, i1 = starts[-3], starts[-2]
# include kernels of this step that started before transpose (side stream zero/fill) -- ignore
step = rows[i0:i1]
t0 = int(step[0]['Start_Timestamp'])
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:40]
print(f"step span: {(int(rows[i1]['Start_Timestamp']) - t0)/1e3:.1f} us, {len(step)} kernels")
# phases by marker kernels
marks = {}
for r in step:
    k = short(r['Kernel_Name'])
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    m = marks.setdefault(k, [s, e, 0, 0.0])
    m[0] = min(m[0], s); m[1] = max(m[1], e); m[2] += 1; m[3] += e - s
for k, (s, e, n, tot) in sorted(marks.items(), key=lambda kv: kv[1][0]):
    print(f"{k:42s} first {s:8.1f}  last-end {e:8.1f}  n={n:4d}  sum={tot:8.1f} us  stream={''}")
