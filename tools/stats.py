"""Print the top rows of a rocprofv3 kernel_stats.csv: python tools/stats.py TAG [N]"""
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f'gpurun_out/prof_{tag}/*/*kernel_stats.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:9.2f}us {r['Percentage']:>6s}%")
