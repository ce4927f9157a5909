#!/bin/bash
# usage (GPU box, repo root): tools/r4_rccl_trace.sh -- kernel trace of the world-1 RCCL rehearsal of the data-parallel step
# (bench.py --force-dp): which HIP stream / hardware queue every RCCL kernel ran on, next to the engine's kernels of main / side /
# aux (profiles/r04_rccl_streams.txt).  rocprofv3 gets the program itself after `--` (no env / shell hop).
OUT=gpurun_out/r4_rccl
mkdir -p $OUT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29577 HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --force-dp --steps 12 --warmup 6 --cpu-steps 0 --no-roofline --strong-global-batch 0 > $GRAFT_REPO_ROOT/$OUT/bench.json 2> $GRAFT_REPO_ROOT/$OUT/bench.log
cd $GRAFT_REPO_ROOT
python3 - <<'EOF' | tee $OUT/rccl_streams.txt
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/r4_rccl/trace/**/*kernel_trace.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1])))
print("columns:", list(rows[0].keys()))
by = collections.defaultdict(lambda: collections.Counter())
key = "Stream_Id" if "Stream_Id" in rows[0] else ("Queue_Id" if "Queue_Id" in rows[0] else None)
for r in rows:
    name = r["Kernel_Name"]
    short = "RCCL:" + name[:60] if ("nccl" in name.lower() or "rccl" in name.lower()) else name.split("(")[0][-60:]
    by[(r.get("Stream_Id", "?"), r.get("Queue_Id", "?"))][short] += 1
for (st, q), c in sorted(by.items()):
    print(f"stream {st} queue {q}: {sum(c.values())} kernels")
    for k, n in c.most_common(12):
        print(f"    {n:6d}  {k}")
EOF
