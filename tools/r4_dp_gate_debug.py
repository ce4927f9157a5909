"""World-1 RCCL rehearsal of the data-parallel step, one step at a time, printing the gate words after each: which gate (if any)
expires with the early heads bucket (round 4).  usage: python tools/r4_dp_gate_debug.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("mlx-vae_amd", "tests", "oracle"): sys.path.insert(0, os.path.join(ROOT, p))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
os.environ["ARCVAE_DP_FORCE_COLLECTIVES"] = "1"
import torch, torch.distributed as dist
from helpers import DEFAULT, HYPER, build_engine, make_case
from arcvae_hip.dp import DataParallelStep, EngineOps
import arcvae_hip.engine as E
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
B, T = 64, 128
params, x, cond, eps, coins = make_case(DEFAULT, B, T, 0.9)
eng, enc, dec = build_engine(DEFAULT, params)
ws = eng.workspace(B, T)
eng.set_hyper(ws, **HYPER)
ops = EngineOps(eng, ws, 2e-4, B, use_graph=True)
print("gated", ops.gated, "heads_early", ops.heads_early, "streams main/side/aux:", torch.cuda.current_stream().cuda_stream, eng.side.cuda_stream, eng.aux.cuda_stream, flush=True)
step = DataParallelStep(ops)
names = "P Q NS NA ERR PROBE R NM D H HG".split()
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    eng.load_inputs(ws, x, cond, eps, coins)
    t0 = time.time()
    step.step()
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    g = eng.gates.mem.cpu().numpy()[::32]
    print(f"step {i}: enqueue {t1 - t0:.3f} s, drain {t2 - t1:.3f} s,", dict(zip(names, g.tolist())), "loss", float(ws.scalars[0]), flush=True)
dist.destroy_process_group()
