"""Host-side cost of one data-parallel step (RCCL group of one rank, forced collectives): the host must stay ahead
of the ~1.6 ms the GPU needs.  Each step is enqueued with the GPU idle (sync before), so the time is pure host work."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch, torch.distributed as dist
import bench as Bn
from arcvae_hip import _lib, engine as E
from arcvae_hip.dp import DataParallelStep, EngineOps
from arcvae_hip.store import ParamStore, decoder_shapes, encoder_shapes
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
os.environ["ARCVAE_DP_FORCE_COLLECTIVES"] = "1"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
V, EMB, H, Z, C, L, T, B = Bn.V, Bn.EMB, Bn.H, Bn.Z, Bn.C, Bn.L, Bn.T, 64
gen = torch.Generator().manual_seed(1234)
enc = ParamStore(encoder_shapes(V, EMB, H, Z, C, L), dev); dec = ParamStore(decoder_shapes(V, EMB, H, Z, C, L), dev)
enc.init_mlx_like(H, gen); dec.init_mlx_like(H, gen)
eng = E.StepEngine(enc, dec, E.ModelDims(V=V, E=EMB, H=H, Z=Z, C=C, L=L))
ws = eng.workspace(B, T, train=True); eng.set_hyper(ws, **Bn.HYPER)
rs = np.random.RandomState(67); x, cond = Bn.synth(rs, B)
ws.x.copy_(torch.tensor(x)); ws.cond.copy_(torch.tensor(cond))
dp = DataParallelStep(EngineOps(eng, ws, Bn.LR, B, use_graph=True))
for _ in range(5): dp.step()
torch.cuda.synchronize()
for name, fn in (("dp.step()", dp.step), ("eng.run_step()", lambda: eng.run_step(ws, Bn.LR, True))):
    ts = []
    for _ in range(30):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    print(f"{name}: host enqueue median {1e3 * sorted(ts)[len(ts)//2]:.3f} ms")
eng.check_gates()
dist.destroy_process_group()
