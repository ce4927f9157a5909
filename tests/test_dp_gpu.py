"""Two ranks on ONE MI355X (gloo backend carrying device tensors; RCCL refuses two ranks on one device):
the real HIP engine behind arcvae_hip.dp.DataParallelStep must reproduce a single process on the global
batch -- loss scalars, summed gradients and post-Adam parameters -- including on the second step, which
replays the captured per-stream hipGraph segments around the collectives."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import HYPER, TINY, SMALL, build_engine, make_case, rel_err

pytestmark = pytest.mark.gpu
STEPS = 3
import arcvae_oracle as O  # noqa: E402
CONFIGS = {"tiny": TINY, "small": SMALL,
           # the default model's recurrent shape (H 256, L 2): the persistent forward sweep and the persistent
           # reduce-scatter BPTT sweep are the kernels under the collectives here
           "h256": O.Config(vocab_size=80, embedding_dim=32, hidden_dim=256, latent_dim=16, num_conditions=1, num_layers=2)}


def E_persistent(eng, ws) -> bool:
    from arcvae_hip import engine as E
    return E.persistent_forward_ok(ws, eng.d) and E.bptt_reduce_scatter_ok(ws, eng.d)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cfg_name, B, T, ret, lock=None):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "mlx-vae_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arcvae_hip.dp import DataParallelStep, EngineOps
    cfg = CONFIGS[cfg_name]
    params, x, cond, eps, coins = make_case(cfg, B, T, 0.6)
    eng, enc, dec = build_engine(cfg, params)
    lo, hi = rank * B // world, (rank + 1) * B // world  # uneven shards when B % world != 0
    ws = eng.workspace(hi - lo, T)
    eng.set_hyper(ws, **HYPER)
    ops = EngineOps(eng, ws, 2e-4, B, use_graph=True)
    if lock is not None:
        # Persistent sweeps need the whole GPU (one block per CU on all 256 CUs): two ranks sharing ONE device must
        # not run theirs at the same time.  Every phase that launches one is enqueued and drained under an
        # inter-process lock, so each rank's sweeps run alone on the device; the collectives (host-blocking gloo)
        # stay outside the lock.
        for name in ("forward_local", "backward_local"):
            def locked(fn=getattr(ops, name)):
                with lock:
                    fn()
                    torch.cuda.synchronize()
            setattr(ops, name, locked)
        assert E_persistent(eng, ws)
    step = DataParallelStep(ops)
    losses = []
    for _ in range(STEPS):
        eng.load_inputs(ws, x[lo:hi], cond[lo:hi], eps[lo:hi], coins)
        step.step()
        torch.cuda.synchronize()
        losses.append(ws.scalars.cpu().numpy()[:9].copy())
    if rank == 0:
        ret["losses"] = np.stack(losses)
        ret["enc"] = enc.flat.cpu().numpy()
        ret["dec"] = dec.flat.cpu().numpy()
        ret["enc_grad"] = enc.grad.cpu().numpy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("cfg_name,B,T", [("tiny", 8, 12), ("small", 21, 9), ("tiny", 6, 2)])  # T = 2: single-chunk sweep
def test_two_ranks_one_gpu_equal_single_process(cfg_name, B, T):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), cfg_name, B, T, ret), nprocs=world, join=True)
    _compare_with_single_process(cfg_name, B, T, ret)


def _compare_with_single_process(cfg_name, B, T, ret):
    cfg = CONFIGS[cfg_name]
    params, x, cond, eps, coins = make_case(cfg, B, T, 0.6)
    eng, enc, dec = build_engine(cfg, params)
    ref_losses = []
    for _ in range(STEPS):
        out = eng.train_step(x, cond, eps, coins, lr=2e-4, **HYPER)
        torch.cuda.synchronize()
        ref_losses.append(eng.workspace(B, T).scalars.cpu().numpy()[:9].copy())
    ref_losses = np.stack(ref_losses)
    got = ret["losses"]
    assert np.allclose(got, ref_losses, rtol=2e-5, atol=2e-6), (got, ref_losses)
    assert rel_err(ret["enc_grad"], enc.grad.cpu().numpy()) < 1e-4
    assert rel_err(ret["enc"], enc.flat.cpu().numpy()) < 1e-5
    assert rel_err(ret["dec"], dec.flat.cpu().numpy()) < 1e-5


@pytest.mark.timeout(600)
def test_two_ranks_late_reduce_form(monkeypatch):
    """ARCVAE_DP_EARLY_REDUCE=0 (the fallback ordering: nothing reduced beside the sweeps, ONE all-reduce of the whole
    gradient bucket on main after the join, the decoder's Adam in the finish segment) gives the same step."""
    monkeypatch.setenv("ARCVAE_DP_EARLY_REDUCE", "0")
    world, B, T = 2, 8, 12
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), "tiny", B, T, ret), nprocs=world, join=True)
    _compare_with_single_process("tiny", B, T, ret)


@pytest.mark.timeout(600)
def test_two_ranks_in_the_tile_regime(monkeypatch):
    """The MFMA-bound regime's machinery under the data-parallel step (what a rank of a 2-GPU run of configs[3] executes at 1024
    rows): three-piece tile sweeps, K-split BPTT tile, weight gradients from the operand planes with the bias rider, dense decoder
    layers on the tile kernels -- forced on at 2 x 32 rows, against one process on 64 rows."""
    for k, v in (("ARCVAE_STEP_TILE", "4"), ("ARCVAE_BWD_KSPLIT3", "2"), ("ARCVAE_DENSE_TILED", "2"), ("ARCVAE_PERSIST", "0")):
        monkeypatch.setenv(k, v)
    world, B, T = 2, 64, 7
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), "tiny", B, T, ret), nprocs=world, join=True)
    _compare_with_single_process("tiny", B, T, ret)


@pytest.mark.timeout(900)
def test_two_ranks_persistent_sweeps_under_the_collectives():
    """H = 256, L = 2 (the default model's recurrence): persistent forward sweep, persistent reduce-scatter BPTT, gates
    and the two-communicator collectives together at world size 2 -- 2 x 20 rows against one process on 40 rows.
    Each rank's sweeps run alone on the device (inter-process lock, see _worker)."""
    world, B, T = 2, 40, 12
    mgr = mp.Manager()
    ret = mgr.dict()
    lock = mgr.Lock()
    mp.spawn(_worker, args=(world, _free_port(), "h256", B, T, ret, lock), nprocs=world, join=True)
    _compare_with_single_process("h256", B, T, ret)
