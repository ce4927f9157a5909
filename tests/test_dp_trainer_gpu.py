"""Data-parallel TRAINING with the real HIP engine: two ranks on ONE MI355X (gloo carrying device tensors: RCCL refuses two
ranks per device) run the product trainer for two epochs -- every batch's rows sharded over the ranks, stats seam and
gradient buckets all-reduced (arcvae_hip.dp.EngineDataParallel), sharded validation / logging forwards, a ragged last
batch and a batch with fewer rows than ranks (replicated + state broadcast) -- and must reproduce the SINGLE-PROCESS
trainer's history (reference trainer.py:242-333 flow) to 1e-5, trained weights included.  Plus the CLI under a launcher's
environment (train.py --dist_backend gloo): rank 0 alone writes history and checkpoints.  N > 2 is unmeasured on hardware."""
import json
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import arcvae_oracle as O
from helpers import TINY, rel_err

pytestmark = pytest.mark.gpu
N_TRAIN, N_VAL, T, BS, LR, EPOCHS = 29, 9, 12, 8, 2e-4, 2     # 29 / 8: a ragged batch of 5; validation ends on ONE row
KW = dict(learning_rate=LR, batch_size=BS, beta_start=0.0, beta_end=0.05, beta_warmup_epochs=2, lambda_collapse=0.001,
          free_bits=1.0, lambda_mi=0.01, progress=False)


def _paths():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "mlx-vae_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _setup(tmp):
    """(trainer, vae, validation set) on cuda:0 from seeded data and the oracle's seeded initial weights."""
    from mlx_data.dataloader import MoleculeDataset
    from models.vae import ARCVAE
    from trainer import ARCVAETrainerWithLoss
    cfg = TINY
    rs = np.random.RandomState(3)
    mols = [list(rs.randint(3, cfg.V, size=rs.randint(4, T - 1))) + [2] for _ in range(N_TRAIN + N_VAL)]
    props = (rs.standard_normal((N_TRAIN + N_VAL, 1)) * 20 + 60).astype(np.float32)
    tr = MoleculeDataset(mols[:N_TRAIN], props[:N_TRAIN], max_length=T)
    va = MoleculeDataset(mols[N_TRAIN:], props[N_TRAIN:], max_length=T, properties_mean=tr.properties_mean,
                         properties_std=tr.properties_std)
    vae = ARCVAE(cfg.V, cfg.E, cfg.H, cfg.Z, cfg.C, cfg.L)
    params = O.init_params(cfg, 1234)
    vae.encoder.load_state_dict(params, prefix="encoder.")
    vae.decoder.load_state_dict(params, prefix="decoder.")
    return vae, tr, va, lambda: ARCVAETrainerWithLoss(vae.encoder, vae.decoder, None, tr, checkpoint_dir=tmp, **KW)


def _run_epochs(trainer, va):
    out = []
    for epoch in range(EPOCHS):
        np.random.seed(100 + epoch)              # the one seeded stream all ranks share (train.py:75)
        out.append(trainer.train_epoch(epoch, 3, va))
    torch.cuda.synchronize()
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, ret):
    _paths()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arcvae_hip import api
    vae, tr, va, make = _setup(tmp)
    dp = api.enable_data_parallel(vae.encoder, vae.decoder)
    assert (dp.rank, dp.world) == (rank, world)
    trainer = make()
    assert (trainer.rank, trainer.world) == (rank, world)
    metrics = _run_epochs(trainer, va)
    trainer.history["epoch"].append(0)
    trainer.save_history(tmp)
    trainer.save_checkpoint(0, is_best=True)
    enc, dec = vae.encoder.store.flat.cpu().numpy(), vae.decoder.store.flat.cpu().numpy()
    # every rank holds the same weights afterwards (also after the replicated batch: rank 0's state was broadcast)
    both = [None, None]
    dist.all_gather_object(both, float(np.abs(enc).sum() + np.abs(dec).sum()))
    assert both[0] == both[1], both
    if rank == 0:
        ret["metrics"], ret["enc"], ret["dec"] = metrics, enc, dec
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_rank_training_matches_the_single_process_trainer(tmp_path):
    _paths()
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path / "dp"), ret), nprocs=world, join=True)
    assert (tmp_path / "dp" / "training_history.json").exists() and (tmp_path / "dp" / "checkpoint_best.npz").exists()
    vae, tr, va, make = _setup(str(tmp_path / "single"))
    ref = _run_epochs(make(), va)
    for epoch in range(EPOCHS):
        for k, r in ref[epoch].items():
            g = ret["metrics"][epoch][k]
            assert abs(g - r) <= 1e-5 * max(1.0, abs(r)), (epoch, k, g, r)
    assert rel_err(ret["enc"], vae.encoder.store.flat.cpu().numpy()) < 1e-5
    assert rel_err(ret["dec"], vae.decoder.store.flat.cpu().numpy()) < 1e-5


def _cli_worker(rank, world, port, ck):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world))
    import train
    trainer = train.main(["--synthetic", "48", "--epochs", "2", "--batch_size", "16", "--hidden_dim", "64", "--embedding_dim", "16",
                          "--latent_dim", "8", "--checkpoint_dir", ck, "--checkpoint_freq", "1", "--no_progress", "--dist_backend", "gloo",
                          "--device", "cuda:0"])
    assert trainer.world == world and len(trainer.history["train_loss"]) == 2
    assert all(np.isfinite(v) for v in trainer.history["train_loss"] + trainer.history["val_loss"])


@pytest.mark.timeout(900)
def test_train_cli_under_a_launcher_environment(tmp_path):
    world = 2
    ck = str(tmp_path / "ck")
    mp.spawn(_cli_worker, args=(world, _free_port(), ck), nprocs=world, join=True)
    hist = json.load(open(os.path.join(ck, "training_history.json")))
    assert hist["epoch"] == [0, 1] and len(hist["train_loss"]) == 2
    assert os.path.exists(os.path.join(ck, "checkpoint_best.npz"))
