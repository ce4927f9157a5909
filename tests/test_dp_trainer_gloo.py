"""world_size-2 (gloo, CPU) test of DATA-PARALLEL TRAINING through the trainer (reference trainer.py:242-333 is
single-process: N ranks x row shard must reproduce it).

Product code under test: ARCVAETrainerWithLoss's epoch flow (batch shuffle, coins, logging forwards, 20-batch TF-0 train
loss, validation, latent statistics, rank-0-only reporting), arcvae_hip.dp.DataParallelStep and arcvae_hip.dp.shard_bounds.
The three calls the trainer makes into the step engine are backed by the test oracle in fp64 (the HIP engine needs a GPU:
tests/test_dp_trainer_gpu.py runs the same comparison with the real kernels).  Two ranks must reproduce the epoch metrics
of the single-process reference flow (tests/ref_epoch.py) to 1e-10 -- epochs 0 and 1, ragged last batch included.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import arcvae_oracle as O
from helpers import TINY
from test_dp_gloo import OracleOps

DT = torch.float64
HP = dict(beta_start=0.0, beta_end=0.05, warmup=2, lambda_collapse=0.001, free_bits=1.0, lambda_mi=0.01)
N_TRAIN, N_VAL, T, BS, LR, EPOCHS = 29, 9, 10, 8, 2e-4, 2     # 29 rows / bs 8: three full batches and a ragged one of 5


def _data():
    rs = np.random.RandomState(3)
    mols = [list(rs.randint(3, TINY.V, size=rs.randint(4, T - 1))) + [2] for _ in range(N_TRAIN + N_VAL)]
    props = (rs.standard_normal((N_TRAIN + N_VAL, 1)) * 20 + 60).astype(np.float32)
    return mols, props


def _make_trainer(rank, world, tmp):
    """The product trainer with its three engine calls answered by the oracle (fp64), sharded over `world` ranks."""
    from arcvae_hip.dp import DataParallelStep, shard_plan
    from arcvae_hip import api
    from mlx_data.dataloader import MoleculeDataset
    from trainer import ARCVAETrainerWithLoss
    cfg = TINY

    class OracleTrainer(ARCVAETrainerWithLoss):
        def _make_engine(self, encoder, decoder):
            self.p = {k: v.astype(np.float64) for k, v in O.init_params(cfg, 1234).items()}
            self.m = {k: np.zeros_like(v) for k, v in self.p.items()}
            self.v = {k: np.zeros_like(v) for k, v in self.p.items()}
            return None

        def _rank_world(self):
            return rank, world

        def _shard(self, molecules, conditions):
            lo, hi, self._replicated = shard_plan(int(molecules.shape[0]), rank, world)   # (fewer rows than ranks: all rows, no collective)
            return molecules[lo:hi].numpy().astype(np.int64), conditions[lo:hi].numpy().astype(np.float64)

        def _train_step(self, molecules, conditions, tf, hyper):
            coins = api.draw_coins(int(molecules.shape[1]), tf)              # the global stream, as the product draws them
            n = int(molecules.shape[0])
            x, c = self._shard(molecules, conditions)
            hy = {k: hyper[k] for k in ("beta", "lambda_collapse", "free_bits", "lambda_mi", "target_mi")}
            ops = OracleOps(cfg, self.p, x, c, np.zeros((len(x), cfg.Z)), coins, int(molecules.shape[1]), self.learning_rate, n, hy)
            ops.m, ops.v = self.m, self.v                                   # Adam state carried across steps and epochs
            self._dp_step(ops)
            self.p = {k: t.detach().numpy() for k, t in ops.p.items()}
            return torch.tensor([ops.scalars["total"], 0.0], dtype=DT)

        def _dp_step(self, ops):
            step = DataParallelStep(ops, early_group=getattr(self, "_eg", None))
            self._eg = step.early_group                                     # one second communicator for the whole run
            if self._replicated:
                step.world = 1                                              # every rank holds all rows: no collective
            step.step()

        def _loss_dict(self, molecules, conditions, beta, tf):
            """The loss forward of the GLOBAL batch from the ranks' partial sums (the layout of csrc/latent.hip): one
            all-reduce, then the scalars -- what EngineDataParallel.forward_loss does on the GPU."""
            coins = api.draw_coins(int(molecules.shape[1]), tf)
            x, c = self._shard(molecules, conditions)
            hy = dict(beta=beta, lambda_collapse=self.lambda_collapse, free_bits=self.free_bits, lambda_mi=self.lambda_mi,
                      target_mi=4.85)
            ops = OracleOps(cfg, self.p, x, c, np.zeros((len(x), cfg.Z)), coins, int(molecules.shape[1]), 0.0,
                            int(molecules.shape[0]), hy)
            with torch.no_grad():
                st = ops._local_stats().detach().clone()
            if world > 1 and not self._replicated:
                dist.all_reduce(st)
            Z = cfg.Z
            Bg = st[2 * Z + 2]
            mm, mv = st[:Z] / Bg, st[Z:2 * Z] / Bg
            agg = -0.5 * (1.0 + torch.log(mv) - mm * mm - mv).sum()
            mi = torch.clamp(st[2 * Z] / Bg - agg, min=0.0)
            pen = torch.clamp(4.85 - mi, min=0.0)
            kl = st[2 * Z + 1] / Bg
            recon = st[2 * Z + 3] / (Bg * molecules.shape[1])
            collapse = self.lambda_collapse * pen
            total = recon + beta * kl + collapse + self.lambda_mi * pen
            return dict(total_loss=total, recon_loss=recon, kl_loss=kl, collapse_penalty=collapse,
                        prop_loss=torch.zeros((), dtype=DT))

        def _encode(self, molecules, conditions):
            pe = {k[len("encoder."):]: torch.tensor(v) for k, v in self.p.items() if k.startswith("encoder.")}
            with torch.no_grad():
                return O.encoder_forward(pe, molecules.to(torch.int64), conditions.to(DT), cfg.L)

        @staticmethod
        def _compute_mutual_information(mu, logvar):
            return float(O.mutual_information(mu, logvar, log_eps=1e-8))

    mols, props = _data()
    tr = MoleculeDataset(mols[:N_TRAIN], props[:N_TRAIN], max_length=T, device="cpu")
    va = MoleculeDataset(mols[N_TRAIN:], props[N_TRAIN:], max_length=T, properties_mean=tr.properties_mean,
                         properties_std=tr.properties_std, device="cpu")
    trainer = OracleTrainer(None, None, None, tr, learning_rate=LR, batch_size=BS, beta_start=HP["beta_start"],
                            beta_end=HP["beta_end"], beta_warmup_epochs=HP["warmup"], lambda_collapse=HP["lambda_collapse"],
                            free_bits=HP["free_bits"], lambda_mi=HP["lambda_mi"], checkpoint_dir=tmp, progress=False)
    return trainer, tr, va


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp, ret):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (os.path.join(root, "mlx-vae_amd"), os.path.join(root, "oracle"), os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    trainer, tr, va = _make_trainer(rank, world, tmp)
    assert (trainer.rank, trainer.world) == (rank, world)
    out = []
    for epoch in range(EPOCHS):
        np.random.seed(100 + epoch)                       # the one seeded stream every rank shares (train.py:75)
        out.append(trainer.train_epoch(epoch, 3, va))
    trainer.history["epoch"].append(0)
    trainer.save_history(tmp)                              # rank 0 alone writes
    if rank == 0:
        ret["metrics"] = out
        ret["params"] = trainer.p
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_training_reproduces_the_single_process_epochs(tmp_path):
    import ref_epoch as R
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), ret), nprocs=world, join=True)
    assert (tmp_path / "training_history.json").exists()
    # the single-process reference flow on the same data, seeds and hyper-parameters, fp64
    from mlx_data.dataloader import MoleculeDataset
    mols, props = _data()
    tr = MoleculeDataset(mols[:N_TRAIN], props[:N_TRAIN], max_length=T, device="cpu")
    va = MoleculeDataset(mols[N_TRAIN:], props[N_TRAIN:], max_length=T, properties_mean=tr.properties_mean,
                         properties_std=tr.properties_std, device="cpu")
    tr_np = (tr._tokens.numpy().astype(np.int64), tr._props.numpy().astype(np.float64))
    va_np = (va._tokens.numpy().astype(np.int64), va._props.numpy().astype(np.float64))
    p = {k: v.astype(np.float64) for k, v in O.init_params(TINY, 1234).items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(vv) for k, vv in p.items()}
    for epoch in range(EPOCHS):
        np.random.seed(100 + epoch)
        ref = R.reference_epoch(TINY, p, m, v, tr_np, va_np, BS, T, LR, epoch, 3, HP, dtype=DT)
        got = ret["metrics"][epoch]
        for k, r in ref.items():
            assert abs(got[k] - r) <= 1e-10 * max(1.0, abs(r)), (epoch, k, got[k], r)
    for k in p:
        assert np.abs(ret["params"][k] - p[k]).max() < 1e-10, k


def test_shard_bounds_cover_every_row_once():
    from arcvae_hip.dp import shard_bounds
    for n in (1, 5, 8, 63, 64, 2048):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
