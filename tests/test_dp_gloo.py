"""world_size-2 (gloo, CPU) test of the data-parallel step driver arcvae_hip.dp.DataParallelStep.

The driver's control flow (stats all-reduce seam, early/late gradient buckets, update) is the
product code; the per-rank math behind the `ops` protocol is supplied here by the test oracle,
split exactly the way the HIP kernels split it (per-rank partial `stats` -> global sums ->
gradients of the GLOBAL loss w.r.t. LOCAL rows).  N ranks x shard must equal 1 process x global
batch: loss scalars, summed gradients and post-Adam parameters.
"""
import contextlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import arcvae_oracle as O
from helpers import HYPER, TINY, make_case

DT = torch.float64


class OracleOps:
    """StepOps backed by the oracle (CPU).  stats layout = csrc/latent.hip."""

    def __init__(self, cfg, params, x, cond, eps, coins, T, lr, global_rows, hyper=None):
        self.cfg, self.T, self.lr, self.coins = cfg, T, lr, coins
        self.hy = dict(HYPER if hyper is None else hyper)
        self.global_rows = global_rows              # known a priori (the engine takes it as an argument too)
        self.order = []                             # issue order of the collectives' producers (same on every rank)
        self.p = {k: torch.tensor(v, dtype=DT, requires_grad=True) for k, v in params.items()}
        self.x = torch.as_tensor(x, dtype=torch.int64)
        self.cond = torch.as_tensor(cond, dtype=DT)
        self.eps = torch.as_tensor(eps, dtype=DT)
        Z = cfg.Z
        self._all = torch.zeros(2 * Z + 4, dtype=DT)
        self.stats = self._all[:2 * Z + 3]          # latent partial sums (forward seam)
        self.recon_stat = self._all[2 * Z + 3:]     # CE sum (reduced off the critical path)
        self.scalars = {}
        self.enc_names = [k for k in params if k.startswith("encoder.")]
        self.dec_names = [k for k in params if k.startswith("decoder.")]
        # the encoder HEADS' parameters (SURVEY 8e's second bucket: complete right behind the stats seam) / the LSTM stack's
        self.head_names = [k for k in self.enc_names if k.split(".")[1] in ("condition_fc", "fc_mu", "fc_logvar_hidden", "fc_logvar")]
        self.lstm_names = [k for k in self.enc_names if k not in self.head_names]
        self.seam_issue_early = False               # no device-side gate here: the heads bucket is reduced behind backward_local
        self.enc_grad = None
        self.dec_grad = None
        self.m = {k: np.zeros_like(v, dtype=np.float64) for k, v in params.items()}
        self.v = {k: np.zeros_like(v, dtype=np.float64) for k, v in params.items()}

    def _local_stats(self):
        cfg = self.cfg
        pe = {k[len("encoder."):]: v for k, v in self.p.items() if k.startswith("encoder.")}
        pd = {k[len("decoder."):]: v for k, v in self.p.items() if k.startswith("decoder.")}
        mu, logvar = O.encoder_forward(pe, self.x, self.cond, cfg.L)
        z = O.reparameterize(mu, logvar, self.eps)
        logits, _ = O.decoder_forward(pd, z, self.cond, cfg.L, self.x, self.coins)
        ce = O.reconstruction_loss(logits, self.x, reduction="sum")
        mc, lc = O.mlx_clip(mu, -3.0, 3.0), O.mlx_clip(logvar, -6.0, 3.0)
        var = torch.exp(lc)
        k = -0.5 * (1.0 + lc - mc * mc - var)
        kf = O.mlx_maximum(O.mlx_maximum(k, 0.0), self.hy["free_bits"] / cfg.Z)
        rows = torch.tensor([float(mu.shape[0])], dtype=DT)
        return torch.cat([mc.sum(0), var.sum(0), k.sum().reshape(1), kf.sum().reshape(1), rows, ce.reshape(1)])

    def forward_local(self):
        self.order.append("forward_local")
        self._attached = self._local_stats()
        self._all.copy_(self._attached.detach())
        self._local_detached = self._all.clone()
        # the decoder's gradients depend on nothing the stats seam delivers (Q2: it never reads z; the 1/(B_global*T)
        # scale is a constant): they are complete here, BEFORE the seam, and are reduced early (second communicator)
        Z = self.cfg.Z
        local_recon = self._attached[2 * Z + 3] / (self.global_rows * self.T)
        dec_params = [self.p[k] for k in self.dec_names]
        gd = torch.autograd.grad(local_recon, dec_params, allow_unused=True, retain_graph=True)
        self.dec_grad = torch.cat([(g if g is not None else torch.zeros_like(p)).reshape(-1)
                                   for g, p in zip(gd, dec_params)]).clone()

    def recon_local(self):
        self.order.append("recon_local")  # the oracle-side CE sum was already written by forward_local

    def early_done(self):
        self.order.append("early_done")

    def backward_local(self):
        """stats[:2Z+3] are global.  Gradients of the GLOBAL loss w.r.t. this rank's rows need the global latent
        sums and B_global, but only the LOCAL CE sum (d recon / d local params = d local_ce / (B_global*T)); the
        global CE sum is reduced later, off the critical path, exactly as on the GPU."""
        Z, T = self.cfg.Z, self.T
        lat = self.stats.detach() - self._local_detached[:2 * Z + 3] + self._attached[:2 * Z + 3]
        Bg = lat[2 * Z + 2].detach()
        mm, mv = lat[:Z] / Bg, lat[Z:2 * Z] / Bg
        agg = -0.5 * (1.0 + torch.log(mv) - mm * mm - mv).sum()
        mi = O.mlx_maximum(lat[2 * Z] / Bg - agg, 0.0)
        d = self.hy.get("target_mi", 4.85) - mi
        dpos = torch.where(torch.zeros_like(d) > d, torch.zeros_like(d), d)
        kl = lat[2 * Z + 1] / Bg
        self._latent_part = self.hy["beta"] * kl + self.hy["lambda_collapse"] * dpos + self.hy["lambda_mi"] * dpos
        self._Bg, self._mi, self._kl = float(Bg), float(mi), float(kl)
        self.order.append("backward_local")
        assert float(Bg) == float(self.global_rows)
        for v in self.p.values():
            v.grad = None
        self._latent_part.backward()               # the encoder only: the reconstruction term never reaches it (Q2)
        g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in self.p.items()}
        self.heads_grad = torch.cat([g[k].reshape(-1) for k in self.head_names]).clone()
        self.lstm_grad = torch.cat([g[k].reshape(-1) for k in self.lstm_names]).clone()
        assert all(float(g[k].abs().max()) == 0.0 for k in self.dec_names)

    def early_buckets(self):
        return [self.dec_grad]

    def seam_buckets(self):
        self.order.append("seam_buckets")
        return [self.heads_grad]

    def late_buckets(self):
        self.order.append("late_buckets")
        return [self.lstm_grad]

    def early_context(self):
        return contextlib.nullcontext()

    def apply_update(self):
        recon = float(self.recon_stat[0]) / (self._Bg * self.T)   # recon_stat is global by now
        self.scalars = dict(total=recon + float(self._latent_part), recon=recon, kl=self._kl, mi=self._mi)
        params = {k: v.detach().numpy().copy() for k, v in self.p.items()}
        grads = {}
        for names, flat in ((self.lstm_names, self.lstm_grad), (self.head_names, self.heads_grad), (self.dec_names, self.dec_grad)):
            o = 0
            for k in names:
                n = params[k].size
                grads[k] = flat[o:o + n].numpy().reshape(params[k].shape)
                o += n
        self.enc_grad = torch.cat([torch.as_tensor(grads[k]).reshape(-1) for k in self.enc_names])
        O.adam_update(params, grads, self.m, self.v, self.lr)
        self.p = {k: torch.tensor(v, dtype=DT, requires_grad=True) for k, v in params.items()}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _CountingGroupCalls:
    """Records (communicator, elements) of every all-reduce DataParallelStep issues: which bucket went through which group."""

    def __init__(self, step):
        self.step, self.calls = step, []
        inner = step._all_reduce

        def spy(t, group=None, tag=None):
            if t is not None:
                self.calls.append(("early" if group is step.early_group else "main", int(t.numel()), tag))
            inner(t, group, tag)
        step._all_reduce = spy


def _worker(rank, world, port, ret, sub_ranks=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from arcvae_hip.dp import DataParallelStep, agree_all
    group, nranks = None, world
    if sub_ranks is not None:
        # a 2-rank SUBGROUP of a 3-rank job drives the step: new_group is collective over the default group, so
        # the outsider creates both communicators too -- and then takes no part in any collective of the step
        group = dist.new_group(ranks=sub_ranks)
        early = dist.new_group(ranks=sub_ranks)
        nranks = len(sub_ranks)
        if rank not in sub_ranks:
            dist.barrier()
            dist.destroy_process_group()
            return
        # the agreement EngineOps makes on its form: on the subgroup (on the default group the outsider would be missing)
        assert agree_all(rank == sub_ranks[0], group, "cpu") is False and agree_all(True, group, "cpu") is True
    cfg, B, T = TINY, 6, 10
    params, x, cond, eps, coins = make_case(cfg, B, T, 0.6)
    me = rank if sub_ranks is None else sub_ranks.index(rank)
    sl = slice(me * B // nranks, (me + 1) * B // nranks)
    ops = OracleOps(cfg, params, x[sl], cond[sl], eps[sl], coins, T, 2e-4, B)
    step = DataParallelStep(ops, group=group, early_group=early if sub_ranks is not None else None)
    assert step.world == nranks
    # two communicators: the early (decoder) bucket does not share one with the stats seam / the late bucket
    assert step.early_group is not None and step.early_group is not step.group
    spy = _CountingGroupCalls(step)
    step.step()
    # the decoder's reduces are issued before the stats seam (they overlap the encoder's sweeps on the GPU); the encoder heads'
    # bucket follows the backward's enqueue (on the GPU: a device-side gate, issued with the early buckets), the LSTM bucket last
    assert ops.order == ["forward_local", "recon_local", "early_done", "backward_local", "seam_buckets", "late_buckets"], ops.order
    # which communicator carried what, in issue order: {CE sum, decoder, heads} early; {stats, LSTM + embedding} main
    assert spy.calls == [("early", 1, "ce_sum"), ("early", ops.dec_grad.numel(), "dec_bucket"), ("main", 2 * cfg.Z + 3, "stats"),
                         ("early", ops.heads_grad.numel(), "heads_bucket"), ("main", ops.lstm_grad.numel(), "enc_bucket")], spy.calls
    assert 0 < ops.heads_grad.numel() < ops.lstm_grad.numel()
    out = {k: v.detach().numpy() for k, v in ops.p.items()}
    if me == 0:
        ret["params"] = out
        ret["scalars"] = ops.scalars
        ret["enc_grad"] = ops.enc_grad.numpy()
        ret["dec_grad"] = ops.dec_grad.numpy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world,sub_ranks", [(2, None), (3, [0, 2])])
def test_two_ranks_equal_one_process_global_batch(world, sub_ranks):
    """(3, [0, 2]): the same step driven by a 2-rank SUBGROUP of a 3-rank job -- every collective of the driver, its second
    communicator and the form agreement stay inside the subgroup (rank 1 makes none of those calls; a collective on the
    default group would hang here)."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret, sub_ranks), nprocs=world, join=True)
    cfg, B, T = TINY, 6, 10
    params, x, cond, eps, coins = make_case(cfg, B, T, 0.6)
    p64 = {k: v.astype(np.float64) for k, v in params.items()}
    vals, grads = O.loss_and_grads(p64, cfg, x, cond, eps, coins, dtype=DT, **HYPER)
    sc = ret["scalars"]
    assert abs(sc["total"] - float(vals["total_loss"])) < 1e-10
    assert abs(sc["mi"] - float(vals["mutual_info"])) < 1e-10
    enc_ref = np.concatenate([grads[k].reshape(-1) for k in params if k.startswith("encoder.")])
    dec_ref = np.concatenate([grads[k].reshape(-1) for k in params if k.startswith("decoder.")])
    assert np.abs(ret["enc_grad"] - enc_ref).max() < 1e-12
    assert np.abs(ret["dec_grad"] - dec_ref).max() < 1e-12
    m = {k: np.zeros_like(v) for k, v in p64.items()}
    v = {k: np.zeros_like(vv) for k, vv in p64.items()}
    O.adam_update(p64, grads, m, v, 2e-4)
    for k in p64:
        assert np.abs(ret["params"][k] - p64[k]).max() < 1e-10, k
