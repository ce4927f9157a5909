"""Data-parallel training step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-process (no collective anywhere), so this layer is
defined by equivalence: N ranks x local minibatch == one process x the concatenated minibatch.
Batch rows are independent everywhere except the loss's batch statistics, so a
step needs exactly two exchanges:

  1. forward seam : all-reduce(SUM) of `stats` (2Z+4 floats: sum mu, sum exp(logvar), KL sums,
                    row count, CE row-sum) -> every rank evaluates the GLOBAL loss scalars and
                    the MI gate, and differentiates them w.r.t. its LOCAL rows (1/B_global scaling);
  2. backward     : all-reduce(SUM) of the flat gradient buffers, one bucket per module.  The
                    decoder never reads z (Q2), so its whole forward+backward runs on the side
                    stream and its bucket (3.9 MB) is reduced while the encoder BPTT sweep is still
                    running on the main stream; the encoder bucket (5.3 MB) is the exposed tail.
                    Messages are <= 5.3 MB: on point-to-point xGMI RCCL picks a direct
                    reduce-scatter/all-gather rather than a per-link-bound ring.

Teacher-forcing coins, weights and Adam state must be identical on all ranks (same seed / same
reduced gradients); `DataParallelStep` never touches them.

The control flow is written against a small `ops` protocol so that the same code is exercised by
the CPU/gloo tests (tests/test_dp_gloo.py, where `ops` is backed by the test oracle) and by the
HIP engine (`EngineOps`) on the GPU.
"""
from __future__ import annotations

import contextlib
import os
from typing import ContextManager, List, Optional, Protocol

import torch
import torch.distributed as dist


class StepOps(Protocol):
    stats: torch.Tensor        # per-rank partial latent sums (2Z+3 floats), all-reduced before the backward
    recon_stat: torch.Tensor   # per-rank CE sum (1 float), all-reduced off the critical path

    def forward_local(self) -> None: ...      # fills self.stats with this rank's partial sums
    def backward_local(self) -> None: ...     # self.stats now holds GLOBAL sums; fills all gradients
    def early_buckets(self) -> List[torch.Tensor]: ...  # gradients ready before backward_local completes
    def late_buckets(self) -> List[torch.Tensor]: ...   # gradients ready after backward_local
    def early_context(self) -> ContextManager: ...      # stream context the early reduces are issued from
    def recon_local(self) -> None: ...        # fills self.recon_stat (inside early_context)
    def apply_update(self) -> None: ...       # self.recon_stat now holds the GLOBAL CE sum


class DataParallelStep:
    def __init__(self, ops: StepOps, group: Optional[dist.ProcessGroup] = None):
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # exercise the collectives even at world size 1 (single-GPU rehearsal of the N-rank path)
        self.force = os.environ.get("ARCVAE_DP_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized()

    def _all_reduce(self, t: torch.Tensor, async_op: bool = False):
        if self.world == 1 and not self.force:
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def step(self) -> None:
        ops = self.ops
        ops.forward_local()
        self._all_reduce(ops.stats)                     # forward seam (critical path, 2Z+3 floats)
        ops.backward_local()
        with ops.early_context():                       # decoder-side results: overlap the encoder BPTT
            ops.recon_local()
            works = [self._all_reduce(ops.recon_stat, async_op=True)]
            works += [self._all_reduce(g, async_op=True) for g in ops.early_buckets()]
        for g in ops.late_buckets():
            self._all_reduce(g)
        for w in works:
            if w is not None:
                w.wait()
        ops.apply_update()


class EngineOps:
    """StepOps over the HIP StepEngine (engine.py): the engine's phases with the collectives between.

    main stream : [enc_fwd] -> (all-reduce stats) -> [loss + dcomb + sweep chunks...] -> (all-reduce enc.grad)
                  -> [finalize + adam]
    side stream : [dec_fwd] -> ev_chain -> [dec_bwd] -> ev_dec_bwd
    aux stream  : [heads' and LSTM weight gradients, chunk by chunk behind the sweep]
    comm stream : wait ev_chain -> recon sum -> (all-reduce) ; wait ev_dec_bwd -> (all-reduce dec.grad)
    """

    def __init__(self, engine, ws, lr: float, global_rows: int, use_graph: bool = True):
        self.eng, self.ws, self.lr = engine, ws, float(lr)
        self.global_rows = int(global_rows)
        Z = engine.d.Z
        self.stats = ws.stats[:2 * Z + 3]
        self.recon_stat = ws.stats[2 * Z + 3:2 * Z + 4]
        self.run = engine.runner(ws, lr, global_rows, capture=use_graph)
        self.comm = torch.cuda.Stream(device=engine.device)

    def forward_local(self) -> None:
        # same host order as the single-process step: decoder segments first (they only follow the input copies),
        # then the encoder forward, so the decoder overlaps the forward sweep
        self.eng.side.wait_stream(torch.cuda.current_stream())
        self.eng.enqueue_decoder(self.ws, self.global_rows, self.run, wait_current=False)
        self.eng.enqueue_encoder_forward(self.ws, self.run)

    def backward_local(self) -> None:
        self.eng.enqueue_backward(self.ws, self.run)

    @contextlib.contextmanager
    def early_context(self):
        # decoder-side collectives are issued from the comm stream so they do not queue behind BPTT
        self.comm.wait_event(self.eng.ev_enc_fwd)  # the encoder forward zero-fills `stats` before the CE slot is set
        with torch.cuda.stream(self.comm):
            yield

    def recon_local(self) -> None:
        self.eng.enqueue_recon(self.ws, self.run)            # waits ev_chain on the comm stream

    def early_buckets(self) -> List[torch.Tensor]:
        torch.cuda.current_stream().wait_event(self.eng.ev_dec_bwd)
        return [self.eng.dec.grad]

    def late_buckets(self) -> List[torch.Tensor]:
        return [self.eng.enc.grad]

    def apply_update(self) -> None:
        torch.cuda.current_stream().wait_stream(self.comm)
        self.eng.enqueue_finish(self.ws, self.lr, True, self.run)
