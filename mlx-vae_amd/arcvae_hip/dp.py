"""Data-parallel training step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-process (no collective anywhere), so this layer is
defined by equivalence: N ranks x local minibatch == one process x the concatenated minibatch.
Batch rows are independent everywhere except the loss's batch statistics, so a
step needs exactly two exchanges:

  1. forward seam : all-reduce(SUM) of `stats` (2Z+4 floats: sum mu, sum exp(logvar), KL sums,
                    row count, CE row-sum) -> every rank evaluates the GLOBAL loss scalars and
                    the MI gate, and differentiates them w.r.t. its LOCAL rows (1/B_global scaling);
  2. backward     : all-reduce(SUM) of the flat gradient buffers -- one contiguous bucket [enc.grad |
                    dec.grad | CE sum] reduced in two pieces on the main stream behind the BPTT sweep (gated
                    form, see EngineOps), or one bucket per module with the decoder's reduced on a comm stream
                    beside the sweep (event form).  Messages are <= 5.3 MB: on point-to-point xGMI RCCL
                    picks a direct reduce-scatter/all-gather rather than a per-link-bound ring (SURVEY
                    section 5).

Teacher-forcing coins, weights and Adam state must be identical on all ranks (same seed / same
reduced gradients); `DataParallelStep` never touches them.

The control flow is written against a small `ops` protocol so that the same code is exercised by
the CPU/gloo tests (tests/test_dp_gloo.py, where `ops` is backed by the test oracle) and by the
HIP engine (`EngineOps`) on the GPU.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from typing import ContextManager, List, Optional, Protocol

import torch
import torch.distributed as dist

from ._lib import call, ptr, stream_ptr


class StepOps(Protocol):
    stats: torch.Tensor        # per-rank partial latent sums (2Z+3 floats), all-reduced before the backward
    recon_stat: Optional[torch.Tensor]  # per-rank CE sum (1 float); None when it travels inside a gradient bucket

    def forward_local(self) -> None: ...      # fills self.stats with this rank's partial sums
    def backward_local(self) -> None: ...     # self.stats now holds GLOBAL sums; fills all gradients
    def early_buckets(self) -> List[torch.Tensor]: ...  # gradients ready before backward_local completes
    def late_buckets(self) -> List[torch.Tensor]: ...   # gradients ready after backward_local
    def early_context(self) -> ContextManager: ...      # stream context the early reduces are issued from
    def recon_local(self) -> None: ...        # fills the CE sum (inside early_context)
    def apply_update(self) -> None: ...       # the CE sum is GLOBAL by now


class DataParallelStep:
    """Collectives are issued synchronously (async_op=False) under the stream context the ops choose: PyTorch then
    runs an RCCL collective ON that stream, so no internal communication stream sits blocked on an event beside the
    running LSTM chain (a blocked hardware queue costs ~1 us per dependent launch on MI355X, engine.Gates)."""

    def __init__(self, ops: StepOps, group: Optional[dist.ProcessGroup] = None):
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # exercise the collectives even at world size 1 (single-GPU rehearsal of the N-rank path)
        self.force = os.environ.get("ARCVAE_DP_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized()

    def _all_reduce(self, t: Optional[torch.Tensor]) -> None:
        if t is None or (self.world == 1 and not self.force):
            return
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def step(self) -> None:
        ops = self.ops
        ops.forward_local()
        self._all_reduce(ops.stats)                     # forward seam (critical path, 2Z+3 floats)
        ops.backward_local()
        with ops.early_context():                       # decoder-side results
            ops.recon_local()
            self._all_reduce(ops.recon_stat)
            for g in ops.early_buckets():
                self._all_reduce(g)
        for g in ops.late_buckets():
            self._all_reduce(g)
        ops.apply_update()


class EngineOps:
    """StepOps over the HIP StepEngine (engine.py): the engine's phases with the collectives between.

    Gated form (default; engine.Gates -- side/aux wait in device-side gate kernels, no blocked hardware queue):
      main stream : signal -> [enc_fwd] -> (all-reduce stats) -> [loss + dcomb + sweep chunks + signals]
                    -> gate on side's decoder -> [CE sum] -> (all-reduce dec.grad | CE sum: overlaps the last
                    weight-gradient chunk on aux / side) -> join aux, side -> (all-reduce enc.grad) -> [finalize + adam]
      side stream : gate -> [dec_fwd + dec_bwd] ... gate -> [tail chunk's token-table half]
      aux stream  : gate -> [weight gradients of chunk c] ...
      Three collectives per step, all on the main stream and one communicator (issue order = execution order on
      every rank).  The gradient bucket is one contiguous buffer (the stores' `grad` tensors are re-pointed into it)
      reduced in two pieces: the decoder's 3.9 MB right behind the sweep, beside the tail of the weight-gradient
      work, the encoder's 5.3 MB after the join.  The decoder's share is not overlapped with the BPTT sweep itself
      any more (that needed a second stream blocked on events), which buys a step without any cross-stream event
      wait while the chain runs.

    Event form (fallback when the gate probe fails, ARCVAE_GATES=0): as in round 1 --
      side: [dec_fwd] -> ev_chain -> [dec_bwd] -> ev_dec_bwd;  comm stream: CE sum and dec.grad reduced beside the
      BPTT sweep;  enc.grad reduced after it.
    """

    def __init__(self, engine, ws, lr: float, global_rows: int, use_graph: bool = True):
        self.eng, self.ws, self.lr = engine, ws, float(lr)
        self.global_rows = int(global_rows)
        Z = engine.d.Z
        self.stats = ws.stats[:2 * Z + 3]
        self.run = engine.runner(ws, lr, global_rows, capture=use_graph)
        if dist.is_initialized() and dist.get_backend() == "nccl":
            # warm the communicator up (channel / IPC set-up of the first collective can take seconds on 8 GPUs) before
            # any gate kernel can be left spinning behind it
            warm = torch.zeros(1024, dtype=torch.float32, device=engine.device)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        self.gated = engine.mode != "graph" and engine._gating_ok(torch.cuda.current_stream())
        if dist.is_initialized() and dist.get_world_size() > 1:
            # the two forms issue different collective sequences: every rank must take the same one
            flag = torch.tensor([1.0 if self.gated else 0.0], device=engine.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            self.gated = bool(flag.item() > 0.5)
        if self.gated:
            self._make_bucket()
            self.recon_stat = None
        else:
            self.recon_stat = ws.stats[2 * Z + 3:2 * Z + 4]
            self.comm = torch.cuda.Stream(device=engine.device)

    def _make_bucket(self) -> None:
        """One contiguous gradient bucket [enc.grad | dec.grad | CE sum]; recorded segments hold the old pointers,
        so every captured segment of the engine is dropped."""
        eng = self.eng
        ne, nd = eng.enc.numel_padded, eng.dec.numel_padded
        cur = getattr(eng, "_dp_bucket", None)
        if cur is None or cur.numel() != ne + nd + 64:
            cur = torch.zeros(ne + nd + 64, dtype=torch.float32, device=eng.device)
            cur[:ne].copy_(eng.enc.grad)
            cur[ne:ne + nd].copy_(eng.dec.grad)
            eng.enc.grad = cur[:ne]
            eng.dec.grad = cur[ne:ne + nd]
            eng._dp_bucket = cur
            eng._runners.clear()
            eng._graphs.clear()
            self.run = eng.runner(self.ws, self.lr, self.global_rows, capture=self.run.capture)
        self.bucket = cur
        self.bucket_enc = cur[:ne]               # complete after the join (last weight-gradient chunk)
        self.bucket_dec = cur[ne:ne + nd + 64]   # decoder gradients + CE sum: complete long before the sweep ends
        self.bucket_recon = cur[ne + nd:ne + nd + 1]

    # ---- gated form ---------------------------------------------------------------------------------
    def _grun(self, key, fn, stream) -> None:
        self.run(f"gated:{key}", fn, stream)

    def _forward_gated(self) -> None:
        from .engine import EncoderBackwardPlan
        eng, ws, g = self.eng, self.ws, self.eng.gates
        nc = len(EncoderBackwardPlan(eng.enc, ws, eng.d).chunks)
        # signal #1 is raised by the forward sweep when it starts; the decoder is enqueued behind it (engine.py)
        eng.enqueue_encoder_forward(ws, self._grun, start_signal=g.word(g.P))
        eng.enqueue_decoder(ws, self.global_rows, self._grun, wait_current=False, split_events=False,
                            gate=(g, nc < 2))

    def _recon_gated(self) -> None:
        # behind the BPTT sweep on main: once side reports its decoder segment done (gate on D: finished ~1 ms ago),
        # CE row sums -> the bucket's last cell; the [dec.grad | CE sum] half of the bucket is then reduced while
        # aux and side are still busy with the last chunk's weight gradients
        eng, ws, g = self.eng, self.ws, self.eng.gates
        Z = eng.d.Z
        base = C.c_void_p(self.bucket_recon.data_ptr() - 4 * (2 * Z + 3))

        def fn():
            g.wait(g.D, g.NM, 1, 1)    # ticket of main's current step; NM is advanced by the join below
            call("arcvae_stats_set_recon", ptr(ws.rowloss), ws.B, base, Z, stream_ptr())
        self.run("dp_recon", fn, torch.cuda.current_stream())

    def _join_gated(self) -> None:
        self.run("dp_join", self.eng.gates.join, torch.cuda.current_stream())  # aux and side reported their last piece

    def _finish_gated(self) -> None:
        from .engine import adam_update
        eng, ws = self.eng, self.ws
        Z = eng.d.Z

        def fn():
            ws.stats[2 * Z + 3:2 * Z + 4].copy_(self.bucket_recon)   # GLOBAL CE sum
            ga, gb = eng.guards(ws)
            call("arcvae_loss_finalize", ptr(ws.stats), ptr(ws.scalars), Z, ws.T, ga, gb, stream_ptr())
            adam_update(eng.dec, self.lr, guards=(ga, gb))
            adam_update(eng.enc, self.lr, guards=(ga, gb))
        self.run("dp_finish", fn, torch.cuda.current_stream())

    # ---- StepOps ----------------------------------------------------------------------------------------
    def forward_local(self) -> None:
        if self.gated:
            self._forward_gated()
            return
        # same host order as the single-process step: decoder segments first (they only follow the input copies),
        # then the encoder forward, so the decoder overlaps the forward sweep
        self.eng.side.wait_stream(torch.cuda.current_stream())
        self.eng.enqueue_decoder(self.ws, self.global_rows, self.run, wait_current=False)
        self.eng.enqueue_encoder_forward(self.ws, self.run)

    def backward_local(self) -> None:
        if self.gated:
            self.eng.enqueue_backward(self.ws, self._grun, gates=self.eng.gates)  # joined in _recon_gated
            return
        self.eng.enqueue_backward(self.ws, self.run)

    @contextlib.contextmanager
    def early_context(self):
        if self.gated:
            yield
            return
        # decoder-side collectives are issued from the comm stream so they do not queue behind BPTT
        self.comm.wait_event(self.eng.ev_enc_fwd)  # the encoder forward zero-fills `stats` before the CE slot is set
        with torch.cuda.stream(self.comm):
            yield

    def recon_local(self) -> None:
        if self.gated:
            self._recon_gated()
            return
        self.eng.enqueue_recon(self.ws, self.run)            # waits ev_chain on the comm stream

    def early_buckets(self) -> List[torch.Tensor]:
        if self.gated:
            return [self.bucket_dec]
        torch.cuda.current_stream().wait_event(self.eng.ev_dec_bwd)
        return [self.eng.dec.grad]

    def late_buckets(self) -> List[torch.Tensor]:
        if self.gated:
            self._join_gated()
            return [self.bucket_enc]
        return [self.eng.enc.grad]

    def apply_update(self) -> None:
        if self.gated:
            self._finish_gated()
            return
        torch.cuda.current_stream().wait_stream(self.comm)
        self.eng.enqueue_finish(self.ws, self.lr, True, self.run)
