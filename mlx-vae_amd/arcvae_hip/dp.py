"""Data-parallel training step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-process (no collective anywhere), so this layer is
defined by equivalence: N ranks x local minibatch == one process x the concatenated minibatch.
Batch rows are independent everywhere except the loss's batch statistics, so a
step needs exactly two exchanges:

  1. forward seam : all-reduce(SUM) of `stats` (2Z+4 floats: sum mu, sum exp(logvar), KL sums,
                    row count, CE row-sum) -> every rank evaluates the GLOBAL loss scalars and
                    the MI gate, and differentiates them w.r.t. its LOCAL rows (1/B_global scaling);
  2. backward     : all-reduce(SUM) of the flat gradient buffers in readiness order (north_star:
                    "overlapped with the backward LSTM sweep"): the decoder's share [dec.grad | CE sum]
                    (3.9 MB) is complete ~0.6 ms before the encoder's -- the decoder never reads z (Q2) -- and
                    is reduced on the SIDE stream right behind the decoder's backward, through a SECOND
                    communicator, i.e. beside the BPTT sweep; the encoder HEADS' share (1.58 MB: condition_fc, fc_mu,
                    fc_logvar_hidden, fc_logvar -- complete right behind the stats seam, before the BPTT has moved) follows
                    it on the same stream and communicator (round 4); what stays exposed on the main stream after the
                    sweep's last weight-gradient chunk is the LSTM layers + embedding (3.7 MB).  Messages are <= 3.9 MB:
                    on point-to-point xGMI RCCL picks a direct reduce-scatter/all-gather rather than a
                    per-link-bound ring.

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

    def forward_local(self) -> None: ...      # fills self.stats with this rank's partial sums; starts the decoder
    def backward_local(self) -> None: ...     # self.stats now holds GLOBAL sums; fills the encoder's gradients
    def early_buckets(self) -> List[torch.Tensor]: ...  # gradients that do not depend on the stats seam (decoder)
    def late_buckets(self) -> List[torch.Tensor]: ...   # gradients ready after backward_local
    def early_context(self) -> ContextManager: ...      # stream context the early reduces are issued from
    def recon_local(self) -> None: ...        # fills the CE sum (inside early_context)
    def early_done(self) -> None: ...         # optional: early buckets are GLOBAL (inside early_context)
    def seam_buckets(self) -> List[torch.Tensor]: ...   # optional: gradients complete right behind the stats seam (the encoder
    #                                             heads'): reduced on the EARLY communicator, behind the early buckets
    seam_issue_early: bool                    # optional: True = the ops order them on the device (a gate on the early stream), so
    #                                             they are ISSUED with the early buckets, ahead of backward_local's enqueue
    def apply_update(self) -> None: ...       # the CE sum is GLOBAL by now


class DataParallelStep:
    """Collectives are issued synchronously (async_op=False) under the stream context the ops choose: PyTorch then
    runs an RCCL collective ON that stream, so no internal communication stream sits blocked on an event beside the
    running LSTM chain (a blocked hardware queue costs ~1 us per dependent launch on MI355X, engine.Gates)."""

    def __init__(self, ops: StepOps, group: Optional[dist.ProcessGroup] = None,
                 early_group: Optional[dist.ProcessGroup] = None):
        """group: communicator of the stats seam and the late (encoder) bucket.  early_group: communicator of the
        early (decoder) bucket -- a SECOND one, because those reduces are issued from another stream and run
        concurrently with the first group's; created here (collectively: every rank constructs its driver) when not
        given.  Per communicator the issue order is the same on every rank: {stats, late buckets} and {CE sum, early
        buckets, seam buckets}."""
        self.ops = ops
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # exercise the collectives even at world size 1 (single-GPU rehearsal of the N-rank path)
        self.force = os.environ.get("ARCVAE_DP_FORCE_COLLECTIVES", "0") == "1" and dist.is_initialized()
        self.early_group = early_group
        if early_group is None and dist.is_initialized() and (self.world > 1 or self.force):
            # the same ranks as `group` (a subgroup's second communicator must not span the default group: its ranks
            # outside `group` would never make this collective call), same backend
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self.early_group = dist.new_group(ranks=ranks)
        warm = getattr(ops, "warm_up", None)
        if warm is not None and self.early_group is not None:
            warm(self.early_group)                      # set the second communicator up before any gate can wait behind it
        # Instrument for the first multi-GPU runs (bench.py): `timing = {}` switches on a pair of device events around every
        # collective, recorded on the stream it is issued on -- the time that stream spends between "everything before the
        # collective is done" and "the collective is done", i.e. transfer + waiting for the slowest peer.  On MAIN (stats seam,
        # late bucket) that time is on the step's critical chain: exposed communication.  Off by default (None): no events.
        self.timing = None

    def _all_reduce(self, t: Optional[torch.Tensor], group=None, tag: Optional[str] = None) -> None:
        if t is None or (self.world == 1 and not self.force):
            return
        if self.timing is not None and tag is not None and t.is_cuda:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group if group is not None else self.group)
            e1.record()
            self.timing.setdefault(tag, []).append((e0, e1, int(t.numel()) * t.element_size()))
            return
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group if group is not None else self.group)

    def comm_report(self) -> dict:
        """Median device time (us) of every collective of the instrumented steps (after a synchronize), by bucket, with the
        message size and the stream it was issued on; `exposed_comm_us` = the main-stream ones (stats seam + late bucket)."""
        rep, exposed = {}, 0.0
        for tag, evs in (self.timing or {}).items():
            us = sorted(1e3 * a.elapsed_time(b) for a, b, _ in evs)
            where = "side (beside the sweeps)" if tag in ("ce_sum", "dec_bucket", "heads_bucket") else "main (on the chain)"
            rep[tag] = {"us_median": us[len(us) // 2], "us_max": us[-1], "bytes": evs[0][2], "issued_on": where, "n": len(us)}
            if where.startswith("main"):
                exposed += us[len(us) // 2]
        rep["exposed_comm_us"] = exposed
        return rep

    def step(self) -> None:
        ops = self.ops
        seam = getattr(ops, "seam_buckets", None)
        seam_early = seam is not None and bool(getattr(ops, "seam_issue_early", False))
        ops.forward_local()
        with ops.early_context():                       # decoder-side results: independent of the stats seam (Q2), so
            ops.recon_local()                           # issued first -- they run beside the encoder's sweeps
            self._all_reduce(ops.recon_stat, self.early_group, "ce_sum")
            for g in ops.early_buckets():
                self._all_reduce(g, self.early_group, "dec_bucket")
            done = getattr(ops, "early_done", None)
            if done is not None:
                done()
            if seam_early:                              # the encoder heads' gradients: ready right behind the seam; the ops
                for g in seam():                        # put a device-side gate in front of the reduce (SURVEY 8e bucket order)
                    self._all_reduce(g, self.early_group, "heads_bucket")
        self._all_reduce(ops.stats, None, "stats")      # forward seam (critical path, 2Z+3 floats)
        ops.backward_local()
        if seam is not None and not seam_early:         # host-ordered forms (CPU doubles, event form, a driver's first step):
            with ops.early_context():                   # behind the backward's enqueue
                for g in seam():
                    self._all_reduce(g, self.early_group, "heads_bucket")
            join = getattr(ops, "seam_join", None)
            if join is not None:
                join()                                  # the late buckets' stream follows the seam buckets' reduce
        for g in ops.late_buckets():
            self._all_reduce(g, None, "enc_bucket")
        ops.apply_update()


class EngineOps:
    """StepOps over the HIP StepEngine (engine.py): the engine's phases with the collectives between.

    Gated form (default; engine.Gates -- side/aux wait in device-side gate kernels, no blocked hardware queue):
      main stream : signal -> [enc_fwd] -> (all-reduce stats) -> [loss + dcomb + sweep chunks + signals]
                    -> join aux, side -> (all-reduce enc.grad) -> [finalize + encoder adam]
      side stream : gate -> [dec_fwd + dec_bwd] -> [CE sum] -> (all-reduce dec.grad | CE sum, SECOND communicator)
                    -> [decoder adam] -> gate "heads' gradients formed" -> (all-reduce the encoder heads' gradients, same
                    communicator) ... gate -> [weight-gradient pieces of the sweep chunks]
      aux stream  : gate -> [weight gradients of chunk c] ...
      The gradient bucket is one contiguous buffer (the stores' `grad` tensors are re-pointed into it) reduced in
      two pieces: the decoder's 3.9 MB in SIDE's stream order right behind the decoder's backward -- no event, no
      gate: the collective is just the next operation of that stream, so nothing sits blocked beside the chain --
      which puts it beside the encoder's forward / BPTT sweeps (north_star: "overlapped with the backward LSTM
      sweep"); behind it, on the same stream and communicator, the encoder HEADS' 1.58 MB (the tail of enc.grad: complete
      once aux has run heads_wgrad right behind the seam -- gate word HG); the LSTM layers' + embedding's 3.7 MB on main
      after the join (SURVEY 8e's bucket order; ARCVAE_DP_HEADS_EARLY=0: the whole 5.3 MB after the join, as in round 3).
      Two communicators, each used from exactly one stream with the same issue order on every rank ({stats, enc LSTM
      bucket} on main's, {dec bucket, heads bucket} on side's).

    Event form (fallback when the gate probe fails, ARCVAE_GATES=0): as in round 1 --
      side: [dec_fwd] -> ev_chain -> [dec_bwd] -> ev_dec_bwd;  comm stream: CE sum and dec.grad reduced beside the
      BPTT sweep;  enc.grad reduced after it.
    """

    def __init__(self, engine, ws, lr: float, global_rows: int, use_graph: bool = True,
                 group: Optional[dist.ProcessGroup] = None):
        """group: the communicator of the DataParallelStep this ops object will be driven by (None = the default group).  The
        two collectives issued HERE -- the warm-up and the agreement on the gated / event form -- run on it: with a subgroup,
        ranks outside it never construct an EngineOps, and a collective on the default group would hang."""
        self.eng, self.ws, self.lr = engine, ws, float(lr)
        self.group = group
        self.global_rows = int(global_rows)
        Z = engine.d.Z
        self.stats = ws.stats[:2 * Z + 3]
        self.run = engine.runner(ws, lr, global_rows, capture=use_graph)
        if dist.is_initialized() and dist.get_backend() == "nccl":
            # warm the communicator up (channel / IPC set-up of the first collective can take seconds on 8 GPUs) before
            # any gate kernel can be left spinning behind it
            warm = torch.zeros(1024, dtype=torch.float32, device=engine.device)
            dist.all_reduce(warm, group=group)
            torch.cuda.synchronize()
        self.gated = engine.mode != "graph" and engine._gating_ok(torch.cuda.current_stream())
        self.gated = agree_all(self.gated, group, engine.device)
        # ARCVAE_DP_EARLY_REDUCE=0: round 1's ordering -- nothing is reduced beside the sweeps: the whole bucket [enc.grad |
        # dec.grad | CE sum] is ONE all-reduce on main after the join, and the decoder's Adam update moves into the finish
        # segment.  The default (early decoder reduce on side, second communicator) is UNMEASURED on more than one GPU: an RCCL
        # kernel that waits for a slower peer holds CU resources beside persistent sweeps that need a resident block on
        # every CU; this switch is the fallback if that turns out to cost more than the overlap gains.
        self.early = os.environ.get("ARCVAE_DP_EARLY_REDUCE", "1") != "0"
        # the encoder heads' gradients reduced early as well (gated + early form only): SURVEY 8e's second bucket
        # (only where a collective is ENQUEUED, i.e. RCCL: the reduce sits on side behind a device-side gate that opens once aux has
        # formed those gradients -- which the host enqueues AFTER this reduce.  A host-blocking backend -- gloo carrying device
        # tensors, the one-GPU rehearsal -- would wait in the reduce for a gate whose signal it has not enqueued yet: it keeps
        # round 3's order, the whole encoder bucket on main after the join)
        self.heads_early = (self.gated and self.early and os.environ.get("ARCVAE_DP_HEADS_EARLY", "1") != "0"
                            and dist.is_initialized() and dist.get_backend(group) == "nccl")
        self._steps = 0                  # steps this driver has enqueued (the first one records the segments: seam_issue_early)
        if self.gated:
            self._make_bucket()
            self.recon_stat = None
        else:
            self.recon_stat = ws.stats[2 * Z + 3:2 * Z + 4]
            self.comm = torch.cuda.Stream(device=engine.device)

    def warm_up(self, group) -> None:
        """First collective of a communicator (channel / IPC set-up: seconds on 8 GPUs) outside the step, on the stream
        that will use it -- and, where the heads' bucket is to be reduced behind a device-side gate, a dry run of exactly that
        construction (round 4): [gate on side] [all-reduce on side] released from aux.  If the backend ran the collective on a
        stream of its own that shares a hardware queue with aux, aux's release would sit behind the collective's wait for the gate
        -- the gate would expire.  Found here in a few milliseconds (bounded spin), on every rank, the early heads reduce is then
        switched off for this driver (round 3's order) instead of costing the step its stream order."""
        if dist.is_initialized() and dist.get_backend() == "nccl":
            with torch.cuda.stream(self.eng.side):
                t = torch.zeros(1024, dtype=torch.float32, device=self.eng.device)
                dist.all_reduce(t, group=group)
            torch.cuda.synchronize()
            if self.heads_early:
                ok = self._probe_reduce_behind_gate(group)
                self.heads_early = agree_all(ok, self.group, self.eng.device)
                if not self.heads_early:
                    print("[arcvae_hip] a collective behind a device-side gate on the side stream does not get released from aux "
                          "on this setup: the encoder heads' bucket is reduced with the LSTM bucket after the join")

    def _probe_reduce_behind_gate(self, group) -> bool:
        eng, g = self.eng, self.eng.gates
        torch.cuda.synchronize()
        g.mem[g.PROBE * 32] = 0
        before = g.errors()
        t = torch.zeros(1024, dtype=torch.float32, device=eng.device)
        with torch.cuda.stream(eng.side):
            call("arcvae_gate_wait", g.word(g.PROBE), C.c_void_p(0), 0, 1, 0, 100 * g.SHORT, g.word(g.ERR), stream_ptr())
            dist.all_reduce(t, group=group)
        with torch.cuda.stream(eng.aux):
            call("arcvae_gate_set", g.word(g.PROBE), 1, 0, stream_ptr())
        torch.cuda.synchronize()
        ok = g.errors() == before
        g.mem[g.ERR * 32] = before
        torch.cuda.synchronize()
        return ok

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
        nh = eng.enc.offsets["condition_fc.weight"]   # the heads' parameters are the tail of the encoder's flat buffer (store.py)
        self.bucket_enc = cur[:ne]               # complete after the join (last weight-gradient chunk)
        self.bucket_enc_lstm = cur[:nh]          # embedding + LSTM layers: what stays exposed behind the sweep
        self.bucket_heads = cur[nh:ne]           # condition_fc, fc_mu, fc_logvar_hidden, fc_logvar: complete right behind the seam
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
        # on SIDE, in stream order right behind the decoder's backward: CE row sums -> the bucket's last cell, so that
        # the [dec.grad | CE sum] half of the bucket is one message
        eng, ws = self.eng, self.ws
        Z = eng.d.Z
        base = C.c_void_p(self.bucket_recon.data_ptr() - 4 * (2 * Z + 3))
        self.run("dp_recon", lambda: call("arcvae_stats_set_recon", ptr(ws.rowloss), ws.B, base, Z, stream_ptr()),
                 torch.cuda.current_stream())

    def _dec_adam_gated(self) -> None:
        from .engine import adam_update
        eng, ws = self.eng, self.ws
        self.run("dp_dec_adam", lambda: adam_update(eng.dec, self.lr, guards=eng.guards(ws)), torch.cuda.current_stream())

    def _join_gated(self) -> None:
        self.run("dp_join", self.eng.gates.join, torch.cuda.current_stream())  # aux and side reported their last piece

    def _finish_gated(self) -> None:
        from .engine import adam_update
        eng, ws = self.eng, self.ws
        Z = eng.d.Z

        def fn():
            ws.stats[2 * Z + 3:2 * Z + 4].copy_(self.bucket_recon)   # GLOBAL CE sum (side reduced it long ago)
            ga, gb = eng.guards(ws)
            call("arcvae_loss_finalize", ptr(ws.stats), ptr(ws.scalars), Z, ws.T, ga, gb, stream_ptr())
            if not self.early:                                        # (early form: the decoder's update rode on side)
                adam_update(eng.dec, self.lr, guards=(ga, gb))
            adam_update(eng.enc, self.lr, guards=(ga, gb))
        self.run("dp_finish" if self.early else "dp_finish_late", fn, torch.cuda.current_stream())

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
            self.eng.enqueue_backward(self.ws, self._grun, gates=self.eng.gates)  # joined in late_buckets
            return
        self.eng.enqueue_backward(self.ws, self.run)

    @contextlib.contextmanager
    def early_context(self):
        if self.gated:
            with torch.cuda.stream(self.eng.side):   # the decoder's stream: its reduces are that stream's next operations
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
            return [self.bucket_dec] if self.early else []
        torch.cuda.current_stream().wait_event(self.eng.ev_dec_bwd)
        return [self.eng.dec.grad]

    def early_done(self) -> None:
        if self.gated and self.early:
            self._dec_adam_gated()                   # decoder gradients are GLOBAL: its Adam update rides on side too

    @property
    def seam_issue_early(self) -> bool:
        """DataParallelStep: issue the seam (heads) bucket with the early buckets, AHEAD of backward_local's enqueue -- a device-side
        gate orders it.  Not in a driver's FIRST step: that step records the segments, and a capture begins with a device-wide
        synchronize (torch.cuda.graph) -- with the gate already pending on side and its signal (aux's heads_wgrad, enqueued by
        backward_local) not enqueued yet, the host would sit there until the gate's bounded spin expired (seen: 15 s, ERR = 1).
        The first step therefore issues the bucket behind the backward (host order), like the CPU doubles do."""
        return self.heads_early and self._steps > 0

    def seam_buckets(self) -> List[torch.Tensor]:
        """The encoder heads' gradients, on SIDE behind the decoder's bucket: a device-side gate (word HG, raised by aux once
        heads_wgrad of THIS step has run: engine._encoder_backward_gated) orders the reduce behind them."""
        if not self.heads_early:
            return []
        from .engine import EncoderBackwardPlan
        g = self.eng.gates
        # side's ticket counter NS is advanced by the LAST gate of a step on side: with a single-chunk sweep that is the decoder
        # segment's own gate (in front of this one: offset 0), else the tail chunk's (behind it: offset 1).  Issued late (a
        # driver's first step) it follows the tail chunk's gate too: offset 0.
        early = self.seam_issue_early
        off = 1 if (early and len(EncoderBackwardPlan(self.eng.enc, self.ws, self.eng.d).chunks) >= 2) else 0
        # (a plain launch, NOT a recorded segment: one kernel, and a segment's first use synchronises its stream)
        g.wait(g.HG, g.NS, 1, off)
        return [self.bucket_heads]

    def seam_join(self) -> None:
        """First step only (seam bucket issued late, behind side's report on R): main follows side's reduce by an event."""
        if self.heads_early:
            torch.cuda.current_stream().wait_stream(self.eng.side)

    def late_buckets(self) -> List[torch.Tensor]:
        if self.gated:
            self._join_gated()       # (R: aux's and side's last pieces -- side's follows the heads' reduce in stream order)
            if not self.early:
                return [self.bucket]                      # (late form: side wrote its half before the join)
            return [self.bucket_enc_lstm] if self.heads_early else [self.bucket_enc]
        return [self.eng.enc.grad]

    def apply_update(self) -> None:
        self._steps += 1
        if self.gated:
            self._finish_gated()
            return
        torch.cuda.current_stream().wait_stream(self.comm)
        self.eng.enqueue_finish(self.ws, self.lr, True, self.run)


def agree_all(flag: bool, group: Optional[dist.ProcessGroup], device) -> bool:
    """True iff `flag` holds on EVERY rank of `group` (MIN all-reduce on that group -- never on the default group when a
    subgroup drives the step); the plain value without a process group or at world size 1."""
    if not dist.is_initialized() or dist.get_world_size(group) <= 1:
        return bool(flag)
    t = torch.tensor([1.0 if flag else 0.0], device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() > 0.5)


# ---- the trainer's two calls under data parallelism ------------------------------------------------------------------
def shard_bounds(n_rows: int, rank: int, world: int):
    """Rows [lo, hi) of a batch of n_rows that rank `rank` of `world` processes: contiguous, sizes differ by at most
    one (the ragged last batch of an epoch), every row owned exactly once."""
    return n_rows * rank // world, n_rows * (rank + 1) // world


def shard_plan(n_rows: int, rank: int, world: int):
    """(lo, hi, replicated): this rank's rows of a batch.  A batch with fewer rows than ranks (the ragged tail of an epoch
    can be that small) is REPLICATED: every rank works on all of its rows, without collectives, and the results are those of
    the single process by construction (training: rank 0's updated weights and optimizer state are then broadcast, so the
    ranks stay bit-identical whatever the reduction order inside their kernels)."""
    if n_rows < world:
        return 0, n_rows, True
    lo, hi = shard_bounds(n_rows, rank, world)
    return lo, hi, False


class EngineDataParallel:
    """N ranks x row shard == one process x global batch for the two calls the trainer makes (reference trainer.py:242-333
    is single-process; this layer is defined by that equivalence).  Every rank holds the same weights, the same dataset
    order (np.random.shuffle on the same seeded global stream) and the same teacher-forcing coins; a call receives the
    GLOBAL batch, takes this rank's rows (`shard_bounds`) and

      train_step   : DataParallelStep(EngineOps) -- stats seam + two gradient buckets over RCCL, both Adam updates; the
                     loss scalars every rank reads are the GLOBAL batch's (they are evaluated from the all-reduced sums);
      forward_loss : the loss forward on the local rows, ONE all-reduce of the 2Z+4 partial sums (latent sums, KL sums, row
                     count, CE sum), then the scalars of the global batch on every rank (validation / logging passes).

    One driver (captured segments, bucket views) per (local rows, T, global rows, lr); the second communicator is created
    once and shared."""

    def __init__(self, engine, group: Optional[dist.ProcessGroup] = None):
        if not dist.is_initialized():
            raise RuntimeError("EngineDataParallel needs an initialised torch.distributed process group")
        self.eng, self.group = engine, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self._drivers = {}
        self._early_group = None

    def bounds(self, n_rows: int):
        return shard_plan(n_rows, self.rank, self.world)

    def _broadcast_state(self) -> None:
        """Rank 0's parameters and Adam state to every rank (after a replicated step)."""
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        for st in (self.eng.enc, self.eng.dec):
            for buf in (st.flat, st.adam_m, st.adam_v):
                dist.broadcast(buf, src=src, group=self.group)

    def _load(self, ws, x, cond, eps, coins, lo, hi):
        self.eng.load_inputs(ws, x[lo:hi], cond[lo:hi], None if eps is None else eps[lo:hi], coins)

    def train_step(self, x, cond, eps, coins, lr: float, **hyper):
        """Loss, gradients and both Adam updates for the GLOBAL batch (x, cond, eps: all rows, identical on every rank;
        eps may also be this rank's own draw of the right shape: no loss value depends on it, Q2).  Returns the local
        workspace: ws.scalars are the global batch's loss scalars."""
        eng = self.eng
        n, T = int(x.shape[0]), int(x.shape[1])
        lo, hi, replicated = self.bounds(n)
        if replicated:       # fewer rows than ranks: the single-process step on every rank, then rank 0's state everywhere
            eng.train_step(x, cond, eps, coins, lr=lr, update=True, **hyper)
            self._broadcast_state()
            return eng.workspace(n, T, train=True)
        ws = eng.workspace(hi - lo, T, train=True)
        eng.set_hyper(ws, **hyper)
        self._load(ws, x, cond, eps, coins, lo, hi)
        key = (hi - lo, T, n, float(lr), float(eng.hyper_host["free_bits"]))
        drv = self._drivers.get(key)
        if drv is None:
            ops = EngineOps(eng, ws, lr, n, use_graph=(eng.mode != "eager"), group=self.group)
            drv = DataParallelStep(ops, self.group, early_group=self._early_group)
            self._early_group = drv.early_group
            self._drivers[key] = drv
        drv.step()
        # A rank whose sweep gave up (or whose gate expired) skipped its update and fed garbage into the buckets: every rank
        # must see that at THIS batch -- the step status travels as one more (tiny) all-reduce, so that all ranks raise
        # together instead of one leaving the others parked in the next collective.
        dist.all_reduce(ws.scalars[15:16], op=dist.ReduceOp.MAX, group=self.group)
        return ws

    def forward_loss(self, x, cond, eps, coins, **hyper):
        """complete_vae_loss forward of the GLOBAL batch: local rows -> partial sums -> one all-reduce -> scalars."""
        from .engine import latent_loss
        eng = self.eng
        n, T = int(x.shape[0]), int(x.shape[1])
        lo, hi, replicated = self.bounds(n)
        if replicated:
            eng.forward_loss(x, cond, eps, coins, **hyper)
            return eng.workspace(n, T, train=False)
        ws = eng.workspace(hi - lo, T, train=False)
        eng.set_hyper(ws, **hyper)
        self._load(ws, x, cond, eps, coins, lo, hi)
        main = torch.cuda.current_stream()
        eng.side.wait_stream(main)
        eng.enqueue_encoder_forward(ws, backward=False)      # this rank's partial latent sums in ws.stats
        eng.enqueue_decoder(ws, n, backward=False, wait_current=False)
        eng.enqueue_recon(ws)                                # + its CE row sums (waits for the decoder's walk)
        dist.all_reduce(ws.stats, op=dist.ReduceOp.SUM, group=self.group)
        latent_loss(ws, eng.d, float(eng.hyper_host["free_bits"]), False)
        ga, gb = eng.guards(ws)
        call("arcvae_loss_finalize", ptr(ws.stats), ptr(ws.scalars), eng.d.Z, ws.T, ga, gb, stream_ptr())
        main.wait_stream(eng.side)
        return ws
