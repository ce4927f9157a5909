#!/usr/bin/env python3
"""Headline benchmark: SELFIES sequences/sec of the AR-CVAE training step on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one minibatch of synthetic SELFIES-shaped input:
encoder LSTM sweep + heads, dense decoder, ELBO-style loss, hand-written backward, two Adam
updates (BASELINE.json configs[1]: default AR-CVAE V80 E128 H256 Z128 C1 L2, bs 64, T 128).
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU, each with its
own 64-row shard (weak scaling: the headline `value`), stats + gradient all-reduce over RCCL;
`python bench.py --gpus N` without a launcher starts its N ranks itself (torch.distributed.run as
a child process, before anything touches the GPU) and relays the rank-0 line.  After the weak leg
every run (N = 1 included) also times the STRONG-scaling leg of BASELINE.json configs[3] --
global batch 2048 split over the N ranks -- and reports it as the `strong` object.

One JSON line is printed by rank 0 with `roofline` (dominant kernel, live HIP-event timing) and
`cpu_baseline` (the oracle = CPU restatement of the reference step, timed on this box's host
cores; the MLX-CPU reference itself cannot run offline -- BASELINE.md section 2).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))

# default AR-CVAE (reference train.py:25-30) at BASELINE.json configs[1]
V, EMB, H, Z, C, L = 80, 128, 256, 128, 1, 2
T = 128
HYPER = dict(beta=0.0, lambda_collapse=0.001, lambda_mi=0.01, target_mi=4.85, free_bits=1.0)  # epoch-0 schedule
LR = 2e-4
TF_RATIO = 0.9
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-input MFMA = f32 vector peak
PEAK_HBM_GBS = 8000.0


def fwd_flops_per_seq(V, E, H, Z, C, L, T):
    """SURVEY.md section 8(d) algorithmic forward FLOPs per sequence."""
    enc_in = T * (2 * E * 4 * H + (L - 1) * 2 * H * 4 * H)
    enc_rec = (T - 1) * L * 2 * H * 4 * H
    heads = 2 * C * H + 2 * (2 * H) * Z + 2 * (2 * H) * (2 * H) + 2 * (2 * H) * Z
    dec = T * (2 * (E + C) * 4 * H + (L - 1) * 2 * H * 4 * H + 2 * H * V)
    return enc_in + enc_rec + heads + dec


def executed_flops_per_step(V, E, H, Z, C, L, T, B):
    """FLOPs the engine actually EXECUTES per training step (contractions only), next to the reference's algorithmic
    count: the embedding + layer-0 input projection is a [V,4H] token table (one skinny GEMM per module instead of
    B*T rows), the decoder is evaluated over B*V (row, token) pairs instead of B*T positions (all four gate columns
    are computed; the forget gate's are dead but ride in the same GEMM), and the backward mirrors both."""
    G, R, TB = 4 * H, B * V, T * B
    enc_f = 2 * V * G * E + (L - 1) * 2 * TB * G * H + L * 2 * (T - 1) * B * G * H            # table0, Wx_l, Wh_l
    heads_f = B * (2 * C * H + 2 * (2 * H) * Z + 2 * (2 * H) * (2 * H) + 2 * (2 * H) * Z)
    dec_f = 2 * V * G * E + (L - 1) * 2 * R * G * H + 2 * R * V * H                            # tableD, Wx_l, fc_out
    enc_b = (L * 2 * (T - 1) * B * G * H + (L - 1) * 2 * TB * G * H                            # dh = dG.Wh, dX = dG.Wx
             + L * 2 * (T - 1) * B * G * H + (L - 1) * 2 * TB * G * H                          # dWh, dWx_l
             + 2 * TB * V * G + 2 * (2 * V * G * E))                                           # one-hot GEMM, table finalize
    heads_b = 2 * heads_f
    dec_b = (2 * R * V * H * 2 + (L - 1) * 2 * R * G * H * 2 + 2 * (2 * V * G * E))            # dWout+dh, dWx_l+dh_l, table finalize
    return float(enc_f + heads_f + dec_f + enc_b + heads_b + dec_b)


_VENDOR_BF16 = {}


def vendor_bf16_gemm_tflops(torch, dev):
    """What the vendor library's bf16 GEMM (hipBLASLt through torch.matmul) sustains on THIS device on random operands, 8192^3:
    the practical ceiling of the bf16 matrix pipe under load (the chip lowers its clock in a dense MFMA loop on random data:
    1.27-1.37 PFLOP/s measured in round 4 against the nominal 2.5) -- the second denominator of the three-piece legs."""
    if "v" not in _VENDOR_BF16:
        _VENDOR_BF16["v"] = _measure_vendor_bf16(torch, dev)
    return _VENDOR_BF16["v"]


def _measure_vendor_bf16(torch, dev):
    """(a reference figure, not part of the product path: any failure of the library call reads as None -- the fields are then left out -- never as a failed bench)"""
    try:
        n = 8192
        a = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        b = torch.randn(n, n, device=dev, dtype=torch.bfloat16)
        for _ in range(3):
            c = a @ b
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            c = a @ b
        e1.record()
        torch.cuda.synchronize()
        v = 2.0 * n ** 3 * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12
        del a, b, c
        torch.cuda.empty_cache()
        return v
    except Exception as e:      # noqa: BLE001
        log(f"vendor bf16 GEMM reference not measured: {e!r}")
        return None


def sampler_flops(V, E, H, C, L, rows):
    """Contractions the greedy sampler executes for one batch of `rows` molecules, whatever max_length: the decoder is
    stateless (Q1/Q2), so ONE dense pass over the rows x V (row, token) pairs decides every step of every row."""
    G, R = 4 * H, rows * V
    return float(2 * V * G * E + (L - 1) * 2 * R * G * H + 2 * R * V * H)


def sampler_leg(torch, dev, reps):
    """BASELINE.json configs[4]: models/decoder_sampling.py generate_with_temperature, 10 000 molecules as 10 batches of bs
    1024 (the last one 784 rows), greedy, early stopping on (the API default), captured decode pass; max_length 80 (the API
    default) and 128 (BASELINE's sequence bound).  Random-init weights of the default architecture."""
    from models.vae import ARCVAE
    vae = ARCVAE(vocab_size=V, embedding_dim=EMB, hidden_dim=256, latent_dim=128, num_conditions=C, num_layers=2, device=dev)
    rs = np.random.RandomState(0)
    conds = [torch.tensor(rs.standard_normal((b, C)).astype(np.float32), device=dev) for b in [1024] * 9 + [784]]
    zs = [torch.zeros(c.shape[0], 128, device=dev) for c in conds]      # z is accepted and unused (Q2)
    flops = sum(sampler_flops(V, EMB, 256, C, 2, c.shape[0]) for c in conds)
    legs = {}
    for max_len in (80, 128):
        for c, z in ((conds[0], zs[0]), (conds[-1], zs[-1])):            # warm-up: capture both batch shapes
            vae.decoder_sampling.generate_with_temperature(z, c, max_length=max_len)
        times, ntok = [], 0
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ntok = 0
            for c, z in zip(conds, zs):
                ntok += vae.decoder_sampling.generate_with_temperature(z, c, max_length=max_len).numel()
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        dt = sorted(times)[len(times) // 2]
        legs[f"max_length_{max_len}"] = {
            "molecules_per_s": 10000 / dt, "tokens_per_s": ntok / dt, "ms_per_10k": 1e3 * dt, "reps": reps,
            "tflops_executed": flops / dt / 1e12, "executed_frac_of_f32_mfma_peak": flops / dt / 1e12 / PEAK_F32_MFMA_TFLOPS}
    return {"workload": "greedy sampling (models/decoder_sampling.py), 10k molecules = 9 x bs 1024 + 784, default AR-CVAE, "
                        "early stopping on, captured decode pass, host wall time incl. the per-batch early-stop read "
                        "(BASELINE.json configs[4])",
            "dtype": "f32", "unit": "molecules/s", **legs,
            "roofline": {"bound": "mfma", "limiter": "mfma", "kernel": "gemm_cell_zero_kernel + gemm_tile_kernel (dense decoder "
                         "pass over B*V rows) + dec_sample_chain_kernel (LDS table walk)", "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "achieved": legs["max_length_80"]["tflops_executed"],
                         "frac": legs["max_length_80"]["executed_frac_of_f32_mfma_peak"], "traffic": None,
                         "note": "whole-leg figure (host wall clock over 10 captured passes): executed contraction FLOPs / time; "
                                 "per-kernel times: profiles/r03_sampler_kernel_stats.csv"}}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch-per-gpu", type=int, default=None, help="default 64 (512 with --config big)")
    ap.add_argument("--config", choices=["default", "big"], default="default",
                    help="default = BASELINE.json configs[1]; big = configs[2] (H512 Z256 L4, bs 512: MFMA-bound regime)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32",
                    help="fp32 = the parity path (headline); bf16 = throughput mode (bf16 operands, f32 accumulate: "
                         "SURVEY 8(d) Config 2), reported with dtype bf16 and never as the headline")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--mode", choices=["auto", "graph", "eager", "segments"], default="auto",
                    help="launch mode: one multi-stream hipGraph, eager launches, or per-stream graph segments")
    ap.add_argument("--cpu-steps", type=int, default=6, help="oracle steps for cpu_baseline (0 = skip)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true",
                    help="run only the dominant-kernel probe (for `rocprofv3 --kernel-trace --stats -- python3 bench.py --roofline-only`)")
    ap.add_argument("--force-dp", action="store_true",
                    help="run the data-parallel driver (process group + collectives) even at world size 1")
    ap.add_argument("--strong-global-batch", type=int, default=2048,
                    help="global batch of the strong-scaling leg (BASELINE.json configs[3]); 0 = skip the leg")
    ap.add_argument("--strong-steps", type=int, default=30)
    ap.add_argument("--strong-warmup", type=int, default=6)
    ap.add_argument("--bf16-steps", type=int, default=100,
                    help="timed steps of the bf16 throughput-mode leg reported NEXT TO the fp32 headline (N = 1, default "
                         "config; 0 = skip)")
    ap.add_argument("--configs2-steps", type=int, default=8,
                    help="timed steps of the BASELINE.json configs[2] legs (H512 Z256 L4, bs 512: the MFMA-bound regime) in fp32 "
                         "and in bf16 throughput mode, reported under `other_configs` (N = 1, default config; 0 = skip)")
    ap.add_argument("--shard-steps", type=int, default=40,
                    help="timed steps of the strong leg's shard probe at N = 1: the step at global_batch / 8 rows (the per-GPU "
                         "shard of BASELINE.json configs[3]) on this one GPU (0 = skip)")
    ap.add_argument("--sampler-reps", type=int, default=3,
                    help="repetitions of the BASELINE.json configs[4] leg (10k molecules greedy sampling, bs 1024, max_length 80 "
                         "and 128; N = 1, default config; 0 = skip)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="print the torch.distributed.run command --gpus N would start, and exit (no GPU touched)")
    return ap.parse_args(argv)


def launch_ranks(args, argv) -> int:
    """`python bench.py --gpus N` (N > 1) without a launcher: start the N ranks as a CHILD process
    (python -m torch.distributed.run, one rank per GPU) before this process imports torch or touches the GPU,
    relay the rank-0 JSON line on stdout and return the child's exit code."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:   # a free rendezvous port on the loopback
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    child_argv = [a for a in argv if a != "--dry-launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + child_argv
    if args.dry_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    log(f"starting {args.gpus} ranks: {' '.join(cmd)}")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    out = proc.stdout.decode(errors="replace")
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    sys.stdout.write((lines[-1] + "\n") if lines else out)
    sys.stdout.flush()
    return proc.returncode


def synth(rs, B):
    lengths = rs.randint(20, T - 1, size=B)
    x = np.zeros((B, T), dtype=np.int32)
    for b in range(B):
        n = int(lengths[b])
        x[b, :n] = rs.randint(3, V, size=n)
        x[b, n] = 2
    cond = rs.standard_normal((B, C)).astype(np.float32)
    return x, cond


def host_cores() -> int:
    """Cores this process may actually use (affinity mask and cgroup quota), not the machine total."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))  # a one-GPU box's CPU share is 16


def log(msg: str) -> None:
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(sample_steps: int):
    """Time the oracle (CPU restatement of the reference step) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import arcvae_oracle as O
    cores = host_cores()
    torch.set_num_threads(cores)
    cfg = O.Config(V, EMB, H, Z, C, L)
    params = O.init_params(cfg, 1234)
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v = {k: np.zeros_like(vv) for k, vv in params.items()}
    x, cond = O.synthetic_batch(cfg, 64, T, 67)
    eps = np.random.RandomState(4321).standard_normal((64, Z)).astype(np.float32)
    rs = np.random.RandomState(68)
    hy = dict(beta=HYPER["beta"], lambda_collapse=HYPER["lambda_collapse"], lambda_mi=HYPER["lambda_mi"],
              free_bits=HYPER["free_bits"], target_mi=HYPER["target_mi"])
    O.train_step(params, m, v, cfg, x, cond, eps, O.draw_coins(rs, T, TF_RATIO), LR, **hy)  # warm-up
    t0 = time.perf_counter()
    for _ in range(sample_steps):
        O.train_step(params, m, v, cfg, x, cond, eps, O.draw_coins(rs, T, TF_RATIO), LR, **hy)
    dt = time.perf_counter() - t0
    return dict(value=64 * sample_steps / dt, unit="sequences/s", cores=cores, kind="port",
                sample=f"{sample_steps} training steps of the default config (bs 64, T 128) with the torch-CPU "
                       f"oracle, {dt:.1f} s; MLX-CPU itself is not installable offline")


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if (args.gpus > 1 and "WORLD_SIZE" not in os.environ) or args.dry_launch:
        if args.gpus <= 1:
            print(json.dumps({"launch": None}))
            return 0
        return launch_ranks(args, argv)

    # Everything except the result line goes to stderr: RCCL prints its version banner on stdout at init.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def emit(line: str) -> None:
        sys.stdout.flush()
        os.write(real_stdout, (line + "\n").encode())

    import torch
    import torch.distributed as dist
    from arcvae_hip import _lib
    from arcvae_hip import engine as E
    from arcvae_hip.dp import DataParallelStep, EngineOps
    from arcvae_hip.store import ParamStore, decoder_shapes, encoder_shapes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    _lib.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # (rehearsal on a one-GPU box: ARCVAE_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 -- RCCL refuses two ranks per device, so
    # it goes with ARCVAE_BENCH_BACKEND=gloo and ARCVAE_PERSIST=0; numbers from such a run mean nothing, the code path does)
    if os.environ.get("ARCVAE_BENCH_ONE_DEVICE", "0") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if os.environ.get("ARCVAE_BENCH_MAIN_PRIO", "0") == "1":   # experiment: the critical chain on a high-priority stream
        torch.cuda.set_stream(torch.cuda.Stream(priority=-1))
    dev = torch.device("cuda", local_rank)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.force_dp:
            os.environ["ARCVAE_DP_FORCE_COLLECTIVES"] = "1"
        backend = os.environ.get("ARCVAE_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": dev} if backend == "nccl" else {}))

    global H, Z, L
    if args.config == "big":
        H, Z, L = 512, 256, 4
    B = args.batch_per_gpu or (512 if args.config == "big" else 64)
    wl_name = ("default AR-CVAE V80 E128 H256 Z128 C1 L2" if args.config == "default"
               else "big AR-CVAE V80 E128 H512 Z256 C1 L4")
    wl_ref = "BASELINE.json configs[1]" if (args.config == "default" and B == 64) else (
        "BASELINE.json configs[2]" if (args.config == "big" and B == 512) else "non-headline batch size")
    dims = E.ModelDims(V=V, E=EMB, H=H, Z=Z, C=C, L=L)
    enc = ParamStore(encoder_shapes(V, EMB, H, Z, C, L), dev)
    dec = ParamStore(decoder_shapes(V, EMB, H, Z, C, L), dev)
    mode = args.mode
    if mode == "auto":
        mode = "eager" if args.no_graph else "segments"

    def make_engine(rows, global_rows, trace=False, precision=None):
        """(Re)initialise the weights and build engine, workspace and DP driver under the current ARCVAE_* knobs."""
        gen = torch.Generator().manual_seed(1234)  # identical initial weights on every rank
        enc.init_mlx_like(H, gen)
        dec.init_mlx_like(H, gen)
        enc.p("fc_logvar.bias").fill_(0.35)
        for st in (enc, dec):
            st.grad.zero_()
            st.adam_m.zero_()
            st.adam_v.zero_()
        eng_ = E.StepEngine(enc, dec, dims, precision=precision or args.precision)
        eng_.mode = mode
        ws_ = eng_.workspace(rows, T, train=True)
        if trace:
            # device-side stamps of the sweep ticks / launches (two 8-byte stores each): the in-step cadence of the
            # dominant kernel is reported next to its isolated timing (before the first step: the pointers are baked
            # into the captured graphs)
            eng_.enable_trace(ws_)
        eng_.set_hyper(ws_, **HYPER)
        dp_ = None
        if use_dp:
            dp_ = DataParallelStep(EngineOps(eng_, ws_, LR, global_rows, use_graph=(mode != "eager")))
        return eng_, ws_, dp_

    def make_inputs(rows, total, seed):
        """device-resident synthetic batches (per-rank shard) and per-step coins (same on every rank, Q5)"""
        rs = np.random.RandomState(seed + rank)
        nbuf = 8
        xs, cs, es = [], [], []
        for _ in range(nbuf):
            x, cond = synth(rs, rows)
            xs.append(torch.tensor(x, device=dev))
            cs.append(torch.tensor(cond, device=dev))
            es.append(torch.tensor(rs.standard_normal((rows, Z)).astype(np.float32), device=dev))
        crs = np.random.RandomState(4242)
        coins = torch.tensor((crs.rand(total, T) < TF_RATIO).astype(np.uint8), device=dev)
        return xs, cs, es, coins

    def stepper(eng_, ws_, dp_, inputs):
        xs, cs, es, coins = inputs

        def one_step(i):
            k = i % len(xs)
            eng_.load_inputs(ws_, xs[k], cs[k], es[k], coins[i])   # device-resident batch -> the step's static buffers (one launch)
            if dp_ is None:
                eng_.run_step(ws_, LR, update=True)
            else:
                dp_.step()
        return one_step

    def all_ranks_max(flag: int) -> int:
        if world > 1:
            bt = torch.tensor([flag], device=dev, dtype=torch.int32)
            dist.all_reduce(bt, op=dist.ReduceOp.MAX)
            return int(bt.item())
        return flag

    def timed(one_step, first, count):
        """EXACTLY `count` steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        host_enq = 0.0
        marks = []                                   # ARCVAE_BENCH_INTERVALS=n: an event every n steps (diagnostics)
        every = int(os.environ.get("ARCVAE_BENCH_INTERVALS", "0"))
        for i in range(first, first + count):
            if every and (i - first) % every == 0:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append(ev)
            h0 = time.perf_counter()
            one_step(i)
            host_enq += time.perf_counter() - h0
        torch.cuda.synchronize()
        if len(marks) > 1:
            log("ms/step per interval: " + " ".join(f"{marks[k].elapsed_time(marks[k + 1]) / every:.3f}"
                                                    for k in range(len(marks) - 1)))
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, host_enq

    def dp_form(dp_) -> str:
        """Which data-parallel form the step ran in (arcvae_hip/dp.py), for the record next to the number."""
        if dp_ is None:
            return "single process (no collectives)"
        ops = dp_.ops
        if not getattr(ops, "gated", False):
            return "event form: decoder bucket on a comm stream beside the sweep, encoder bucket after it"
        if getattr(ops, "early", True) and getattr(ops, "heads_early", False):
            return ("gated form: decoder bucket, then the encoder heads' bucket (behind a device-side gate) reduced early on side through "
                    "a second communicator; the LSTM layers + embedding bucket on main after the join")
        return ("gated form: decoder bucket reduced early on side through a second communicator, encoder bucket on main after the join"
                if getattr(ops, "early", True) else
                "gated form, ARCVAE_DP_EARLY_REDUCE=0: the whole gradient bucket reduced on main after the join")

    def comm_probe(dp_, one_step_, n_inputs, n=12):
        """Device-side stamps around every collective over n extra (untimed) steps: medians by bucket and `exposed_comm_us` (the
        main-stream collectives: on the step's chain).  Unmeasured on more than one GPU until the driver's multi-GPU run; at
        world size 1 (--force-dp) the figures are RCCL's launch + single-rank kernel cost."""
        if dp_ is None:
            return None
        torch.cuda.synchronize()
        dp_.timing = {}
        for i in range(n):
            one_step_(i % n_inputs)
        torch.cuda.synchronize()
        rep = dp_.comm_report()
        dp_.timing = None
        rep["note"] = ("HIP events around each collective on the stream it is issued on, median over %d untimed steps after the timed "
                       "region; main-stream time is exposed on the chain, side-stream time runs beside the sweeps" % n)
        return rep

    def healthy(eng_) -> str:
        """'' when every device-side gate opened in order and no persistent sweep gave up, on EVERY rank."""
        bad, why = 0, ""
        try:
            eng_.check_gates()
        except _lib.ArcvaeHipError as e:
            bad, why = 1, str(e)
            log(f"rank {rank}: {why}")
        return (why or "another rank lost its stream order") if all_ranks_max(bad) else ""

    eng, ws, dp = make_engine(B, B * world, trace=True)
    total = args.warmup + args.steps
    one_step = stepper(eng, ws, dp, make_inputs(B, total, 67))

    if args.roofline_only:
        one_step(0)
        torch.cuda.synchronize()
        emit(json.dumps({"roofline": roofline_probe(eng, ws, torch)}))
        return 0
    # The host enqueues a step in less than half its device time, but only a few steps ahead: a generational GC pass over
    # the heap torch leaves behind (~10 ms) lands on the device timeline as a stall.  The loop below allocates nothing
    # that needs cycle collection.  (ARCVAE_BENCH_GC=1 leaves the collector on.)
    if os.environ.get("ARCVAE_BENCH_GC", "0") == "0":
        import gc
        gc.collect()
        gc.freeze()
        gc.disable()
    log(f"rank {rank}/{world}: warm-up ({args.warmup} steps, mode={mode})")
    fallback = None
    for attempt in range(3):
        for i in range(args.warmup):
            one_step(i)
        torch.cuda.synchronize()
        # Health check before anything is timed: a device-side gate or a persistent sweep that gave up waiting (a
        # device whose queues / CUs are not laid out as probed) must not produce a number.  All ranks agree, then
        # everybody drops one level: gates -> event waits, persistent sweeps -> per-step launches.
        bad = 1 if healthy(eng) else 0
        if attempt < int(os.environ.get("ARCVAE_BENCH_TEST_FALLBACK", "0")):
            bad = 1                                  # rehearsal of the fallback path (tests / tools only)
        if not bad:
            break
        if attempt == 0:
            os.environ["ARCVAE_GATES"] = "0"
            fallback = "event waits instead of device-side gates"
        elif attempt == 1:
            os.environ["ARCVAE_PERSIST"] = "0"
            fallback = "event waits, per-step launches instead of persistent sweeps"
        else:
            raise SystemExit("bench.py: the step does not run cleanly on this device even without gates and "
                             "persistent sweeps")
        log(f"rank {rank}: falling back to {fallback}")
        eng, ws, dp = make_engine(B, B * world, trace=True)
        one_step = stepper(eng, ws, dp, make_inputs(B, total, 67))
    log("timing")
    dt, host_enq = timed(one_step, args.warmup, args.steps)
    scal = ws.scalars.cpu().numpy()
    comm = comm_probe(dp, one_step, total) if use_dp else None
    # a device-side gate or a persistent sweep that gave up waiting would mean the streams lost their order: no number
    # then -- and every rank leaves together (a lone exit would park the others in a barrier)
    why = healthy(eng)
    if why:
        if use_dp:
            dist.destroy_process_group()
        raise SystemExit("bench.py: stream ordering was lost during the timed steps " + why)

    out = None
    if rank == 0:
        ms = 1e3 * dt / args.steps
        seqs = B * world * args.steps / dt
        f_seq = 3 * fwd_flops_per_seq(V, EMB, H, Z, C, L, T)
        f_exec = executed_flops_per_step(V, EMB, H, Z, C, L, T, B)
        out = {
            "metric": "SELFIES sequences/sec (whole node), AR-CVAE training step",
            "value": seqs, "unit": "sequences/s", "n_gpus": world, "n_ranks_seen": dist.get_world_size() if use_dp else 1,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": f"{wl_name}, bs {B}/GPU, T 128, tf 0.9, "
                                   f"beta 0 (epoch-0 schedule), fwd+bwd+Adam ({wl_ref})",
                       "global_batch": B * world, "seq_len": T, "parallelism": f"dp{world}",
                       "launch_mode": mode, "fallback": fallback, "dp_form": dp_form(dp), "gates_ok_all_ranks": True},
            "elbo": {"total": float(scal[0]), "recon": float(scal[1]), "kl": float(scal[2]),
                     "mutual_info": float(scal[7])},
            # algorithmic = the REFERENCE's FLOPs for this step (decoder over B*T positions, embedding
            # as per-token GEMM rows) / time; executed = what the engine's kernels really contract (token tables,
            # decoder over B*V pairs): the hardware-use figure
            "step_tflops_algorithmic": seqs * f_seq / 1e12,
            "step_algorithmic_frac_of_f32_mfma_peak": seqs * f_seq / 1e12 / (PEAK_F32_MFMA_TFLOPS * world),
            "step_tflops_executed": f_exec / (dt / args.steps) / 1e12,
            "step_executed_frac_of_f32_mfma_peak": f_exec / (dt / args.steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
        }
        if comm is not None:
            out["comm"] = comm
        log(f"timed: {ms:.3f} ms/step, {seqs:.0f} seq/s (host enqueue {1e3 * host_enq / args.steps:.3f} ms/step)")
    if rank == 0 and not args.no_roofline:
        nb = T + 2 * (L - 1)
        bw = ws.trace_bwd.cpu().numpy().reshape(-1, 2).astype(np.float64)[:nb] / 100.0   # us, last timed step
        out["roofline"] = roofline_probe(eng, ws, torch)
        if nb > 1 and bw[-1, 0] > bw[0, 0]:
            per = float((bw[-1, 0] - bw[0, 0]) / (nb - 1))
            out["roofline"]["in_step_us_per_launch"] = per
            out["roofline"]["in_step_frac"] = out["roofline"]["flop_per_launch"] / (per * 1e-6) / 1e12 / out["roofline"]["peak"]
        log("roofline probe done")

    # ---- strong-scaling leg (BASELINE.json configs[3]: global batch 2048 over the N ranks; N = 1 included so that the
    # 1 -> N ratio comes from one tool).  Its own engine / workspace, timed after the weak leg.
    G2 = args.strong_global_batch
    if G2 > 0 and args.config == "default" and not args.batch_per_gpu:
        strong = None
        if G2 % world != 0:
            strong = {"global_batch": G2, "skipped": f"{G2} rows do not split evenly over {world} ranks"}
        else:
            rows = G2 // world
            del one_step, dp
            eng = ws = None
            torch.cuda.empty_cache()
            eng2, ws2, dp2 = make_engine(rows, G2)
            stotal = args.strong_warmup + args.strong_steps
            step2 = stepper(eng2, ws2, dp2, make_inputs(rows, stotal, 167))
            for i in range(args.strong_warmup):
                step2(i)
            torch.cuda.synchronize()
            why = healthy(eng2)
            if not why:
                dt2, _ = timed(step2, args.strong_warmup, args.strong_steps)
                comm2 = comm_probe(dp2, step2, stotal) if use_dp else None
                why = healthy(eng2)
            if why:
                strong = {"global_batch": G2, "rows_per_gpu": rows, "skipped": why}
            else:
                sc2 = ws2.scalars.cpu().numpy()
                f_ex2 = executed_flops_per_step(V, EMB, H, Z, C, L, T, rows)
                strong = {"global_batch": G2, "rows_per_gpu": rows, "n_gpus": world, "steps": args.strong_steps,
                          "warmup": args.strong_warmup, "ms_per_step": 1e3 * dt2 / args.strong_steps,
                          "value": G2 * args.strong_steps / dt2, "unit": "sequences/s", "scaling": "strong",
                          "elbo": {"total": float(sc2[0]), "recon": float(sc2[1]), "kl": float(sc2[2])},
                          "step_tflops_executed_per_gpu": f_ex2 / (dt2 / args.strong_steps) / 1e12,
                          "step_executed_frac_of_f32_mfma_peak": f_ex2 / (dt2 / args.strong_steps) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                          "dp_form": dp_form(dp2), "gates_ok_all_ranks": True,
                          "bptt_kernel": ("lstm_bwd_persist_rs_kernel" if E.bptt_reduce_scatter_ok(ws2, dims)
                                          else "per-step launches (lstm_bwd_step*/tile kernels)")}
                if getattr(ws2, "planes", False):   # the leg's contractions run in the three-piece form (tile regime)
                    strong["step_executed_frac_of_bf16_pipe"] = 6.0 * strong["step_tflops_executed_per_gpu"] / 2500.0
                    vend = vendor_bf16_gemm_tflops(torch, dev) if rank == 0 else None
                    if vend:
                        strong["vendor_bf16_gemm_tflops"] = vend
                        strong["step_executed_frac_of_vendor_bf16_gemm"] = 6.0 * strong["step_tflops_executed_per_gpu"] / vend
                if comm2 is not None:
                    strong["comm"] = comm2
                if rank == 0 and not args.no_roofline:
                    strong["roofline"] = roofline_probe(eng2, ws2, torch)       # this leg's own dominant kernel, live
                log(f"strong leg: global batch {G2} = {rows} rows/GPU x {world}: {strong['ms_per_step']:.3f} ms/step, "
                    f"{strong['value']:.0f} seq/s")
            del step2, eng2, ws2, dp2
            # ---- N = 1: the per-GPU shard of the 8-GPU run (global batch / 8 rows) timed on this one GPU, and what the two
            # single-GPU figures imply for 8 GPUs BEFORE communication -- an estimate, labelled as such, until the driver's
            # 8-GPU run measures the real thing
            if (world == 1 and strong is not None and "value" in strong and args.shard_steps > 0 and G2 % 8 == 0
                    and not args.force_dp):
                torch.cuda.empty_cache()
                srows = G2 // 8
                eng5, ws5, dp5 = make_engine(srows, srows)
                w5 = 8
                step5 = stepper(eng5, ws5, dp5, make_inputs(srows, w5 + args.shard_steps, 267))
                for i in range(w5):
                    step5(i)
                torch.cuda.synchronize()
                why5 = healthy(eng5)
                if not why5:
                    dt5, _ = timed(step5, w5, args.shard_steps)
                    why5 = healthy(eng5)
                if why5:
                    strong["shard_probe"] = {"rows": srows, "skipped": why5}
                else:
                    t_sh, t_full = dt5 / args.shard_steps, dt2 / args.strong_steps
                    f_ex5 = executed_flops_per_step(V, EMB, H, Z, C, L, T, srows)
                    strong["shard_probe"] = {
                        "rows": srows, "ms_per_step": 1e3 * t_sh, "value": srows / t_sh, "unit": "sequences/s",
                        "steps": args.shard_steps, "warmup": w5,
                        "step_tflops_executed": f_ex5 / t_sh / 1e12,
                        "step_executed_frac_of_f32_mfma_peak": f_ex5 / t_sh / 1e12 / PEAK_F32_MFMA_TFLOPS,
                        "fwd_kernel": ("lstm_fwd_persist_kernel, two-group form (two blocks per CU)"
                                       if _lib.load().arcvae_enc_lstm_persist_groups(srows, H, L) == 2 else
                                       ("lstm_fwd_persist_kernel" if E.persistent_forward_ok(ws5, dims) else "per-step launches")),
                        "bptt_kernel": ("lstm_bwd_persist_rs_kernel" if E.bptt_reduce_scatter_ok(ws5, dims)
                                        else "per-step launches (lstm_bwd_step*/tile kernels)"),
                        "what": f"the {srows}-row shard one GPU of the 8-GPU strong-scaling run steps (no collectives here)"}
                    if not args.no_roofline:
                        strong["shard_probe"]["roofline"] = roofline_probe(eng5, ws5, torch)
                    strong["estimate_8gpu"] = {
                        "speedup_before_communication": 8.0 * (srows / t_sh) / (G2 / t_full),
                        "how": f"8 x ({srows} rows / {1e3 * t_sh:.3f} ms) / ({G2} rows / {1e3 * t_full:.3f} ms), both measured in "
                               "this process on ONE GPU; an ESTIMATE: the stats all-reduce on the chain and the exposed encoder "
                               "bucket are not in it (unmeasured on more than one GPU)",
                        "north_star_target": 6.5}
                    log(f"shard probe: {srows} rows: {1e3 * t_sh:.3f} ms/step = {srows / t_sh:.0f} seq/s -> 8-GPU estimate "
                        f"{strong['estimate_8gpu']['speedup_before_communication']:.2f}x before communication")
                del step5, eng5, ws5
        if rank == 0:
            out["strong"] = strong
    # ---- bf16 throughput-mode leg (SURVEY.md 8(d) Config 2 "bf16-in/fp32-acc"): the same workload on
    # StepEngine(precision="bf16"), reported next to -- never instead of -- the fp32 headline.  Not a parity path.
    if (world == 1 and args.precision == "fp32" and args.bf16_steps > 0 and args.config == "default"
            and not args.batch_per_gpu and not args.force_dp):
        eng = ws = None
        torch.cuda.empty_cache()
        eng3, ws3, dp3 = make_engine(B, B, precision="bf16")
        w3 = max(5, args.warmup // 2)
        step3 = stepper(eng3, ws3, dp3, make_inputs(B, w3 + args.bf16_steps, 67))
        for i in range(w3):
            step3(i)
        torch.cuda.synchronize()
        why = healthy(eng3)
        if not why:
            dt3, _ = timed(step3, w3, args.bf16_steps)
            why = healthy(eng3)
        if why:
            out["bf16_mode"] = {"skipped": why}
        else:
            sc3 = ws3.scalars.cpu().numpy()
            out["bf16_mode"] = {
                "dtype": "bf16", "ms_per_step": 1e3 * dt3 / args.bf16_steps, "value": B * args.bf16_steps / dt3,
                "unit": "sequences/s", "steps": args.bf16_steps, "warmup": w3,
                "elbo": {"total": float(sc3[0]), "recon": float(sc3[1]), "kl": float(sc3[2])},
                "what": "bf16 operands, f32 accumulation: the persistent sweeps' 4x4 MFMA blocks on v_mfma_f32_4x4x4_16b_bf16 "
                        "(48 instead of 192 matrix instructions per wave and tick) and the decoder's B*V-row GEMMs; parameters, "
                        "optimizer state, exchange, gates and cell state stay f32; the weight-gradient GEMMs beside the sweep "
                        "keep their split-bf16 (fp32-accurate) form",
                "tolerance": "loss 2e-2, gradients 8e-2 relative L2 / cosine > 0.995 vs the fp64 oracle "
                             "(tests/test_bf16_mode_gpu.py); NOT the 1e-4 parity path"}
            log(f"bf16 mode: {out['bf16_mode']['ms_per_step']:.3f} ms/step, {out['bf16_mode']['value']:.0f} seq/s")
        del step3, eng3, ws3
    # ---- BASELINE.json configs[2] (H512 Z256 L4, bs 512, "MFMA-bound regime") in fp32 and in bf16 throughput mode: its own
    # parameter stores, engine and inputs, timed in this process after the headline legs; never the headline.
    if (world == 1 and args.precision == "fp32" and args.configs2_steps > 0 and args.config == "default"
            and not args.batch_per_gpu and not args.force_dp):
        eng = ws = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        H2, Z2, L2, B2, w2 = 512, 256, 4, 512, 3
        dims2 = E.ModelDims(V=V, E=EMB, H=H2, Z=Z2, C=C, L=L2)
        enc2 = ParamStore(encoder_shapes(V, EMB, H2, Z2, C, L2), dev)
        dec2 = ParamStore(decoder_shapes(V, EMB, H2, Z2, C, L2), dev)
        rs2 = np.random.RandomState(67)
        xs2, cs2, es2 = [], [], []
        for _ in range(4):
            x2, c2 = synth(rs2, B2)
            xs2.append(torch.tensor(x2, device=dev)); cs2.append(torch.tensor(c2, device=dev))
            es2.append(torch.tensor(rs2.standard_normal((B2, Z2)).astype(np.float32), device=dev))
        coins2 = torch.tensor((np.random.RandomState(4242).rand(w2 + args.configs2_steps, T) < TF_RATIO).astype(np.uint8), device=dev)
        f_seq2 = 3 * fwd_flops_per_seq(V, EMB, H2, Z2, C, L2, T)
        legs = {}
        for prec in ("fp32", "bf16"):
            gen = torch.Generator().manual_seed(1234)
            enc2.init_mlx_like(H2, gen)
            dec2.init_mlx_like(H2, gen)
            enc2.p("fc_logvar.bias").fill_(0.35)
            for st in (enc2, dec2):
                st.grad.zero_(); st.adam_m.zero_(); st.adam_v.zero_()
            eng4 = E.StepEngine(enc2, dec2, dims2, precision=prec)
            eng4.mode = mode
            ws4 = eng4.workspace(B2, T, train=True)
            eng4.set_hyper(ws4, **HYPER)

            def step4(i):
                k = i % len(xs2)
                eng4.load_inputs(ws4, xs2[k], cs2[k], es2[k], coins2[i])
                eng4.run_step(ws4, LR, update=True)
            for i in range(w2):
                step4(i)
            torch.cuda.synchronize()
            why = healthy(eng4)
            if not why:
                dt4, _ = timed(step4, w2, args.configs2_steps)
                why = healthy(eng4)
            if why:
                legs[prec] = {"skipped": why}
            else:
                sq = B2 * args.configs2_steps / dt4
                f_ex4 = executed_flops_per_step(V, EMB, H2, Z2, C, L2, T, B2)
                legs[prec] = {"dtype": "f32" if prec == "fp32" else "bf16", "ms_per_step": 1e3 * dt4 / args.configs2_steps,
                              "value": sq, "unit": "sequences/s", "steps": args.configs2_steps, "warmup": w2,
                              "step_tflops_algorithmic": sq * f_seq2 / 1e12,
                              "step_tflops_executed": f_ex4 / (dt4 / args.configs2_steps) / 1e12,
                              "elbo_total": float(ws4.scalars.cpu().numpy()[0])}
                if prec == "fp32":
                    legs[prec]["frac_of_f32_mfma_peak"] = sq * f_seq2 / 1e12 / PEAK_F32_MFMA_TFLOPS
                    legs[prec]["executed_frac_of_f32_mfma_peak"] = legs[prec]["step_tflops_executed"] / PEAK_F32_MFMA_TFLOPS
                    legs[prec]["frac_note"] = ("frac_of_f32_mfma_peak is SURVEY 8(d)'s stated ratio (algorithmic f32 FLOP / 157.3 TFLOP/s) and "
                                               "can exceed 1: the three-piece kernels issue bf16 products, not f32 MFMAs -- read the two "
                                               "executed_frac_of_* fields against the pipe in use")
                    # every large contraction of this leg runs in the three-piece form (sweeps, plane GEMMs, dense decoder stack):
                    # the bf16 products issued per second against the pipe they issue on
                    legs[prec]["executed_frac_of_bf16_pipe"] = 6.0 * legs[prec]["step_tflops_executed"] / 2500.0
                    # ... and against what the vendor's bf16 GEMM sustains on this device on random data (measured here)
                    vend = vendor_bf16_gemm_tflops(torch, dev)
                    if vend:
                        legs[prec]["vendor_bf16_gemm_tflops"] = vend
                        legs[prec]["executed_frac_of_vendor_bf16_gemm"] = 6.0 * legs[prec]["step_tflops_executed"] / vend
                    if not args.no_roofline:
                        legs[prec]["roofline"] = roofline_probe(eng4, ws4, torch)   # the MFMA-bound regime's dominant kernel, live
                else:
                    legs[prec]["frac_of_bf16_mfma_peak"] = sq * f_seq2 / 1e12 / 2500.0
                    legs[prec]["note"] = "throughput mode: not a parity path (tests/test_bf16_mode_gpu.py states its tolerance)"
                log(f"configs[2] {prec}: {legs[prec]['ms_per_step']:.2f} ms/step, {sq:.0f} seq/s")
            del step4, eng4, ws4
            torch.cuda.empty_cache()
        out["other_configs"] = {"configs[2]": {"workload": "big AR-CVAE V80 E128 H512 Z256 C1 L4, bs 512/GPU, T 128, tf 0.9, "
                                                           "fwd+bwd+Adam (BASELINE.json configs[2])", **legs}}
    # ---- BASELINE.json configs[4]: autoregressive sampling, 10k molecules, bs 1024 (N = 1 only; replicas need no collective)
    if (world == 1 and args.precision == "fp32" and args.sampler_reps > 0 and args.config == "default"
            and not args.batch_per_gpu and not args.force_dp):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        out.setdefault("other_configs", {})["configs[4]"] = sampler_leg(torch, dev, args.sampler_reps)
        sl = out["other_configs"]["configs[4]"]
        log(f"configs[4] sampler: {sl['max_length_80']['ms_per_10k']:.2f} ms / 10k molecules at max_length 80 "
            f"({sl['max_length_80']['molecules_per_s']:.0f} molecules/s), {sl['max_length_128']['ms_per_10k']:.2f} ms at 128")
    if rank == 0:
        if args.cpu_steps > 0 and args.config == "default" and world == 1:   # the CPU baseline is an N = 1 figure
            log(f"cpu baseline on {host_cores()} host cores")
            out["cpu_baseline"] = cpu_baseline(args.cpu_steps)
            out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]   # (vs_baseline stays null: BASELINE.md
            # holds no published number for this metric; the CPU figure is this repo's restatement, kind "port")
        emit(json.dumps(out))
    if use_dp:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def _lib_load():
    from arcvae_hip import _lib
    return _lib.load()


def roofline_probe(eng, ws, torch):
    """Live HIP-event timing of the dominant kernel on the stream it is launched on.

    Dominant kernel of a training step at this shape = the BPTT sweep of the encoder stack, in whichever family the
    engine runs it (profiles/): lstm_bwd_persist_rs_kernel (persistent reduce-scatter sweep: up to 128 rows per GPU at
    H 256, one launch per chunk, a "launch" below = one TICK of it), lstm_bwd_tile_ks3_kernel / lstm_bwd_tile_kernel (the
    register-tiled step kernels of the MFMA-bound regime: bs 2048, configs[2]) or lstm_bwd_step / step2_kernel (per-step
    launches).  A tick / launch carries 2L-1 single-source jobs (L cell steps dG_{t+1} . Wh^T and L-1 input-gradient
    projections dG^{l+1}_t . Wx^T), each a [B,4H] x [4H,H] contraction = 2*B*4H*H FLOP.
    Timed as the whole sweep through the engine's own call (replayed as one linear hipGraph) between two events on
    the launching stream, divided by the tick / launch count: launches of a dependent chain are back to back, so this
    is the per-launch figure rocprofv3's kernel trace reports (its start stamp of launch n+1 is the end stamp of n).
    `traffic`: fabric-side bytes per launch from rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE, MI355X_MICROARCH.md
    HBM section), taken from profiles/ (separate --pmc passes), not measured here.
    """
    from arcvae_hip import engine as E
    d = eng.d
    B, Tn = ws.B, ws.T
    wx, _k1 = E._layer_ptrs(eng.enc, d.L, "Wx", skip0=True)
    wh, _k2 = E._layer_ptrs(eng.enc, d.L, "Wh")
    launches = Tn + 2 * (d.L - 1)
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    persistent = E.bptt_reduce_scatter_ok(ws, d)   # the BPTT sweep of this shape is ONE persistent launch per chunk
    plan = E.EncoderBackwardPlan(eng.enc, ws, d)
    fused = bool(getattr(plan, "fused", False))    # ... which also forms the stack's weight gradients (FW variant)

    def sweep():   # the whole sweep through the engine's own call: persistent (one launch of `launches` ticks) or launches
        plan.sweep(0, launches, None, 0)

    sweep()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        sweep()
    s = torch.cuda.current_stream()
    g.replay()
    torch.cuda.synchronize()
    if persistent:   # one replay = one launch: bracket every replay by its own pair of events, take the median
        times = []
        for _ in range(reps):
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record(s)
            g.replay()
            a1.record(s)
            torch.cuda.synchronize()
            times.append(a0.elapsed_time(a1))
        us = 1e3 * sorted(times)[len(times) // 2] / launches
    else:
        e0.record(s)
        for _ in range(reps):
            g.replay()
        e1.record(s)
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps / launches
    flops_total = 2.0 * B * 4 * d.H * d.H * (d.L * (Tn - 1) + (d.L - 1) * Tn)   # dh = dG.Wh and dX = dG.Wx contractions
    if fused:
        flops_total *= 2.0    # + dWh_l += dG_l^T.h_l[t-1] (L(T-1) of them) and dWx_l += dG_l^T.h_{l-1}[t] ((L-1)T): same count
    ach = flops_total / launches / (us * 1e-6) / 1e12
    tiled = (not persistent) and bool(_lib_load().arcvae_enc_lstm_tiled_for(B, d.H, d.L, E._lstm_flags(ws)) & 2)
    bf16 = getattr(eng, "precision", "fp32") == "bf16" and (persistent or tiled)   # the sweep really runs bf16 blocks / tiles
    kernel = ("lstm_bwd_persist_rs_kernel" if persistent else
              (("lstm_bwd_tile_ks_kernel" if bf16 else "lstm_bwd_tile_ks3_kernel / lstm_bwd_tile_kernel") if tiled else
               ("lstm_bwd_step2_kernel" if B >= 256 else "lstm_bwd_step_kernel")))
    peak = 2500.0 if bf16 else PEAK_F32_MFMA_TFLOPS       # MI355X_MICROARCH.md: dense bf16 MFMA ~2.5 PFLOP/s
    # three-piece tile kernels (the fp32 parity path of the MFMA-bound regime): every f32 product is SIX products on
    # v_mfma_f32_16x16x32_bf16 -- what the kernel executes on the bf16 matrix pipe, next to the section-8(d) f32 figure
    from arcvae_hip import _lib as _L
    three_piece = bool(tiled and not bf16 and (E._lstm_flags(ws) & _L.LSTM_SPLIT3))
    # traffic: fabric-side bytes per tick / launch from the newest committed rocprofv3 --pmc summary (separate passes,
    # tools/pmc.sh; 2*FETCH_SIZE + WRITE_SIZE per the gfx950 correction) -- a profile figure, named by its file, not
    # measured in this run.  tick_model: where a tick's time goes (tools/probe_persist.hip on the same chip).
    def newest(pattern):
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        return files[-1] if files else None
    traffic, traffic_source = None, None
    pmc = newest("r*_pmc_summary.json")
    if pmc:
        try:
            ent = json.load(open(pmc)).get(kernel.split(" / ")[0], {})   # (a family is looked up by its first, dominant member)
            if "shape" in ent and list(ent["shape"]) != [B, d.H, d.L]:
                ent = {}     # (counters of this kernel were taken at another shape [B, H, L]: no figure for this leg)
            traffic = ent.get("bytes_per_tick", ent.get("bytes_per_launch"))
            traffic_source = "profiles/" + os.path.basename(pmc) + " (rocprofv3 --pmc, separate passes; not measured in this run)"
        except Exception:
            traffic = None
    tick_model = None
    tm = newest("r*_tick_model.json")
    if tm and persistent:
        try:
            tick_model = json.load(open(tm))
            tick_model["source"] = "profiles/" + os.path.basename(tm)
        except Exception:
            tick_model = None
    if persistent:
        note = (("dominant kernel: the persistent BPTT sweep with the stack's weight gradients formed inside it "
                 "(lstm_bwd_persist_rs_kernel, FW variant: ONE launch per step; DESIGN.md section 6d); a 'launch' here is "
                 "one TICK of it: the 2L-1 contractions of the per-step kernel it replaced PLUS the 2L-1 weight-gradient "
                 "outer products of the same tick." if fused else
                 "dominant kernel: the persistent BPTT sweep (lstm_bwd_persist_rs_kernel, one launch per chunk; DESIGN.md "
                 "section 6b); a 'launch' here is one TICK of it (same 2L-1 contractions as a launch of the per-step "
                 "kernel it replaced).") + "  achieved = isolated sweep as ONE launch (HIP events on its stream) / ticks; "
                "in_step_* = tick cadence inside the last timed step (device-side stamps, weight-gradient GEMMs beside it). "
                "`bound` names the section-8(d) denominator (f32-input MFMA peak, exact-f32 path); the LIMITER is the "
                "tick latency of a 259-tick dependent chain (exchange through the XCD's L2, block barriers, epilogue), "
                "not MFMA issue or HBM bytes: see tick_model")
    elif tiled:
        note = ("achieved = isolated BPTT sweep (HIP events on its stream) / launches; the register-tiled step kernels of the "
                "MFMA-bound regime in their three-piece bf16 form (fp32-class accuracy, six bf16 products per f32 product): a "
                "block owns 64 rows x 64 units, its four waves contract a quarter of K = 4H each into a 64 x 64 register tile, "
                "operands straight from k-chunk-major bf16 planes, partial tiles through LDS, four adjacent units per lane in the "
                "cell epilogue.  `bound` = f32-input MFMA peak (the section-8(d) denominator of the ALGORITHMIC f32 FLOP); what is "
                "left is operand delivery (64 B/clk per CU at the matrix rate), the launch seam and the epilogue (DESIGN.md 6g)")
    else:
        note = ("achieved = isolated BPTT sweep (HIP events on its stream); in_step_* = start-to-start cadence of the "
                "same launches inside the last timed step (device-side stamps, side-stream GEMMs running beside "
                "them).  `bound` names the section-8(d) denominator (f32-input MFMA peak); the LIMITER is the "
                "dependent-chain seam (1.6 us boundary + cold operand fetch of ~128 KB per CU), see DESIGN.md section 6")
    if bf16:
        note += "  (throughput mode: bf16 operands, so `peak` is the dense bf16 MFMA peak; not the parity path)"
    out = {"bound": "mfma", "limiter": "operand delivery + launch seam" if tiled else "latency", "kernel": kernel, "achieved": ach,
           "peak": peak,
           "unit": "TFLOP/s", "frac": ach / peak, "traffic": traffic, "traffic_source": traffic_source,
           "us_per_launch": us, "launches_per_sweep": launches,
           "flop_per_launch": flops_total / launches, "tick_model": tick_model, "note": note}
    if three_piece:
        # ("98 % of the f32 peak" is a ratio against a pipe these kernels do not run on)
        out["executed_tflops_on_bf16_pipe"] = 6.0 * ach
        out["executed_frac_of_bf16_pipe"] = 6.0 * ach / 2500.0
        vend = vendor_bf16_gemm_tflops(torch, torch.device("cuda", torch.cuda.current_device()))
        if vend:
            out["vendor_bf16_gemm_tflops"] = vend
            out["executed_frac_of_vendor_bf16_gemm"] = 6.0 * ach / vend
        out["pipe_note"] = ("three-piece form: six v_mfma_f32_16x16x32_bf16 products per f32 product -- `achieved` / `frac` price the "
                            "f32-equivalent FLOP against the f32 MFMA peak (SURVEY 8(d)); executed_* price the bf16 products "
                            "actually issued against the dense bf16 peak (2.5 PFLOP/s) and against the vendor library's bf16 GEMM "
                            "(8192^3, random operands) measured on this device in this run")
    return out


if __name__ == "__main__":
    sys.exit(main())
