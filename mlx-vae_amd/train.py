#!/usr/bin/env python3
"""Training CLI for the AR-CVAE SELFIES path on MI355X.

Drop-in for the reference's `python train.py ...` (train.py:17-255): identical flag names, types and
argparse defaults (note Q17: these are the ARGPARSE defaults, not the README's), identical seed,
80/10/10 split, checkpoint clearing / `--resume` behaviour, history keys and output files.
Additive flags: --device, --synthetic N (generate a SELFIES-shaped dataset when the JSON is absent;
the reference's dataset blob is not distributed), --no_progress, --precision, --world_size / --dist_backend.

Data parallel over the GPUs of one node (the reference is single-process):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29544 \
           train.py --batch_size 2048 ...
one process per GPU (RANK / LOCAL_RANK / WORLD_SIZE from the launcher); --batch_size stays the GLOBAL batch, every rank
steps its row shard, loss statistics and gradients are all-reduced over RCCL (arcvae_hip/dp.py), and the history, the
checkpoints and the plots (written by rank 0) are those of the single-process run on the same batches.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# (flag, type, default, help) -- reference train.py:21-54
FLAGS = [
    ("--data", str, "mlx_data/chembl_cns_selfies.json", "Path to dataset JSON file"),
    ("--vocab_size", int, 80, "Vocabulary size"),
    ("--embedding_dim", int, 128, "Embedding dimension"),
    ("--hidden_dim", int, 256, "Hidden dimension"),
    ("--latent_dim", int, 128, "Latent dimension"),
    ("--num_conditions", int, 1, "Number of conditions"),
    ("--num_layers", int, 2, "Number of LSTM layers"),
    ("--dropout", float, 0.2, "Dropout rate (accepted and ignored, as in the reference)"),
    ("--epochs", int, 30, "Number of epochs"),
    ("--batch_size", int, 32, "Batch size"),
    ("--learning_rate", float, 2e-4, "Learning rate"),
    ("--beta_start", float, 0.0, "Initial beta value"),
    ("--beta_end", float, 0.05, "Final beta value"),
    ("--beta_warmup_epochs", int, 20, "Beta warmup epochs"),
    ("--lambda_prop", float, 0.1, "Property loss weight"),
    ("--lambda_collapse", float, 0.001, "Posterior collapse weight"),
    ("--free_bits", float, 1.0, "Free bits constraint (min KL per dimension)"),
    ("--lambda_mi", float, 0.01, "Mutual information penalty weight"),
    ("--grad_clip", float, 1.0, "Gradient clipping norm (a no-op in the reference, Q6)"),
    ("--checkpoint_dir", str, "./checkpoints", "Checkpoint directory"),
    ("--checkpoint_freq", int, 10, "Checkpoint frequency (epochs)"),
]


def build_parser() -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(description="Train AR-CVAE for molecular generation (MI355X)")
    for flag, typ, default, hlp in FLAGS:
        ap.add_argument(flag, type=typ, default=default, help=hlp)
    ap.add_argument("--verbose", action="store_true", help="Print detailed epoch summaries")
    ap.add_argument("--resume", action="store_true",
                    help="Resume from checkpoint_best.npz in checkpoint directory (otherwise old checkpoints are cleared)")
    ap.add_argument("--device", type=str, default=None, help="HIP device, e.g. cuda:0 (extension)")
    ap.add_argument("--synthetic", type=int, default=0, help="use N synthetic SELFIES-shaped rows (extension)")
    ap.add_argument("--no_progress", action="store_true", help="disable progress bars (extension)")
    ap.add_argument("--world_size", type=int, default=None,
                    help="number of data-parallel ranks (extension; default: WORLD_SIZE of the launcher, else 1).  Start the "
                         "ranks with `python -m torch.distributed.run --nproc-per-node N train.py ...`")
    ap.add_argument("--dist_backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend (extension): nccl = RCCL over xGMI, one rank per GPU; gloo carries "
                         "device tensors through the host (several ranks on one GPU: tests)")
    ap.add_argument("--precision", choices=["fp32", "bf16"], default=None,
                    help="fp32: the parity path (default); bf16: throughput mode -- matrix products on bf16 operands with "
                         "f32 accumulation, parameters and optimizer state in f32 (extension; ARCVAE_PRECISION does the same)")
    return ap


def synthetic_dataset(n: int, vocab: int, max_length: int = 128):
    """Same schema as the reference JSON (train.py:79-83,102): molecules[i].tpsa, tokenized_sequences, max_length."""
    rs = np.random.RandomState(67)
    seqs, mols = [], []
    for _ in range(n):
        ln = int(rs.randint(20, max_length - 1))
        seqs.append([int(t) for t in rs.randint(3, vocab, size=ln)] + [2])
        mols.append({"tpsa": float(rs.gamma(4.0, 20.0))})
    return {"molecules": mols, "tokenized_sequences": seqs, "max_length": max_length}


def init_data_parallel(args):
    """(rank, world): joins the launcher's process group when WORLD_SIZE / --world_size > 1 (one process per GPU; the
    device defaults to cuda:LOCAL_RANK), else (0, 1) without touching torch.distributed."""
    world = args.world_size if args.world_size is not None else int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) != world:
        raise SystemExit(f"--world_size {world}: start the ranks with `python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {world} --master-addr 127.0.0.1 --master-port 29544 train.py ...`")
    import torch
    import torch.distributed as dist
    rank, local_rank = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    if args.device is None:
        args.device = f"cuda:{local_rank}"
    torch.cuda.set_device(torch.device(args.device))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: what RCCL needs on this host driver
    kw = dict(device_id=torch.device(args.device)) if args.dist_backend == "nccl" else {}
    dist.init_process_group(args.dist_backend, rank=rank, world_size=world, **kw)
    return rank, world


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.precision:                      # read when the step engine of the model is created (arcvae_hip/engine.py)
        os.environ["ARCVAE_PRECISION"] = args.precision
    rank, world = init_data_parallel(args)
    import builtins
    print = builtins.print if rank == 0 else (lambda *a, **k: None)   # noqa: A001  N ranks run the same flow; rank 0 reports
    args.no_progress = args.no_progress or rank != 0
    from mlx_data.dataloader import MoleculeDataset
    from models.vae import ARCVAE
    from trainer import ARCVAETrainerWithLoss

    print("=" * 80 + "\nAR-CVAE Training (MI355X)\n" + "=" * 80)
    if world > 1:
        print(f"  Data parallel: {world} ranks ({args.dist_backend}), global batch {args.batch_size}")
    print(f"  Dataset: {args.data if not args.synthetic else f'synthetic x{args.synthetic}'}")
    print(f"  Model: embedding={args.embedding_dim}, hidden={args.hidden_dim}, latent={args.latent_dim}")
    print(f"  Training: epochs={args.epochs}, batch_size={args.batch_size}, lr={args.learning_rate}")
    print(f"  Beta: start={args.beta_start}, end={args.beta_end}, warmup={args.beta_warmup_epochs}")

    np.random.seed(67)  # train.py:75 (the only seed the reference sets)
    if args.synthetic:
        data = synthetic_dataset(args.synthetic, args.vocab_size)
    else:
        with open(args.data, "r") as f:
            data = json.load(f)
    properties = np.array([[mol["tpsa"]] for mol in data["molecules"]], dtype=np.float32)
    sequences = data["tokenized_sequences"]

    indices = np.arange(len(sequences))
    np.random.shuffle(indices)
    n_total = len(sequences)
    n_train, n_val = int(0.8 * n_total), int(0.1 * n_total)
    parts = {"train": indices[:n_train], "val": indices[n_train:n_train + n_val], "test": indices[n_train + n_val:]}

    def make(idx, mean=None, std=None):
        return MoleculeDataset([sequences[i] for i in idx], properties[idx], max_length=data["max_length"],
                               pad_token=0, properties_mean=mean, properties_std=std, device=args.device)

    train_dataset = make(parts["train"])
    val_dataset = make(parts["val"], train_dataset.properties_mean, train_dataset.properties_std)
    test_dataset = make(parts["test"], train_dataset.properties_mean, train_dataset.properties_std)
    print(f"Loaded {n_total:,} samples: train {len(train_dataset):,} / val {len(val_dataset):,} / "
          f"test {len(test_dataset):,}; property mean {train_dataset.properties_mean.flatten()}, "
          f"std {train_dataset.properties_std.flatten()}")

    checkpoint_dir = Path(args.checkpoint_dir)
    start_epoch, best_val_loss = 0, float("inf")
    if args.resume:
        ckpt = checkpoint_dir / "checkpoint_best.npz"
        if not ckpt.exists():
            raise FileNotFoundError(f"Checkpoint not found: {ckpt}")
        print(f"\nResuming from checkpoint: {ckpt}")
    elif checkpoint_dir.exists() and rank == 0:
        for f in checkpoint_dir.glob("*.npz"):
            f.unlink()
        plot = checkpoint_dir / "training_history.png"
        if plot.exists():
            plot.unlink()

    generator = None
    if world > 1:       # identical initial weights on every rank: rank 0's seed (the reference's init is unseeded MLX RNG)
        import torch
        import torch.distributed as dist
        seed = torch.tensor([torch.seed() % (2 ** 31)], dtype=torch.int64,
                            device=args.device if args.dist_backend == "nccl" else "cpu")
        dist.broadcast(seed, src=0)
        generator = torch.Generator().manual_seed(int(seed.item()))
    vae = ARCVAE(vocab_size=args.vocab_size, embedding_dim=args.embedding_dim, hidden_dim=args.hidden_dim,
                 latent_dim=args.latent_dim, num_conditions=args.num_conditions, num_layers=args.num_layers,
                 dropout=args.dropout, device=args.device, generator=generator)
    if world > 1:
        from arcvae_hip import api
        api.enable_data_parallel(vae.encoder, vae.decoder)
    trainer = ARCVAETrainerWithLoss(
        encoder=vae.encoder, decoder=vae.decoder, property_predictor=None, dataset=train_dataset,
        batch_size=args.batch_size, learning_rate=args.learning_rate, beta_start=args.beta_start,
        beta_end=args.beta_end, beta_warmup_epochs=args.beta_warmup_epochs, lambda_prop=args.lambda_prop,
        lambda_collapse=args.lambda_collapse, free_bits=args.free_bits, lambda_mi=args.lambda_mi,
        grad_clip=args.grad_clip, checkpoint_dir=args.checkpoint_dir, progress=not args.no_progress)
    if args.resume:
        loaded = trainer.load_checkpoint(str(checkpoint_dir / "checkpoint_best.npz"))
        start_epoch = loaded + 1  # best_val_loss restarts at inf, as in the reference (Q21)
        print(f"Loaded model weights from epoch {loaded}")

    for epoch in range(start_epoch, args.epochs):
        print(f"\nEpoch {epoch + 1}/{args.epochs}")
        metrics = trainer.train_epoch(epoch=epoch, total_epochs=args.epochs, val_dataset=val_dataset)
        trainer.history["epoch"].append(epoch)
        for k in ("train_loss", "train_recon", "train_kl", "train_collapse", "train_prop", "val_loss", "val_recon",
                  "val_kl", "val_collapse", "val_prop", "beta", "teacher_forcing", "mutual_info"):
            trainer.history[k].append(metrics[k])
        trainer.history["learning_rate"].append(args.learning_rate)
        is_best = metrics["val_loss"] < best_val_loss
        if is_best:
            best_val_loss = metrics["val_loss"]
            trainer.best_val_loss = best_val_loss
        if (epoch + 1) % args.checkpoint_freq == 0 or is_best:
            trainer.save_checkpoint(epoch=epoch, is_best=is_best)
            trainer.save_history(args.checkpoint_dir)
        if args.verbose:
            print(f"Epoch {epoch + 1}/{args.epochs}: Train Loss: {metrics['train_loss']:.4f}, "
                  f"Val Loss: {metrics['val_loss']:.4f}, Beta: {metrics['beta']:.4f}")
    trainer.plot_history(save_path=f"{args.checkpoint_dir}/training_history.png")
    print("\nTraining complete")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return trainer


if __name__ == "__main__":
    main()
