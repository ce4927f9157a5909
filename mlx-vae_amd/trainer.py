"""ARCVAETrainerWithLoss on MI355X (reference trainer.py:12-736).

Same constructor, schedules, epoch structure, `history` keys, checkpoint file names and the same
consumption order of NumPy's global RNG stream (batch shuffles and teacher-forcing coins, SURVEY
Q5/Q16), so an epoch visits the same batches with the same coins as the reference would.  The
step itself (loss, gradients, two un-bias-corrected Adam updates; trainer.py:292-333) is one
captured hipGraph replay of the arcvae_hip engine.

Reproduced on purpose: gradient clipping is a no-op (Q6: the reference's `_clip_gradients` sums
nothing, so `grad_clip` never changes a gradient); a "loss explosion" only drops the value from
the running mean, the update has already been applied (Q15); every 25th batch a second forward is
run for logging with post-update weights and T more coins (Q16).
Checkpoints: same file names, but a flat non-pickle .npz keyed by the parameter names (Q21).

Data parallelism (the reference is single-process): under `torch.distributed` (train.py started by
`python -m torch.distributed.run`, one process per GPU) every rank runs THIS epoch flow on the same dataset order and the
same coins (one seeded NumPy stream), every batch is the GLOBAL batch, and the step / the loss forwards work on the
rank's row shard with the stats seam and the gradient buckets all-reduced over RCCL (arcvae_hip.dp.EngineDataParallel,
enabled by `api.enable_data_parallel`).  The loss scalars every rank logs are the global batch's, so the history is the
single-process history; rank 0 alone prints and writes checkpoints / history / plots.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from arcvae_hip import api
from complete_vae_loss import complete_vae_loss

try:  # progress bars are cosmetic
    from tqdm import tqdm
except Exception:  # pragma: no cover
    def tqdm(it, **kw):
        return it


class ARCVAETrainerWithLoss:
    def __init__(self, encoder, decoder, property_predictor, dataset, learning_rate: float = 1e-4,
                 batch_size: int = 32, beta_start: float = 0.0, beta_end: float = 0.4,
                 beta_warmup_epochs: int = 100, lambda_prop: float = 0.1, lambda_collapse: float = 0.01,
                 free_bits: float = 0.5, lambda_mi: float = 0.01, grad_clip: float = 1.0,
                 checkpoint_dir: str = "./checkpoints", progress: bool = True):
        if property_predictor is not None:
            raise NotImplementedError("property_predictor is unreachable in the reference (Q10)")
        self.encoder, self.decoder, self.property_predictor = encoder, decoder, None
        self.dataset, self.batch_size, self.grad_clip = dataset, batch_size, grad_clip
        self.lambda_prop, self.lambda_collapse = lambda_prop, lambda_collapse
        self.free_bits, self.lambda_mi = free_bits, lambda_mi
        self.beta_start, self.beta_end, self.beta_warmup_epochs = beta_start, beta_end, beta_warmup_epochs
        self.learning_rate = learning_rate
        self.checkpoint_dir = Path(checkpoint_dir)
        self.checkpoint_dir.mkdir(exist_ok=True)
        self.progress = progress
        self.persist_best_val = False  # Q21: the reference never writes best_val_loss
        self.best_val_loss = float("inf")
        self.engine = self._make_engine(encoder, decoder)
        self.rank, self.world = self._rank_world()
        if self.rank != 0:
            self.progress = False
        # set when a step reported a lost stream order: the encoder's update was skipped on the device, but the decoder's
        # (applied ~1 ms earlier, at the end of its own segment) may already be in -- the two modules can be one step
        # apart, so this state is never checkpointed
        self.poisoned = False
        self.history = {k: [] for k in (
            "epoch", "train_loss", "train_recon", "train_kl", "train_collapse", "train_prop", "val_loss",
            "val_recon", "val_kl", "val_collapse", "val_prop", "beta", "teacher_forcing", "learning_rate",
            "mutual_info")}

    # ---- the three calls into the step engine (tests drive the same epoch flow with oracle-backed ones) ----------
    def _make_engine(self, encoder, decoder):
        return api.engine_for(encoder, decoder)

    def _rank_world(self) -> Tuple[int, int]:
        dp = api.data_parallel_of(self.encoder, self.decoder)
        return (dp.rank, dp.world) if dp is not None else (0, 1)

    def _train_step(self, molecules, conditions, teacher_forcing_ratio: float, hyper: Dict[str, float]):
        """loss + grads + (no-op clip, Q6) + both Adam updates: one captured step (N ranks: arcvae_hip.dp).  Returns
        [total_loss, step status] of the GLOBAL batch as one tensor: read together, one host sync per batch."""
        out, _ = api.value_and_grad(self.encoder, self.decoder, molecules, conditions,
                                    teacher_forcing_ratio=teacher_forcing_ratio, lr=self.learning_rate, **hyper)
        return out["loss_and_status"]

    def _encode(self, molecules, conditions):
        return self.encoder(molecules, conditions)

    # ---- schedules (trainer.py:102-114) ------------------------------------------------------
    def compute_beta(self, epoch: int) -> float:
        if epoch < self.beta_warmup_epochs:
            return float(self.beta_start + (self.beta_end - self.beta_start) * (epoch / self.beta_warmup_epochs))
        return float(self.beta_end)

    def compute_teacher_forcing_ratio(self, epoch: int, total_epochs: int) -> float:
        return float(max(0.5, 0.9 - 0.4 * (epoch / total_epochs)))

    # ---- helpers -------------------------------------------------------------------------------
    def _hyper(self, beta: float) -> Dict[str, float]:
        return dict(beta=beta, lambda_collapse=self.lambda_collapse, lambda_mi=self.lambda_mi, target_mi=4.85,
                    free_bits=self.free_bits)

    def _loss_dict(self, molecules, conditions, beta: float, tf: float) -> Dict[str, torch.Tensor]:
        return complete_vae_loss(self.encoder, self.decoder, None, molecules, conditions, beta=beta,
                                 lambda_prop=self.lambda_prop, lambda_collapse=self.lambda_collapse,
                                 teacher_forcing_ratio=tf, free_bits=self.free_bits, lambda_mi=self.lambda_mi,
                                 target_mi=4.85)

    def _eval_batches(self, dataset, beta: float, limit: Optional[int], desc: str) -> Dict[str, float]:
        tot = dict(loss=0.0, recon=0.0, kl=0.0, collapse=0.0, prop=0.0)
        n = 0
        it = dataset.to_batches(self.batch_size, shuffle=False)
        if self.progress and limit is None:
            it = tqdm(it, total=len(dataset) // self.batch_size, desc=desc)
        for bi, (mol, cond) in enumerate(it):
            if limit is not None and bi >= limit:
                break
            d = self._loss_dict(mol, cond, beta, 0.0)  # teacher_forcing_ratio = 0.0: pure argmax chain
            vals = torch.stack([d["total_loss"], d["recon_loss"], d["kl_loss"], d["collapse_penalty"],
                                d["prop_loss"]]).tolist()  # one host sync per batch
            for k, v in zip(("loss", "recon", "kl", "collapse", "prop"), vals):
                tot[k] += v
            n += 1
        return {k: (v / n if n > 0 else 0.0) for k, v in tot.items()}

    def _compute_true_train_loss(self, epoch: int, num_batches: int = 10) -> Dict[str, float]:
        """trainer.py:116-175: training loss WITHOUT teacher forcing on the first `num_batches` batches."""
        return self._eval_batches(self.dataset, self.compute_beta(epoch), num_batches, "True train loss")

    def _validate(self, val_dataset, beta: float) -> Dict[str, float]:
        """trainer.py:418-487."""
        return self._eval_batches(val_dataset, beta, None, "Validating")

    # ---- epoch -------------------------------------------------------------------------------------
    def train_epoch(self, epoch: int, total_epochs: int, val_dataset=None) -> Dict[str, float]:
        beta = self.compute_beta(epoch)
        tf = self.compute_teacher_forcing_ratio(epoch, total_epochs)
        self.last_train_metrics = self._train_epoch_batches(beta, tf)
        if self.engine is not None:
            self.engine.check_gates()  # a device-side gate that expired would mean the step's streams lost their order
        true_train = self._compute_true_train_loss(epoch, num_batches=20)
        val = self._validate(val_dataset, beta) if val_dataset is not None else dict(
            loss=0.0, recon=0.0, kl=0.0, collapse=0.0, prop=0.0)
        mu, logvar = self._get_latent_stats()
        mi_value = float(self._compute_mutual_information(mu, logvar))
        return {
            "train_loss": true_train["loss"], "train_recon": true_train["recon"], "train_kl": true_train["kl"],
            "train_collapse": true_train["collapse"], "train_prop": true_train["prop"],
            "val_loss": val.get("loss", 0.0), "val_recon": val.get("recon", 0.0), "val_kl": val.get("kl", 0.0),
            "val_collapse": val.get("collapse", 0.0), "val_prop": val.get("prop", 0.0), "beta": beta,
            "teacher_forcing": tf, "mutual_info": mi_value,
        }

    def _train_epoch_batches(self, beta: float, teacher_forcing_ratio: float) -> Dict[str, float]:
        """trainer.py:242-416."""
        total_loss, num_batches = 0.0, 0
        sums = dict(recon=0.0, kl=0.0, collapse=0.0, prop=0.0)
        comp_count = 0
        hyper = self._hyper(beta)
        it = self.dataset.to_batches(self.batch_size, shuffle=True)
        if self.progress:
            it = tqdm(it, total=len(self.dataset) // self.batch_size, desc="Training batches")
        for batch_idx, (molecules, conditions) in enumerate(it):
            loss_st = self._train_step(molecules, conditions, teacher_forcing_ratio, hyper)
            if batch_idx == 0 or batch_idx % 25 == 0:  # Q16: second forward, post-update weights, T more coins
                d = self._loss_dict(molecules, conditions, beta, teacher_forcing_ratio)
                vals = torch.cat([loss_st, torch.stack([d["recon_loss"], d["kl_loss"], d["collapse_penalty"],
                                                        d["prop_loss"]])]).tolist()
                loss_val, status = vals[0], vals[1]
                for k, v in zip(("recon", "kl", "collapse", "prop"), vals[2:]):
                    sums[k] += v
                comp_count += 1
            else:
                loss_val, status = loss_st.tolist()  # the reference syncs here too (trainer.py:366)
            if status != 0.0:
                # a device-side gate expired or a persistent sweep gave up in THIS step: the device has already
                # skipped both Adam updates (the weights are those of the previous batch); stop here, not at epoch end
                self.poisoned = True
                if self.engine is not None:
                    self.engine.check_gates()
                raise RuntimeError(f"training step {batch_idx}: stream order was lost on the device (the encoder's update "
                                   "was skipped; the decoder's may already be applied: do not checkpoint this state)")
            if not np.isfinite(loss_val) or loss_val > 2000.0 or loss_val < -10.0:  # Q15
                if self.rank == 0:
                    print(f"\nWARNING: loss explosion detected at batch {batch_idx}: {loss_val:.2e} "
                          "(update already applied; value excluded from the epoch mean)")
                self._loss_dict(molecules, conditions, beta, teacher_forcing_ratio)  # reference re-evaluates here
                continue
            total_loss += loss_val
            num_batches += 1
        return {"loss": total_loss / max(1, num_batches),
                **{k: (v / comp_count if comp_count > 0 else 0.0) for k, v in sums.items()}}

    @staticmethod
    def _clip_gradients(grads, max_norm: float = 1.0) -> Tuple:
        """trainer.py:490-522 sums only top-level arrays of the two grad trees; every top-level entry is a
        sub-module dict, so the norm is 0 and the gradients are returned unchanged (Q6)."""
        return grads

    def _get_latent_stats(self):
        """trainer.py:524-545: encoder pass on the first 64 training rows, printed summary."""
        molecules, conditions = next(iter(self.dataset.to_batches(64, shuffle=False)))
        mu, logvar = self._encode(molecules, conditions)      # (N ranks: every rank encodes the same 64 rows)
        if self.rank == 0:
            m, lv = mu.cpu().numpy(), logvar.cpu().numpy()
            print(f"   Latent Stats: mu=[{m.min():.3f}, {m.max():.3f}] (mean={m.mean():.3f}, std={m.std():.3f}), "
                  f"logvar=[{lv.min():.3f}, {lv.max():.3f}] (mean={lv.mean():.3f}, std={lv.std():.3f})")
        return mu, logvar

    @staticmethod
    def _compute_mutual_information(mu, logvar) -> float:
        """trainer.py:547-575: the monitoring variant with log(mean_var + 1e-8) (Q20).  The batch sums come
        from the device (arcvae_latent_stats); the 2Z+3 numbers are finished on the host."""
        from losses._dev import latent_stats
        stats, _, B, Z = latent_stats(mu, logvar, 0.0)
        s = stats.cpu().numpy().astype(np.float32)
        f = np.float32
        mean_mu, mean_var = s[:Z] / f(B), s[Z:2 * Z] / f(B)
        agg = f(-0.5) * np.sum(f(1.0) + np.log(mean_var + f(1e-8)) - mean_mu * mean_mu - mean_var, dtype=np.float32)
        return float(max(s[2 * Z] / f(B) - agg, f(0.0)))

    # ---- checkpoints / history (trainer.py:577-736) -----------------------------------------------------
    def save_checkpoint(self, epoch: int, is_best: bool = False):
        if self.poisoned:
            raise RuntimeError("a training step lost its stream order (encoder and decoder may be one update apart): "
                               "refusing to checkpoint this state; resume from the last checkpoint")
        if self.rank != 0:       # N ranks hold identical weights and optimizer state: rank 0 writes
            return
        ck = {"epoch": np.array(epoch), "history_json": np.array(json.dumps(self.history))}
        for tag, mod in (("encoder", self.encoder), ("decoder", self.decoder)):
            st = mod.store
            for n in st.names():
                ck[f"{tag}_weights/{n}"] = st.p(n).cpu().numpy()
                ck[f"{tag}_optimizer_state/m/{n}"] = st._view(st.adam_m, n).cpu().numpy()
                ck[f"{tag}_optimizer_state/v/{n}"] = st._view(st.adam_v, n).cpu().numpy()
        ck["learning_rate"] = np.array(self.learning_rate)
        if self.persist_best_val:
            ck["best_val_loss"] = np.array(self.best_val_loss)
        if is_best:
            self._save_checkpoint(ck, self.checkpoint_dir / "checkpoint_best.npz")
        self._save_checkpoint(ck, self.checkpoint_dir / f"checkpoint_epoch_{epoch:03d}.npz")

    @staticmethod
    def _save_checkpoint(checkpoint: dict, path: Path):
        np.savez(str(path), **checkpoint)
        print(f"    Saved checkpoint: {path}")

    def load_checkpoint(self, checkpoint_path: str) -> int:
        ck = np.load(checkpoint_path, allow_pickle=False)
        for tag, mod in (("encoder", self.encoder), ("decoder", self.decoder)):
            st = mod.store
            for n in st.names():
                if f"{tag}_weights/{n}" in ck:
                    st.p(n).copy_(torch.from_numpy(ck[f"{tag}_weights/{n}"]))
                if f"{tag}_optimizer_state/m/{n}" in ck:
                    st._view(st.adam_m, n).copy_(torch.from_numpy(ck[f"{tag}_optimizer_state/m/{n}"]))
                    st._view(st.adam_v, n).copy_(torch.from_numpy(ck[f"{tag}_optimizer_state/v/{n}"]))
        if "history_json" in ck:
            self.history = json.loads(str(ck["history_json"]))
        return int(ck["epoch"]) if "epoch" in ck else 0

    def save_history(self, path: str):
        if self.rank != 0:
            return
        p = Path(path) / "training_history.json"
        with open(p, "w") as f:
            json.dump(self.history, f, indent=2)
        print(f"    Saved history: {p}")

    def plot_history(self, save_path: str = None):
        if self.rank != 0:
            return
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except ImportError:
            print("    matplotlib not available for plotting")
            return
        h = self.history
        panels = [("Total Loss", [("train_loss", "Train"), ("val_loss", "Val")]),
                  ("Loss Components", [("train_recon", "Recon"), ("train_kl", "KL")]),
                  ("Posterior Collapse Penalty", [("train_collapse", "Collapse Penalty")]),
                  ("Property Prediction Loss", [("train_prop", "Train"), ("val_prop", "Val")]),
                  ("Annealing Schedules", [("beta", "Beta"), ("teacher_forcing", "TF Ratio")]),
                  ("Latent Space Health", [("mutual_info", "MI")])]
        fig, axes = plt.subplots(2, 3, figsize=(15, 10))
        for ax, (title, series) in zip(axes.ravel(), panels):
            for key, label in series:
                ax.plot(h["epoch"], h[key], label=label)
            if title == "Latent Space Health":
                ax.axhline(y=4.85, color="r", linestyle="--", label="Target")
                ax.axhline(y=1.0, color="orange", linestyle="--", label="Collapse")
            ax.set_xlabel("Epoch")
            ax.set_title(title)
            ax.legend()
            ax.grid(True, alpha=0.3)
        plt.tight_layout()
        if save_path:
            plt.savefig(save_path, dpi=150)
            print(f"    Saved plot: {save_path}")
        plt.close(fig)
