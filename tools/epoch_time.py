"""End-to-end epoch wall time of the reference-shaped trainer (BASELINE.json configs[0] shape: N=1000 rows,
80/10/10 split, bs 64, T 128) including the logging forwards, true-train-loss and validation passes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mlx-vae_amd"))
import numpy as np, torch
import train
from mlx_data.dataloader import MoleculeDataset
from models.vae import ARCVAE
from trainer import ARCVAETrainerWithLoss
np.random.seed(67)
data = train.synthetic_dataset(1000, 80)
props = np.array([[m["tpsa"]] for m in data["molecules"]], dtype=np.float32)
seqs = data["tokenized_sequences"]
idx = np.arange(1000); np.random.shuffle(idx)
tr, va = idx[:800], idx[800:900]
trd = MoleculeDataset([seqs[i] for i in tr], props[tr], max_length=128)
vad = MoleculeDataset([seqs[i] for i in va], props[va], max_length=128, properties_mean=trd.properties_mean, properties_std=trd.properties_std)
vae = ARCVAE(80, 128, 256, 128, 1, 2)
t = ARCVAETrainerWithLoss(vae.encoder, vae.decoder, None, trd, batch_size=64, learning_rate=2e-4, beta_start=0.0, beta_end=0.05,
                          beta_warmup_epochs=20, lambda_collapse=0.001, free_bits=1.0, lambda_mi=0.01, checkpoint_dir="/tmp/ck_epoch", progress=False)
for ep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m = t.train_epoch(ep, 30, vad)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"epoch {ep}: {dt*1e3:.1f} ms wall ({800/dt:.0f} train seq/s incl. eval passes)  train_loss={m['train_loss']:.4f} val_loss={m['val_loss']:.4f} "
          f"recon={m['train_recon']:.4f} kl={m['train_kl']:.3f} mi={m['mutual_info']:.3f} beta={m['beta']:.4f}")
