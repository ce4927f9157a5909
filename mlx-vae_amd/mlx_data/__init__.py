"""Data module under the reference's name (mlx_data/__init__.py)."""
from .dataloader import MoleculeDataset

__all__ = ["MoleculeDataset"]
