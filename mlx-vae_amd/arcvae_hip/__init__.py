"""arcvae_hip: MI355X (gfx950) kernels + host driver for the AR-CVAE SELFIES training path.

`_lib`    ctypes binding of libarcvae_hip.so (C ABI in include/arcvae_hip.h)
`store`   flat parameter / gradient / Adam-state buffers with the reference's parameter names
`engine`  the training step as a two-stream launch sequence captured into a hipGraph
`dp`      data-parallel wrapper (torch.distributed / RCCL)
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
