"""The C-ABI library loads and exports exactly what include/arcvae_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "arcvae_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\bint\s+(arcvae_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(2).split(",") if a.strip()]
        decls[m.group(1)] = len(args)
    return decls


def test_header_declares_the_path():
    d = _declared()
    for name in ("arcvae_enc_lstm_forward", "arcvae_enc_lstm_backward", "arcvae_dec_forward_dense",
                 "arcvae_dec_backward_dense", "arcvae_latent_loss", "arcvae_adam_update", "arcvae_gemm_f32"):
        assert name in d


def test_library_exports_every_declared_symbol():
    from arcvae_hip import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the extension first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in arcvae_hip.h but not exported"


def test_binding_signatures_match_header_arity():
    from arcvae_hip import _lib
    d = _declared()
    assert set(_lib.SIGNATURES) == set(d), set(_lib.SIGNATURES) ^ set(d)
    for name, n in d.items():
        assert len(_lib.SIGNATURES[name]) == n, (name, len(_lib.SIGNATURES[name]), n)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from arcvae_hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        _lib.load()
    except _lib.ArcvaeHipError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("load() must raise when the HIP extension is missing")


def test_cpu_tensors_are_refused():
    import torch
    from arcvae_hip import _lib
    try:
        _lib.ptr(torch.zeros(4))
    except _lib.ArcvaeHipError:
        pass
    else:
        raise AssertionError("CPU tensors must be refused: there is no CPU path")
