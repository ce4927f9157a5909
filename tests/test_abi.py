"""The C-ABI library loads and exports exactly what include/arcvae_hip.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "arcvae_hip.h")


def _declared_args():
    """name -> list of C parameter declarations (comments stripped, whitespace normalised)."""
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|long)\s+(arcvae_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        decls[m.group(1)] = [" ".join(a.split()) for a in m.group(2).split(",") if a.strip()]
    return decls


def _declared():
    return {k: len(v) for k, v in _declared_args().items()}


def _ctype_of(decl: str):
    """The ctypes class a C parameter declaration must be bound with (the binding's conventions: pointer-to-pointer
    = POINTER(c_void_p) host arrays of device pointers, int* / long* = POINTER(c_int / c_long) host arrays, every other pointer and the
    stream = c_void_p)."""
    ty = re.sub(r"\b\w+$", "", decl).strip() if not decl.endswith("*") else decl   # drop the parameter name
    ty = ty.replace("const", "").replace(" ", "")
    if ty.count("*") == 2:
        return ctypes.POINTER(ctypes.c_void_p)
    if ty == "int*":
        return ctypes.POINTER(ctypes.c_int)
    if ty == "long*":
        return ctypes.POINTER(ctypes.c_long)
    if "*" in ty or ty == "arcvae_stream_t":
        return ctypes.c_void_p
    return {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
            "unsigned": ctypes.c_uint, "unsignedint": ctypes.c_uint, "unsignedlonglong": ctypes.c_ulonglong}[ty]


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


def test_binding_signatures_match_header_types():
    """Every parameter's C type in include/arcvae_hip.h against the ctypes class it is bound with -- not just the
    count: an int bound as a pointer (or a double as a float) would pass the arity check and corrupt the call."""
    from arcvae_hip import _lib
    for name, decls in _declared_args().items():
        want = [_ctype_of(d) for d in decls]
        got = _lib.SIGNATURES[name]
        for i, (w, g, d) in enumerate(zip(want, got, decls)):
            assert w is g or (w == g), (name, i, d, w, g)


def test_csrc_takes_its_declarations_from_the_public_header():
    """include/arcvae_hip.h is the ONLY place the C ABI is declared: csrc/common.h includes it (with hipStream_t behind
    arcvae_stream_t), so a definition that drifts from its declaration is a compile error; nothing in csrc re-declares a flag
    bit, an error code or an extern "C" prototype (round 3's header had gone stale that way: ARCVAE_LSTM_SPLIT3 and the 3/2
    plane sizes lived only in csrc/ops.h)."""
    src_dir = os.path.join(ROOT, "mlx-vae_amd", "csrc")
    header = open(HEADER).read()
    public = set(re.findall(r"^#define\s+(ARCVAE_\w+)", header, flags=re.M))
    assert {"ARCVAE_LSTM_SPLIT3", "ARCVAE_LSTM_BF16", "ARCVAE_ERR_ARG", "ARCVAE_DEC_PART_TAIL"} <= public
    assert '#include "arcvae_hip.h"' in open(os.path.join(src_dir, "common.h")).read()
    assert "-I../../include" in open(os.path.join(src_dir, "Makefile")).read()
    assert not re.search(r'^\s*extern "C"', open(os.path.join(src_dir, "ops.h")).read(), flags=re.M)
    for fn in os.listdir(src_dir):
        if not fn.endswith((".hip", ".h")):
            continue
        text = open(os.path.join(src_dir, fn)).read()
        again = set(re.findall(r"^\s*#define\s+(ARCVAE_\w+)", text, flags=re.M)) & (public - {"ARCVAE_HIP_BUILD"})
        assert not again, f"{fn} re-defines {sorted(again)} (declared in include/arcvae_hip.h)"
    # every entry point the header declares is DEFINED extern "C" in csrc under exactly that name
    defined = set()
    for fn in os.listdir(src_dir):
        if fn.endswith(".hip"):
            defined |= set(re.findall(r'extern "C"\s+(?:int|long)\s+(arcvae_\w+)\s*\(', open(os.path.join(src_dir, fn)).read()))
            for blk in re.findall(r'extern "C"\s*\{(.*?)\n\}', open(os.path.join(src_dir, fn)).read(), flags=re.S):
                defined |= set(re.findall(r"^(?:int|long)\s+(arcvae_\w+)\s*\(", blk, flags=re.M))
    missing = set(_declared()) - defined
    assert not missing, f"declared in the header, no extern \"C\" definition found in csrc: {sorted(missing)}"


def test_header_keeps_no_hidden_host_state():
    """The header promises re-entrancy: no entry point may depend on a previous call through mutable host globals.
    Checked at the source level: no mutable `static` / namespace-scope variable in csrc (read-once `static const`
    environment knobs excepted), and no set-and-forget entry point in the ABI."""
    src_dir = os.path.join(ROOT, "mlx-vae_amd", "csrc")
    for fn in os.listdir(src_dir):
        if not fn.endswith((".hip", ".h")):
            continue
        for ln, line in enumerate(open(os.path.join(src_dir, fn)), 1):
            code = line.split("//")[0]
            if re.search(r"^\s*static\s+(?!const|inline|constexpr|__device__|int tile_all_weights)", code) and "static_assert" not in code:
                raise AssertionError(f"{fn}:{ln}: mutable static host state: {line.strip()}")
    assert "arcvae_set_step_trace" not in _declared()


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
